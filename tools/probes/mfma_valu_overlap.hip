// Do VALU instructions of one wave overlap the MFMAs of ANOTHER wave on the same SIMD (gfx950)?  Blocks of 8 waves on every
// CU (one block per CU: 2 waves per SIMD): waves 0-3 run a chain-free stream of v_mfma_f32_32x32x16_bf16, waves 4-7 a
// stream of independent v_fma_f32 / v_perm / ds_write; each role alone, then both, then both in the SAME wave.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mvo tools/probes/mfma_valu_overlap.hip && /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// mode bit 0: MFMA waves work; bit 1: VALU waves work; mode 4: every wave does both, interleaved (1 MFMA : R VALU)
template <int R, int KIND>
__global__ __launch_bounds__(512) void k(int iters, int mode, float* out) {
  const int wave = threadIdx.x >> 6;
  floatx16 acc[4];
  for (int i = 0; i < 4; i++)
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i); }
  float v[8];
  for (int i = 0; i < 8; i++) v[i] = threadIdx.x * 0.001f + i;
  unsigned u[8];
  for (int i = 0; i < 8; i++) u[i] = threadIdx.x * 77u + i;
  const float c1 = 1.0001f, c2 = 0.0001f;
  if (mode == 4) {
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < R; j++) {
          if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(c1), "v"(c2));
          else asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[j & 7]) : "v"(u[(j + 1) & 7]), "s"(0x07060302u));
        }
      }
    }
  } else if (wave < 4) {
    if (mode & 1)
      for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q], 0, 0, 0);
      }
  } else {
    if (mode & 2)
      for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
          for (int j = 0; j < R; j++) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(c1), "v"(c2));
            else asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[j & 7]) : "v"(u[(j + 1) & 7]), "s"(0x07060302u));
          }
      }
  }
  float s = 0.f;
  for (int i = 0; i < 4; i++)
    for (int r = 0; r < 16; r++) s += acc[i][r];
  for (int i = 0; i < 8; i++) s += v[i] + (float)u[i];
  if (s == 12345.678f) out[0] = s;
}

template <int R, int KIND>
static void run(const char* name) {
  float* out;
  hipMalloc(&out, 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms[5];
  for (int mode = 1; mode <= 4; mode++) {
    hipLaunchKernelGGL((k<R, KIND>), dim3(256), dim3(512), 0, 0, 100, mode, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<R, KIND>), dim3(256), dim3(512), 0, 0, iters, mode, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms[mode], e0, e1);
  }
  // per iteration: 4 MFMA (128 pipe cycles) and 4 R VALU (16 R issue cycles) per wave
  printf("%-10s R=%2d VALU per MFMA: MFMA waves alone %.3f ms | VALU waves alone %.3f ms | both (separate waves) %.3f ms | "
         "same wave interleaved %.3f ms   -> sum %.3f, max %.3f\n", name, R, ms[1], ms[2], ms[3], ms[4], ms[1] + ms[2],
         ms[1] > ms[2] ? ms[1] : ms[2]);
  hipFree(out);
}

int main() {
  run<2, 0>("v_fma_f32");
  run<4, 0>("v_fma_f32");
  run<6, 0>("v_fma_f32");
  run<8, 0>("v_fma_f32");
  run<12, 0>("v_fma_f32");
  run<6, 1>("v_perm_b32");
  run<12, 1>("v_perm_b32");
  return 0;
}
