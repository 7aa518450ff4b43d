// Micro-benchmark: what limits v_mfma_f32_32x32x2_f32 issue on gfx950?  (build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ float lds[2 * 128 * 33 + 2 * 32 * 128];
  floatx16 acc[4];
  for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  const int t = threadIdx.x, lane = t & 63;
  for (int i = t; i < 2 * 128 * 33 + 2 * 32 * 128; i += 256) lds[i] = (float)(i & 7);
  __syncthreads();
  float a0 = lane * 0.001f, a1 = 1.f - a0, b0 = 0.5f, b1 = 0.25f;
  const float* as = lds + (lane & 31) * 33 + (lane >> 5);
  const float* bs = lds + 2 * 128 * 33 + (lane >> 5) * 128 + (lane & 31);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int ks = 0; ks < 16; ks++) {
      if (MODE >= 1) { a0 = as[2 * ks]; a1 = as[32 * 33 + 2 * ks]; b0 = bs[2 * ks * 128]; b1 = bs[2 * ks * 128 + 32]; }
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
    }
    if (MODE >= 2) __syncthreads();
    if (MODE >= 3) {   // LDS writes like store_chunk
      float* d = lds + ((it & 1) * 128 * 33) + (t >> 3) * 33 + (t & 7) * 4;
      for (int ps = 0; ps < 4; ps++) { d[ps * 32 * 33] = a0; d[ps * 32 * 33 + 1] = a1; d[ps * 32 * 33 + 2] = b0; d[ps * 32 * 33 + 3] = b1; }
      __syncthreads();
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  out[blockIdx.x * 256 + t] = s;
}
template <int MODE> void run(const char* name, int blocks) {
  float* out; hipMalloc(&out, blocks * 256 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, iters); hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = 2.0 * 32 * 32 * 2 * 64.0 * iters * 4 * blocks;   // 64 MFMAs per iter per wave, 4 waves
  printf("%-28s blocks=%d: %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, fl / ms / 1e9);
  hipFree(out);
}
int main() {
  for (int blocks : {256, 512, 768}) {
    run<0>("mfma only", blocks);
    run<1>("+ LDS fragment reads", blocks);
    run<2>("+ barrier per chunk", blocks);
    run<3>("+ LDS writes + 2nd barrier", blocks);
  }
  return 0;
}
