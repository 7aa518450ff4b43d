// Pure streaming-read bandwidth on gfx950 (a) by bytes in flight and (b) by WHERE the concurrently resident blocks read:
// grid-stride (all blocks sweep one window together) vs one private contiguous region per block (as many independent
// sequential streams as there are blocks - what a kernel with one row chunk per block does).
//   hipcc --offload-arch=gfx950 -O3 -w -o tools/bin/read_bw tools/probes/read_bw.hip && tools/bin/read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// PRIV: block b reads [b * per, (b + 1) * per) front to back, 256 * U * 16 bytes per step
template <int U, bool PRIV>
__global__ __launch_bounds__(256) void k(const f32x4* __restrict__ src, long n4, float* out) {
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  long i, end, stride;
  if (PRIV) {
    const long per = n4 / gridDim.x;
    i = (long)blockIdx.x * per + threadIdx.x;
    end = (long)(blockIdx.x + 1) * per;
    stride = 256L * U;
  } else {
    i = (long)blockIdx.x * 256 * U + threadIdx.x;
    end = n4;
    stride = (long)gridDim.x * 256 * U;
  }
  for (; i + 256L * (U - 1) < end; i += stride) {
    f32x4 x[U];
#pragma unroll
    for (int u = 0; u < U; u++) x[u] = src[i + 256L * u];
#pragma unroll
    for (int u = 0; u < U; u++) s += x[u];
  }
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = s[0];
}

template <int U, bool PRIV>
static void run(const f32x4* src, long n4, float* out, int grid) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<U, PRIV>), dim3(grid), dim3(256), 0, 0, src, n4, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL((k<U, PRIV>), dim3(grid), dim3(256), 0, 0, src, n4, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 3;
  printf("%-28s U=%2d, %4d blocks (%4.1f per CU), %5.1f KB in flight per CU: %.3f ms  %.2f TB/s\n",
         PRIV ? "private region per block" : "grid-stride", U, grid, grid / 256.0, grid / 256.0 * 256 * U * 16 / 1024.0, ms,
         n4 * 16.0 / ms / 1e9);
}

int main() {
  const long bytes = 4L << 30;
  f32x4* src;
  float* out;
  hipMalloc(&src, bytes);
  hipMalloc(&out, 4);
  hipMemset(src, 0, bytes);
  const long n4 = bytes / 16;
  for (int grid : {256, 512, 1024, 2048}) {
    run<2, false>(src, n4, out, grid);
    run<8, false>(src, n4, out, grid);
    run<2, true>(src, n4, out, grid);
    run<4, true>(src, n4, out, grid);
    run<8, true>(src, n4, out, grid);
  }
  return 0;
}
