// Matrix-pipe ceiling micro-benchmark: v_mfma_f32_32x32x2_f32 only, NACC independent accumulators
// per wave, WAVES waves per SIMD.  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a)
    for (int e = 0; e < 16; ++e) s += acc[a][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu, float* out) {
  const int iters = 2000;
  const int blocks = 256 * blocks_per_cu;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double flops = (double)blocks * 4 * iters * 8 * NACC * 2.0 * 32 * 32 * 2;
  printf("nacc=%d blocks/CU=%d: %.3f ms  %.1f TFLOP/s\n", NACC, blocks_per_cu, ms, flops / ms / 1e9);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  for (int bpc = 1; bpc <= 4; ++bpc) {
    run<1>(bpc, out);
    run<2>(bpc, out);
    run<4>(bpc, out);
  }
  return 0;
}
