// Sustained matrix-pipe throughput: the v_mfma_f32_32x32x2_f32-only loop of mfma_peak.hip launched back to back
// for ~3 s; prints TFLOP/s per 100 ms window (DVFS / power-cap behaviour under a pure MFMA load).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_sustain.hip -o /tmp/mfma_sustain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a)
    for (int e = 0; e < 16; ++e) s += acc[a][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  float seconds = argc > 1 ? atof(argv[1]) : 3.f;
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  const int iters = 1000, blocks = 256 * 2;
  const double flops = (double)blocks * 4 * iters * 8 * 4 * 2.0 * 32 * 32 * 2;
  const int per_window = 40;  // ~2.7 ms per launch at peak -> ~100 ms windows
  std::vector<hipEvent_t> ev;
  int windows = (int)(seconds * 10);
  for (int w = 0; w <= windows; ++w) {
    hipEvent_t e;
    hipEventCreate(&e);
    ev.push_back(e);
  }
  hipEventRecord(ev[0]);
  for (int w = 0; w < windows; ++w) {
    for (int i = 0; i < per_window; ++i) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(ev[w + 1]);
  }
  hipDeviceSynchronize();
  float t = 0;
  for (int w = 0; w < windows; ++w) {
    float ms;
    hipEventElapsedTime(&ms, ev[w], ev[w + 1]);
    t += ms;
    printf("t=%7.1f ms  %.1f TFLOP/s\n", t, flops * per_window / ms / 1e9);
  }
  return 0;
}
