// Micro-benchmark: cost of back-to-back dependent kernel launches on one stream (the detector issues ~80 per forward).
// hipcc --offload-arch=gfx950 -O3 tools/launch_gap.hip -o tools/launch_gap.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void empty_k(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void spin_k(int* p, int cycles) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while ((long long)(__builtin_amdgcn_s_memtime() - t0) < cycles) {}
  if (p && threadIdx.x == 9999) p[0] = 1;
}
__global__ void touch_k(float* dst, const float* src, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i] + 1.f;
}
int main() {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float ms; int* d; hipMalloc(&d, 64);
  float *x, *y; size_t n = 8 << 20; hipMalloc(&x, n * 4); hipMalloc(&y, n * 4);
  auto t = [&](const char* what, auto launch, int reps) {
    for (int i = 0; i < 20; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    printf("%-64s %7.2f us per launch\n", what, ms * 1e3 / reps);
  };
  t("empty kernel, 1 workgroup", [&] { hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, 0, d); }, 2000);
  t("empty kernel, 1024 workgroups x 256", [&] { hipLaunchKernelGGL(empty_k, dim3(1024), dim3(256), 0, 0, d); }, 2000);
  t("empty kernel, 1024 workgroups x 512, 96 KB LDS", [&] { hipLaunchKernelGGL(empty_k, dim3(1024), dim3(512), 96 * 1024, 0, d); }, 2000);
  t("spin 10k cycles, 512 workgroups x 512", [&] { hipLaunchKernelGGL(spin_k, dim3(512), dim3(512), 0, 0, d, 10000); }, 1000);
  t("spin 50k cycles, 512 workgroups x 512", [&] { hipLaunchKernelGGL(spin_k, dim3(512), dim3(512), 0, 0, d, 50000); }, 500);
  t("stream 32 MB read + 32 MB write (ping-pong x<->y)", [&] { hipLaunchKernelGGL(touch_k, dim3(2048), dim3(256), 0, 0, y, x, n);
                                                                hipLaunchKernelGGL(touch_k, dim3(2048), dim3(256), 0, 0, x, y, n); }, 200);
  return 0;
}
