// Store-pattern roofline for the stem epilogues: every wave writes 4 KB blocks (32 pixels x 32 channels fp32) either as 16
// stores of 4 B per lane (lane = channel: two 128-byte rows per instruction - the MFMA accumulator layout) or as 4 stores of 16 B
// per lane (after a transpose: eight 128-byte rows per instruction).  hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern.bin
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* y, long long nblk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r32 = lane & 31, hh = lane >> 5;
  const long long stride = (long long)gridDim.x * 4;
  for (long long b = (long long)blockIdx.x * 4 + wave; b < nblk; b += stride) {
    float* base = y + b * 1024;
    if (MODE == 0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) base[((e & 3) + 8 * (e >> 2) + 4 * hh) * 32 + r32] = (float)e;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) reinterpret_cast<float4*>(base)[q * 64 + lane] = make_float4(1.f, 2.f, 3.f, (float)q);
    }
  }
}
int main() {
  const long long bytes = 775ll << 20, nblk = bytes / 4096;
  float* y;
  hipMalloc(&y, bytes);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 2; ++mode)
    for (int grid : {2048, 6656, 16384}) {
      for (int it = 0; it < 3; ++it) { if (mode == 0) k<0><<<grid, 256>>>(y, nblk); else k<1><<<grid, 256>>>(y, nblk); }
      hipEventRecord(a);
      for (int it = 0; it < 10; ++it) { if (mode == 0) k<0><<<grid, 256>>>(y, nblk); else k<1><<<grid, 256>>>(y, nblk); }
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("mode %d (%s) grid %5d: %7.1f us  %.2f TB/s\n", mode, mode ? "16 B / lane x 4" : "4 B / lane x 16", grid, ms / 10 * 1e3, bytes / (ms / 10 * 1e-3) / 1e12);
    }
  return 0;
}
