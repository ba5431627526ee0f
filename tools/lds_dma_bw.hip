// Micro-benchmark: sustained L2 -> LDS rate of `buffer_load_dwordx4 ... lds` (LDS-DMA) per CU, as a function of waves per
// CU, loads in flight per wave, row length (64-byte rows at a stride vs fully contiguous 1 KiB) and footprint (L2-resident
// vs HBM).  Also a plain `global_load_dwordx4` (to VGPRs) variant.  Answers: is ~9 TB/s (14 B/clk/CU) the memory path's
// ceiling, or are the conv kernels latency-bound?
// hipcc --offload-arch=gfx950 -O3 tools/lds_dma_bw.hip -o tools/lds_dma_bw.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH, int MODE>  // MODE 0: LDS-DMA, 1: global_load to VGPRs
__global__ __launch_bounds__(256) void stream(const unsigned char* src, unsigned long long bytes_per_wg, unsigned row_stride,
                                              int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const unsigned long long base = (unsigned long long)(src + (unsigned long long)blockIdx.x * bytes_per_wg);
  u32x4 rsrc;
  rsrc.x = __builtin_amdgcn_readfirstlane((unsigned)base);
  rsrc.y = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32) & 0xffffu);
  rsrc.z = 0x80000000u;
  rsrc.w = 0x00020000u;
  // one instruction = 16 rows x 64 B; row r of the instruction at (lane / 4) * row_stride + (lane & 3) * 16
  const unsigned voff = (unsigned)(lane >> 2) * row_stride + (unsigned)(lane & 3) * 16u;
  const unsigned inst_bytes = 16u * row_stride;          // address advance per instruction (rows are consecutive)
  const unsigned span = (unsigned)bytes_per_wg;
  unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)wave * inst_bytes);
  const unsigned step = __builtin_amdgcn_readfirstlane((unsigned)nw * inst_bytes);
  const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)wave * DEPTH * 1024u);
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (MODE == 0) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0v + d * 1024u), "v"(voff),
                     "s"(rsrc), "s"(soff)
                     : "memory", "m0");
      } else {
        u32x4 v;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
        asm volatile("s_waitcnt vmcnt(%1)\n\tv_xor_b32 %0, %0, %2" : "+v"(acc) : "n"(DEPTH - 1), "v"(v.x));
      }
      soff += step;
      if (soff + inst_bytes > span) soff = __builtin_amdgcn_readfirstlane((unsigned)wave * inst_bytes);
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345678u) sink[0] = acc + lds[threadIdx.x];
}

template <int DEPTH, int MODE>
double run(const unsigned char* src, size_t bytes_per_wg, unsigned row_stride, int wgs, int threads, int iters, unsigned* sink) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const size_t lds = (size_t)(threads / 64) * DEPTH * 1024;
  auto k = stream<DEPTH, MODE>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(wgs), dim3(threads), lds, 0, src, bytes_per_wg, row_stride, iters, sink);
  hipEventRecord(a);
  hipLaunchKernelGGL(k, dim3(wgs), dim3(threads), lds, 0, src, bytes_per_wg, row_stride, iters, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)wgs * (threads / 64) * iters * DEPTH * 1024.0;
  return bytes / (ms * 1e-3) / 1e12;
}

int main() {
  const size_t total = 1ull << 30;
  unsigned char* src;
  unsigned* sink;
  hipMalloc(&src, total);
  hipMalloc(&sink, 64);
  hipMemset(src, 1, total);
  printf("L2 -> CU streaming rate, TB/s (256 CUs; 8 TB/s = 13 B/clk/CU at 2.4 GHz)\n");
  printf("%-58s %8s %8s %8s\n", "configuration", "depth 2", "depth 4", "depth 8");
  struct Cfg { const char* name; size_t per_wg; unsigned stride; int wgs; int threads; } cfgs[] = {
      {"LDS-DMA  16 KB/WG (L2 hits)  64B rows @256B  1 WG/CU x4 waves", 16 << 10, 256, 256, 256},
      {"LDS-DMA  16 KB/WG (L2 hits)  64B rows @256B  2 WG/CU x4 waves", 16 << 10, 256, 512, 256},
      {"LDS-DMA  16 KB/WG (L2 hits)  64B rows @256B  4 WG/CU x4 waves", 16 << 10, 256, 1024, 256},
      {"LDS-DMA  16 KB/WG (L2 hits)  64B rows @256B  8 WG/CU x4 waves", 16 << 10, 256, 2048, 256},
      {"LDS-DMA  16 KB/WG (L2 hits)  contiguous 1KB  4 WG/CU x4 waves", 16 << 10, 64, 1024, 256},
      {"LDS-DMA  16 KB/WG (L2 hits)  contiguous 1KB  8 WG/CU x4 waves", 16 << 10, 64, 2048, 256},
      {"LDS-DMA  1 MB/WG  (HBM)      contiguous 1KB  4 WG/CU x4 waves", 1 << 20, 64, 1024, 256},
      {"LDS-DMA  1 MB/WG  (HBM)      64B rows @256B  4 WG/CU x4 waves", 1 << 20, 256, 1024, 256},
  };
  for (auto& c : cfgs) {
    const int iters = 2000;
    double r2 = run<2, 0>(src, c.per_wg, c.stride, c.wgs, c.threads, iters, sink);
    double r4 = run<4, 0>(src, c.per_wg, c.stride, c.wgs, c.threads, iters / 2, sink);
    double r8 = run<8, 0>(src, c.per_wg, c.stride, c.wgs, c.threads, iters / 4, sink);
    printf("%-58s %8.2f %8.2f %8.2f\n", c.name, r2, r4, r8);
  }
  Cfg g[] = {
      {"global_load to VGPRs 16 KB/WG (L2 hits) 64B rows @256B 4 WG/CU", 16 << 10, 256, 1024, 256},
      {"global_load to VGPRs 16 KB/WG (L2 hits) contiguous     8 WG/CU", 16 << 10, 64, 2048, 256},
  };
  for (auto& c : g) {
    double r2 = run<2, 1>(src, c.per_wg, c.stride, c.wgs, c.threads, 2000, sink);
    double r4 = run<4, 1>(src, c.per_wg, c.stride, c.wgs, c.threads, 1000, sink);
    double r8 = run<8, 1>(src, c.per_wg, c.stride, c.wgs, c.threads, 500, sink);
    printf("%-58s %8.2f %8.2f %8.2f\n", c.name, r2, r4, r8);
  }
  return 0;
}
