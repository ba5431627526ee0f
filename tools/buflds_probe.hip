// Probe: semantics of `buffer_load_dwordx4 ... offen lds` on gfx950 (LDS-DMA through a raw buffer descriptor):
//  (1) do out-of-range lanes write zeros into LDS or leave it untouched?  (2) is the SGPR offset part of the
//  range check?  (3) lane -> LDS placement (M0 base + lane * 16).
// hipcc --offload-arch=gfx950 -O3 tools/buflds_probe.hip -o tools/buflds_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void probe(const float* src, float* out, unsigned num_records, unsigned soff, unsigned oob_lane_mask_lo) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 4];
  const int lane = threadIdx.x;
  for (int i = 0; i < 4; ++i) lds[lane * 4 + i] = -1.f;
  __syncthreads();
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)lds;
  const unsigned long long base = (unsigned long long)src;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 rsrc;
  rsrc.x = (unsigned)base;
  rsrc.y = (unsigned)(base >> 32) & 0xffffu;
  rsrc.z = num_records;
  rsrc.w = 0x00020000u;
  unsigned voff = lane * 16;
  if ((oob_lane_mask_lo >> (lane & 31)) & 1) voff = 0x80000000u;
  rsrc.x = __builtin_amdgcn_readfirstlane(rsrc.x);
  rsrc.y = __builtin_amdgcn_readfirstlane(rsrc.y);
  rsrc.z = __builtin_amdgcn_readfirstlane(rsrc.z);
  rsrc.w = __builtin_amdgcn_readfirstlane(rsrc.w);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_waitcnt vmcnt(0)"
               :: "s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
  __syncthreads();
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = lds[lane * 4 + i];
}

int main() {
  const int N = 4096;
  std::vector<float> h(N);
  for (int i = 0; i < N; ++i) h[i] = (float)i;
  float *src, *out;
  hipMalloc(&src, N * 4);
  hipMalloc(&out, 256 * 4);
  hipMemcpy(src, h.data(), N * 4, hipMemcpyHostToDevice);
  std::vector<float> r(256);
  auto show = [&](const char* what) {
    hipMemcpy(r.data(), out, 256 * 4, hipMemcpyDeviceToHost);
    printf("%s\n  lane0: %g %g %g %g | lane1: %g %g %g %g | lane2: %g .. | lane33: %g %g | lane63: %g %g %g %g\n", what, r[0], r[1],
           r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[33 * 4], r[33 * 4 + 1], r[252], r[253], r[254], r[255]);
  };
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, out, 0x80000000u, 0u, 0u);
  show("in range, soff 0 (expect lane L -> floats 4L..4L+3)");
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, out, 0x80000000u, 0u, 0x2u);
  show("lanes 1 and 33 out of range via voffset (expect 0 if zero-fill, -1 if skipped)");
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, out, 0x80000000u, 1024u, 0x2u);
  show("soff 1024 B (expect +256 on in-range lanes)");
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, out, 512u, 0u, 0u);
  show("num_records 512 B, soff 0 (lanes >= 32 out of range)");
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, out, 512u, 256u, 0u);
  show("num_records 512 B, soff 256: lane 16..31 -> in range only if soff is NOT part of the check");
  hipMemcpy(r.data(), out, 256 * 4, hipMemcpyDeviceToHost);
  printf("  lane15: %g lane16: %g lane31: %g lane32: %g\n", r[60], r[64], r[124], r[128]);
  return 0;
}
