// Would a 32-deep K stage pay for the 64x64 tile?  Synthetic stage loop of conv_igemm_buf_f32<64,64> (per wave: 4
// ds_read_b128, 8 MFMAs, 1 barrier, 2 LDS-DMA loads, ~14 scalar ops) against the BK=32 shape (8 reads, 16 MFMAs, 1 barrier,
// 4 DMA loads, ~18 scalar ops), same total MFMA count.   hipcc --offload-arch=gfx950 -O3 tools/mfma_bk.hip -o tools/mfma_bk.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KH>  // KH = number of 8-deep K halves per stage: 2 (BK=16) or 4 (BK=32)
__global__ __launch_bounds__(256) void stage_loop(const float* __restrict__ src, float* out, int stages) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int STAGE_F = 128 * 8 * KH;  // (64 + 64) rows x BK floats
  for (int i = tid; i < 3 * STAGE_F; i += 256) smem[i] = 1e-3f * (i & 255);
  __syncthreads();
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int r32 = lane & 31, hh = lane >> 5;
  const float* a_frag = smem + ((wave >> 1) * 32 + r32) * 8 * KH + hh * 4;
  const float* b_frag = smem + (64 + (wave & 1) * 32 + r32) * 8 * KH + hh * 4;
  unsigned voff = (unsigned)(((size_t)blockIdx.x * 1024 + tid * 4) % (1 << 18)) * 4u;
  const unsigned wave_lds = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)wave * 1024u);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 rsrc;
  const unsigned long long b = (unsigned long long)src;
  rsrc.x = __builtin_amdgcn_readfirstlane((unsigned)b);
  rsrc.y = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
  rsrc.z = 0x80000000u;
  rsrc.w = 0x00020000u;
  constexpr int LPW = KH;  // DMA loads per wave per stage: (128 rows * BK*4 B) / 1 KiB / 4 waves
  unsigned soff = 0;
  auto step = [&](auto slot_c) {
    constexpr int SLOT = decltype(slot_c)::value;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned dst = wave_lds + ((SLOT + 2) % 3) * (STAGE_F * 4u);
    unsigned keep;
    if constexpr (LPW == 2)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
                   "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "s"(dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "scc");
    else
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
                   "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
                   "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
                   "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "s"(dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "scc");
    soff = (soff + 64u * KH) & 0xffffu;
    float4 fa[KH], fb[KH];
#pragma unroll
    for (int h = 0; h < KH; ++h) {
      fa[h] = *reinterpret_cast<const float4*>(a_frag + SLOT * STAGE_F + h * 8);
      fb[h] = *reinterpret_cast<const float4*>(b_frag + SLOT * STAGE_F + h * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < KH; ++h) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h].x, fb[h].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h].y, fb[h].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h].z, fb[h].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h].w, fb[h].w, acc, 0, 0, 0);
    }
  };
  for (int s = 0; s + 3 <= stages; s += 3) {
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float sum = 0.f;
  for (int e = 0; e < 16; ++e) sum += acc[e];
  out[(size_t)blockIdx.x * 256 + tid] = sum;
}


// Variant: the same 64x64x16 stage, but 2 waves per workgroup, each a 64x32 (MT=2, NT=1) wave tile: 16 MFMAs, 6 fragment
// reads and 4 DMA loads per wave and stage, a 2-wave barrier.
__global__ __launch_bounds__(128) void stage_loop_2w(const float* __restrict__ src, float* out, int stages) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int STAGE_F = 128 * 16;
  for (int i = tid; i < 3 * STAGE_F; i += 128) smem[i] = 1e-3f * (i & 255);
  __syncthreads();
  f32x16 acc[2];
  for (int e = 0; e < 16; ++e) acc[0][e] = acc[1][e] = 0.f;
  const int r32 = lane & 31, hh = lane >> 5;
  const float* a_frag = smem + r32 * 16 + hh * 4;                        // two A blocks: rows r32, 32 + r32
  const float* b_frag = smem + (64 + wave * 32 + r32) * 16 + hh * 4;
  unsigned voff = (unsigned)(((size_t)blockIdx.x * 1024 + tid * 4) % (1 << 18)) * 4u;
  const unsigned wave_lds = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)wave * 1024u);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 rsrc;
  const unsigned long long b = (unsigned long long)src;
  rsrc.x = __builtin_amdgcn_readfirstlane((unsigned)b);
  rsrc.y = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
  rsrc.z = 0x80000000u;
  rsrc.w = 0x00020000u;
  unsigned soff = 0;
  auto step = [&](auto slot_c) {
    constexpr int SLOT = decltype(slot_c)::value;
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned dst = wave_lds + ((SLOT + 2) % 3) * (STAGE_F * 4u);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
                 "s_add_u32 m0, m0, 0x800\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
                 "s_add_u32 m0, m0, 0x800\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
                 "s_add_u32 m0, m0, 0x800\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "scc");
    soff = (soff + 128u) & 0xffffu;
    float4 fa[2][2], fb[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      fa[h][0] = *reinterpret_cast<const float4*>(a_frag + SLOT * STAGE_F + h * 8);
      fa[h][1] = *reinterpret_cast<const float4*>(a_frag + SLOT * STAGE_F + 32 * 16 + h * 8);
      fb[h] = *reinterpret_cast<const float4*>(b_frag + SLOT * STAGE_F + h * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h][i].x, fb[h].x, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h][i].y, fb[h].y, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h][i].z, fb[h].z, acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h][i].w, fb[h].w, acc[i], 0, 0, 0);
    }
  };
  for (int s = 0; s + 3 <= stages; s += 3) {
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float sum = 0.f;
  for (int e = 0; e < 16; ++e) sum += acc[0][e] + acc[1][e];
  out[(size_t)blockIdx.x * 128 + tid] = sum;
}

void run2w(const float* src, float* out, int wg_per_cu) {
  const int stages = 6000, blocks = 256 * wg_per_cu;
  const int lds = 3 * 128 * 16 * 4;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 30; ++i) hipLaunchKernelGGL(stage_loop_2w, dim3(blocks), dim3(128), lds, 0, src, out, stages);
  hipEventRecord(a);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(stage_loop_2w, dim3(blocks), dim3(128), lds, 0, src, out, stages);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double flops = 10.0 * blocks * 2 * (double)(stages / 3 * 3) * 16.0 * 2 * 32 * 32 * 2;
  printf("2-wave 64x64 (wave tile 64x32) WG/CU=%d: %.1f TFLOP/s\n", wg_per_cu, flops / ms / 1e9);
}

template <int KH>
void run(const float* src, float* out, int wg_per_cu) {
  const int stages = 6000 / KH * 2, blocks = 256 * wg_per_cu;  // same MFMA count for both shapes
  auto k = stage_loop<KH>;
  const int lds = 3 * 128 * 8 * KH * 4;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 30; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, src, out, stages);
  hipEventRecord(a);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, src, out, stages);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double flops = 10.0 * blocks * 4 * (double)(stages / 3 * 3) * (4.0 * KH) * 2 * 32 * 32 * 2;
  printf("BK=%d WG/CU=%d: %.1f TFLOP/s\n", 8 * KH, wg_per_cu, flops / ms / 1e9);
}

int main() {
  float *src, *out;
  hipMalloc(&src, (size_t)(1 << 21) * sizeof(float));
  hipMemset(src, 0, (size_t)(1 << 21) * sizeof(float));
  hipMalloc(&out, (size_t)256 * 8 * 256 * sizeof(float));
  for (int occ : {2, 3, 4, 6}) {
    run<2>(src, out, occ);
    if (occ <= 3) run<4>(src, out, occ);
  }
  for (int occ : {4, 6, 8, 12}) run2w(src, out, occ);
  return 0;
}
