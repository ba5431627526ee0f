// dma.h - buffer-addressed LDS-DMA helpers shared by the fp32 and bf16 implicit-GEMM kernels (gfx950).
//   `buffer_load_dwordx4 v_off, rsrc, s_off offen lds`: every lane moves 16 bytes from
//   base + v_off + s_off straight into LDS at M0 + 16 * lane; lanes whose offset is >= num_records write zeros and
//   touch no memory (tools/buflds_probe.hip).  See the kernel comments in conv.hip for how the offsets are built.
#pragma once
#include <hip/hip_runtime.h>

namespace me_dma {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOobOffset = 0x80000000u;

__device__ __forceinline__ u32x4 make_rsrc(const void* base) {
  const unsigned long long b = (unsigned long long)base;
  u32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)b);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);  // stride 0 (raw buffer)
  r.z = kOobOffset;                                                      // num_records (bytes)
  r.w = 0x00020000u;                                                     // gfx9 raw-buffer dword 3
  return r;
}

// All of one wave's DMAs of one stage in a single asm block.  M0 is left pointing at the last destination: nothing the
// compiler emits for these kernels reads M0 (gfx950 LDS instructions do not; tools/check_m0.sh greps the ISA), and the two
// s_mov of a save / restore per stage cost matrix-pipe issue slots.  LA A-type loads
// (descriptor ra, scalar offset sa) are followed by LPW - LA B-type loads (rb, sb); LDS destinations advance by STEP.
template <int LPW, int LA, int STEP>
__device__ __forceinline__ void dma_stage(const unsigned (&v)[LPW], u32x4 ra, u32x4 rb, unsigned sa, unsigned sb,
                                          unsigned dst) {
  static_assert(LPW >= 2 && LPW <= 4 && LA >= 1 && LA <= 2, "unsupported DMA shape");
#define ME_DMA_HEAD "s_mov_b32 m0, %[d]\n\ts_nop 0\n\t"
#define ME_DMA_NEXT "s_add_u32 m0, m0, %[st]\n\ts_nop 0\n\t"
#define ME_DMA_A(i) "buffer_load_dwordx4 %[v" #i "], %[ra], %[sa] offen lds\n\t"
#define ME_DMA_B(i) "buffer_load_dwordx4 %[v" #i "], %[rb], %[sb] offen lds\n\t"
#define ME_DMA_TAIL ""
  if constexpr (LPW == 4 && LA == 2) {
    asm volatile(ME_DMA_HEAD ME_DMA_A(0) ME_DMA_NEXT ME_DMA_A(1) ME_DMA_NEXT ME_DMA_B(2) ME_DMA_NEXT ME_DMA_B(3) ME_DMA_TAIL
                 :
                 : [d] "s"(dst), [st] "n"(STEP), [ra] "s"(ra), [rb] "s"(rb), [sa] "s"(sa), [sb] "s"(sb), [v0] "v"(v[0]),
                   [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3])
                 : "memory", "scc");
  } else if constexpr (LPW == 3 && LA == 2) {
    asm volatile(ME_DMA_HEAD ME_DMA_A(0) ME_DMA_NEXT ME_DMA_A(1) ME_DMA_NEXT ME_DMA_B(2) ME_DMA_TAIL
                 :
                 : [d] "s"(dst), [st] "n"(STEP), [ra] "s"(ra), [rb] "s"(rb), [sa] "s"(sa), [sb] "s"(sb), [v0] "v"(v[0]),
                   [v1] "v"(v[1]), [v2] "v"(v[2])
                 : "memory", "scc");
  } else if constexpr (LPW == 2 && LA == 1) {
    asm volatile(ME_DMA_HEAD ME_DMA_A(0) ME_DMA_NEXT ME_DMA_B(1) ME_DMA_TAIL
                 :
                 : [d] "s"(dst), [st] "n"(STEP), [ra] "s"(ra), [rb] "s"(rb), [sa] "s"(sa), [sb] "s"(sb), [v0] "v"(v[0]),
                   [v1] "v"(v[1])
                 : "memory", "scc");
  } else {
    static_assert(LPW == 0, "add the (LPW, LA) combination");
  }
#undef ME_DMA_HEAD
#undef ME_DMA_NEXT
#undef ME_DMA_A
#undef ME_DMA_B
#undef ME_DMA_TAIL
}


// N loads through ONE descriptor (scalar offset s), LDS destinations dst, dst + STEP, ...; M0 saved / restored once.
// OFF selects the first of the N per-lane offsets inside the caller's array (passed by reference and copied by value here:
// the array must stay in registers).
template <int N, int STEP, int OFF, int LEN>
__device__ __forceinline__ void dma_same(const unsigned (&va)[LEN], u32x4 r, unsigned s, unsigned dst) {
  unsigned keep;
  static_assert(N >= 1 && N <= 4 && OFF + N <= LEN, "1..4 loads per block, inside the array");
  unsigned v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = va[OFF + (i < N ? i : 0)];
#define ME_DMA_HEAD "s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\t"
#define ME_DMA_NEXT "s_add_u32 m0, m0, %[st]\n\ts_nop 0\n\t"
#define ME_DMA_L(i) "buffer_load_dwordx4 %[v" #i "], %[r], %[s] offen lds\n\t"
#define ME_DMA_TAIL "s_mov_b32 m0, %[k]"
  if constexpr (N == 1) {
    asm volatile(ME_DMA_HEAD ME_DMA_L(0) ME_DMA_TAIL
                 : [k] "=&s"(keep)
                 : [d] "s"(dst), [st] "n"(STEP), [r] "s"(r), [s] "s"(s), [v0] "v"(v[0])
                 : "memory", "scc");
  } else if constexpr (N == 2) {
    asm volatile(ME_DMA_HEAD ME_DMA_L(0) ME_DMA_NEXT ME_DMA_L(1) ME_DMA_TAIL
                 : [k] "=&s"(keep)
                 : [d] "s"(dst), [st] "n"(STEP), [r] "s"(r), [s] "s"(s), [v0] "v"(v[0]), [v1] "v"(v[1])
                 : "memory", "scc");
  } else if constexpr (N == 3) {
    asm volatile(ME_DMA_HEAD ME_DMA_L(0) ME_DMA_NEXT ME_DMA_L(1) ME_DMA_NEXT ME_DMA_L(2) ME_DMA_TAIL
                 : [k] "=&s"(keep)
                 : [d] "s"(dst), [st] "n"(STEP), [r] "s"(r), [s] "s"(s), [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2])
                 : "memory", "scc");
  } else {
    asm volatile(ME_DMA_HEAD ME_DMA_L(0) ME_DMA_NEXT ME_DMA_L(1) ME_DMA_NEXT ME_DMA_L(2) ME_DMA_NEXT ME_DMA_L(3) ME_DMA_TAIL
                 : [k] "=&s"(keep)
                 : [d] "s"(dst), [st] "n"(STEP), [r] "s"(r), [s] "s"(s), [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]),
                   [v3] "v"(v[3])
                 : "memory", "scc");
  }
#undef ME_DMA_HEAD
#undef ME_DMA_NEXT
#undef ME_DMA_L
#undef ME_DMA_TAIL
}

}  // namespace me_dma
