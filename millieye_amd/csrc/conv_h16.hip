// conv_h16.hip - the detector's conv blocks with 16-bit storage for gfx950: bfloat16 (half_type 0) or IEEE half
// (half_type 1) activations and weights, fp32 accumulation - BASELINE configs[2] / [4] ("bf16 inference", "fp16 MFMA
// convs").  Opt-in (Darknet.compute_dtype / MILLIEYE_DTYPE); the fp32 path of conv.hip stays the default and the one the
// 1e-3 parity bar is quoted on.  Every kernel is templated on the storage type (H16<F16>: operand vector, MFMA, conversions).
//
//   conv_igemm_buf_h16 : the buffer-addressed LDS-DMA implicit GEMM of conv.hip on v_mfma_f32_32x32x16_bf16 / _f16
//                        (2.5 PFLOP/s dense peak, 16x the fp32 matrix rate).  A stage is KSUB sub-stages of
//                        (BM + BN) rows x 32 channels (64-byte rows - the same 1 KiB-per-DMA LDS image and XOR
//                        swizzle as the fp32 kernel); one ds_read_b128 is one MFMA operand (8 values per lane).
//                        Epilogue: fp32 affine (folded BN / bias) + LeakyReLU + residual, one RNE rounding to the
//                        storage type - through a per-wave LDS transpose so that loads / stores are 16 bytes per lane -
//                        or fp32 output for the detection convs that feed the YOLO decode.
//   (3x3 / stride 1 layers: the patch-resident big-tile generation lives in conv_p8_impl.h / conv_p8_h16.hip, tile ids >= 100;
//   round 1's experimental tile 41 was its first, image-bounded version and is gone.)
//   conv_stem3_h16     : the cin = 3 stem on the VALU (fp32 frames, rounded to the storage type here; fp32 weights holding
//                        values already rounded by the host; 16-bit NHWC out) - the fallback of stem_mfma_h16.hip (same
//                        rounding points) for couts other than 32.
//   maxpool / upsample / add / copy on 16-bit NHWC (16 bytes = 8 channels per lane).
//
// Rounding points (what oracle/darknet_ref.py's storage="bf16" / "f16" modes restate): weights once (host, RNE), every
// stored activation once (after activation + residual), nothing else; accumulation and the affine are fp32.
#include <math.h>
#include <type_traits>
#include <utility>
#include "common.h"
#include "dma.h"

#include "conv16_common.h"

namespace me16 {  // conv_p8_h16.hip: patch-resident big-tile generation (tile ids >= 100)
bool p8_eligible(const Conv16P& p, int tile);
int launch_p8_tile(const Conv16P& p, int tile, hipStream_t stream);
long long p8_workspace_bytes(const Conv16P& p, int tile, int split);
bool ws1x1_eligible(const Conv16P& p);      // conv1x1_ws_h16.hip: tile id 50, weight-stationary streaming 1x1
int launch_ws1x1(const Conv16P& p, hipStream_t stream);
bool ws3x3_eligible(const Conv16P& p);      // conv3x3_ws_h16.hip: tile id 60, weight-stationary 3x3 for cin 32 / 64
int launch_ws3x3(const Conv16P& p, hipStream_t stream);
// conv_kw_h16.hip: small-batch kernel, K split over the eight waves of a workgroup (tile ids 40 / 41)
int launch_kw(const Conv16P& p, int tile, hipStream_t stream);
bool stem_mfma_eligible(const Conv16P& p);  // stem_mfma_h16.hip
int launch_stem_mfma(const Conv16P& p, hipStream_t stream);
}  // namespace me16

namespace {
using namespace me_dma;
// ---------------------------------------------------------------------------------------------
// implicit GEMM on the bf16 matrix cores.  M = n*ho*wo output pixels, N = cout, K = ks*ks*cin.
// Needs cin % (32*KSUB) == 0 (the planner pads the one odd tensor of the tiny cfgs, engine.py).
// ---------------------------------------------------------------------------------------------
// ABL (tuning only, wrong results): 1 = every DMA lane out of range (zero fill: no L2 / HBM traffic), 2 = no MFMAs,
// 3 = no DMA instructions at all.
// MASKED (ABI 13): the instance with the column-class tap masks (me_conv16_desc.tap_mask) - the K walk of a tile visits only the set
// taps of its column class.  Separate instances on purpose: the inference kernels' code does not change.
template <int BM, int BN, int WR, int WC, int KSUB, int MINW = 1, int ABL = 0, int F16 = 0, int MASKED = 0>
__global__ __launch_bounds__(64 * WR * WC, MINW) void conv_igemm_buf_h16(Conv16P p) {
  using frag = typename H16<F16>::v8;
  constexpr int NST = 3;
  constexpr int NW = WR * WC;
  static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");

  constexpr int TM = BM / WR, TN = BN / WC;
  constexpr int MT = TM / 32, NT = TN / 32;
  constexpr int GA = BM / 16, G = (BM + BN) / 16;  // 16-row groups (1 KiB each): A first, then B
  constexpr int LPW = (G + NW - 1) / NW;            // DMA instructions per wave per sub-stage
  constexpr int LA = GA / NW;                       // ... of which A-type
  constexpr unsigned SUB_B = LPW * NW * 1024u;      // bytes per sub-stage image (incl. dummy groups)
  constexpr unsigned STAGE_B = SUB_B * KSUB;
  static_assert(GA % NW == 0, "A groups must split evenly over the waves");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem16;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC;
  const int r32 = lane & 31, hh = lane >> 5;

  int tile_m, tile_n;
  {  // XCD-aware: consecutive tiles of one XCD share weight rows / neighbouring pixels in that XCD's L2
    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    tile_n = wg % p.tiles_n;
    tile_m = wg / p.tiles_n;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int hw = p.ho * p.wo;

  const int img0 = (int)udiv_magic16((unsigned)m0, p.hw_m, p.hw_s);
  const long long img_elems = (long long)p.h * p.w * p.x_pitch;
  const long long bias_elems = ((long long)p.pad * p.w + p.pad) * p.x_pitch;
  const u32x4 rsrc_a = make_rsrc(p.x + (long long)img0 * img_elems - bias_elems);
  const u32x4 rsrc_b = make_rsrc(p.wgt + (long long)n0 * p.ktot);

  const int lrow = lane >> 2;
  constexpr bool KS3 = false;
  unsigned v_base[LPW], v_pad[LA], v_cur[LPW];
  auto setup_lane = [&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int g = wave + NW * j;
    const int row = g * 16 + lrow;
    const int q = (lane & 3) ^ ((row >> 2) & 3);  // source 16-byte chunk (8 channels) for this LDS slot
    v_base[j] = kOobOffset;
    if constexpr (j < LA) {
      unsigned padmask = 0xFFFFFFFFu;
      const int m = m0 + row;
      if (m < p.M) {
        // (every wave of every tile pays this setup, and at 16-bit matrix rates a 1x1 layer's tile is ~500 clocks of MFMAs:
        //  multiply-shift divisions with the host's magic numbers, tap mask from ks row flags x ks column flags)
        const int nimg = (int)udiv_magic16((unsigned)m, p.hw_m, p.hw_s);
        const int rem = m - nimg * hw;
        const int oy = (int)udiv_magic16((unsigned)rem, p.wo_m, p.wo_s), ox = rem - oy * p.wo;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        unsigned ok = 0;
        auto in_w = [&](int kx) { return (unsigned)(ix0 + kx) < (unsigned)p.w ? 1u : 0u; };
        auto in_h = [&](int ky) { return (unsigned)(iy0 + ky) < (unsigned)p.h; };
        if (KS3 || p.ks == 3) {  // straight-line for the filter sizes of the networks (a runtime-ks loop is ~100 instructions)
          const unsigned cols = in_w(0) | in_w(1) << 1 | in_w(2) << 2;
          ok = (in_h(0) ? cols : 0u) | (in_h(1) ? cols << 3 : 0u) | (in_h(2) ? cols << 6 : 0u);
        } else if (p.ks == 1) {
          ok = in_h(0) ? in_w(0) : 0u;
        } else {
          unsigned cols = 0;
          for (int kx = 0; kx < p.ks; ++kx) cols |= in_w(kx) << kx;
          for (int ky = 0; ky < p.ks; ++ky) ok |= (in_h(ky) ? cols : 0u) << (ky * p.ks);
        }
        padmask = ~ok;
        const long long e = (long long)(nimg - img0) * img_elems + ((long long)iy0 * p.w + ix0) * p.x_pitch + bias_elems;
        v_base[j] = (unsigned)(e * 2) + 16u * q;
      }
      v_pad[j] = padmask;
    } else if (g < G) {
      const int co_local = row - BM;
      if (n0 + co_local < p.cout) v_base[j] = (unsigned)co_local * (unsigned)p.ktot * 2u + 16u * q;
    }
    v_cur[j] = ABL == 1 ? kOobOffset : v_base[j];
  };
  static_for(setup_lane, std::make_integer_sequence<int, LPW>{});  // forced expansion: the arrays must stay in VGPRs

  const int sid = blockIdx.y;
  const int s_begin = sid * p.sps;
  const int s_end = (s_begin + p.sps < p.stages) ? s_begin + p.sps : p.stages;
  int tap = 0, cc = 0, ky = 0, kx = 0;  // wave-uniform K walk: filter tap outer, channel chunk inner
  unsigned tmask = 0xFFFFFFFFu;
  if constexpr (MASKED) {               // (whole tiles only: the host refuses a K split with masks; a class is a multiple of BN wide)
    tmask = p.tapmask[n0 / p.mask_cols] & ((1u << (p.ks * p.ks)) - 1u);
    tap = __builtin_ctz(tmask);
    ky = tap / p.ks;
    kx = tap - ky * p.ks;
  } else if (s_begin != 0) {            // (K-split pieces only: whole tiles skip the divisions)
    tap = s_begin / p.cs;
    cc = s_begin - tap * p.cs;
    ky = tap / p.ks;
    kx = tap - ky * p.ks;
  }
  unsigned a_off = 0, b_off = 0;
  auto enter_tap = [&]() {  // VALU work only here: once per filter tap
#pragma unroll
    for (int j = 0; j < LA; ++j) v_cur[j] = (ABL == 1 || ((v_pad[j] >> (MASKED ? (tap & 31) : tap)) & 1u)) ? kOobOffset : v_base[j];
    a_off = (unsigned)(((long long)ky * p.w + kx) * p.x_pitch * 2);
    b_off = (unsigned)tap * (unsigned)p.cin * 2u;
  };
  enter_tap();
  a_off += (unsigned)cc * (64u * KSUB);
  b_off += (unsigned)cc * (64u * KSUB);

  const unsigned wave_lds = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)wave * 1024u);
  auto issue_stage = [&](unsigned lds_dst) {
    if (ABL == 3) return;  // ablation: no DMA at all (LDS reads + MFMAs + barriers only)
#pragma unroll
    for (int u = 0; u < KSUB; ++u) {
      if constexpr (LPW <= 4 && LA <= 2) {
        dma_stage<LPW, LA, NW * 1024>(v_cur, rsrc_a, rsrc_b, a_off + 64u * u, b_off + 64u * u, lds_dst + u * SUB_B);
      } else {  // 192-row tiles: three A groups + two B groups per wave
        constexpr int LB = LPW - LA;
        static_assert(LA <= 4 && LB <= 4, "at most 4 groups per operand per wave");
        dma_same<LA, NW * 1024, 0>(v_cur, rsrc_a, a_off + 64u * u, lds_dst + u * SUB_B);
        dma_same<LB, NW * 1024, LA>(v_cur, rsrc_b, b_off + 64u * u, lds_dst + u * SUB_B + LA * NW * 1024u);
      }
    }
    a_off += 64u * KSUB;
    b_off += 64u * KSUB;
    if (++cc == p.cs) {
      cc = 0;
      do {   // the next tap (MASKED: the next SET tap of this tile's column class)
        ++tap;
        if (++kx == p.ks) {
          kx = 0;
          ++ky;
        }
      } while (MASKED && tap < p.ks * p.ks && !((tmask >> tap) & 1u));
      enter_tap();
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nstages = MASKED ? __builtin_popcount(tmask) * p.cs : s_end - s_begin;
  issue_stage(wave_lds);
  if (nstages > 1) issue_stage(wave_lds + STAGE_B);

  // lane's operand = 8 consecutive channels (one 16-byte chunk) of row r32: chunk 2*kstep + hh, XOR-swizzled
  const int sw = (r32 >> 2) & 3;
  const unsigned char* a_frag = smem16 + (wr * TM + r32) * 64;
  const unsigned char* b_frag = smem16 + (BM + wc * TN + r32) * 64;
  const int offk[2] = {((0 + hh) ^ sw) * 16, ((2 + hh) ^ sw) * 16};

  auto compute_stage = [&](const unsigned char* Ab, const unsigned char* Bb) {
    frag af[KSUB][2][MT], bf[KSUB][2][NT];
#pragma unroll
    for (int u = 0; u < KSUB; ++u)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
          af[u][ks][i] = *reinterpret_cast<const frag*>(Ab + u * SUB_B + i * 32 * 64 + offk[ks]);
#pragma unroll
        for (int j = 0; j < NT; ++j)
          bf[u][ks][j] = *reinterpret_cast<const frag*>(Bb + u * SUB_B + j * 32 * 64 + offk[ks]);
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < KSUB; ++u)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            if (ABL != 2) acc[i][j] = H16<F16>::mfma(af[u][ks][i], bf[u][ks][j], acc[i][j]);
            else acc[i][j][0] += (float)af[u][ks][i][0] + (float)bf[u][ks][j][0];
  };

  // lgkmcnt(0) in front of every barrier: the compiler is free to sink the tail of a stage's MFMAs - and the waits of the
  // ds_reads that feed them - below the next barrier; a wave must not signal "done reading stage s-1" with reads in flight,
  // because the other waves refill that slot right after the barrier.
  auto step = [&](auto slot_c) {
    constexpr int SLOT = decltype(slot_c)::value;
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPW * KSUB) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_stage(wave_lds + ((SLOT + 2) % NST) * STAGE_B);
    compute_stage(a_frag + SLOT * STAGE_B, b_frag + SLOT * STAGE_B);
  };
  int s = 0;
  for (; s + 3 <= nstages - 2; s += 3) {
    step(std::integral_constant<int, 0>{});
    step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{});
  }
  int slot = 0;
  for (; s < nstages; ++s) {
    if (s + 1 < nstages)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPW * KSUB) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + 2 < nstages) issue_stage(wave_lds + (slot == 0 ? NST - 1 : slot - 1) * STAGE_B);
    compute_stage(a_frag + slot * STAGE_B, b_frag + slot * STAGE_B);
    slot = slot == NST - 1 ? 0 : slot + 1;
  }

  // ---- epilogue: lane r32 = output channel, accumulator element e = pixel row --------------------
  if (p.splitk > 1) {
    float* slab = p.partial + (long long)sid * p.M * p.cout;
    auto slab_out = [&](auto ic, auto jc) {
      constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
      const int co = n0 + wc * TN + j * 32 + r32;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wr * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        if (co < p.cout && m < p.M) slab[(long long)m * p.cout + co] = acc[i][j][e];
      }
    };
    static_for([&](auto jc) { static_for([&](auto ic) { slab_out(ic, jc); }, std::make_integer_sequence<int, MT>{}); },
               std::make_integer_sequence<int, NT>{});
    return;
  }
  const float slope = p.act == ME_ACT_LEAKY ? 0.1f : 1.0f;
  // Vector epilogue (the common case: full tile, bf16 output, no upsample): the MFMA result has lane = output channel,
  // register = pixel row, so a direct store is 16 two-byte stores (+ 16 two-byte residual loads) per 32x32 block - at
  // bf16 matrix rates that costs more than the whole K loop of the short-K layers.  Each wave instead passes its 32x32
  // blocks through a private 32 x 36-float LDS patch (the stage buffers are dead by now): fp32 values after the affine +
  // activation go in channel-major, come back as 8 consecutive channels of one pixel per lane, the residual is one 16-byte
  // load, the result one 16-byte store (64 lanes = 16 pixels x 64 contiguous bytes).  Same arithmetic, same single RNE.
  if (p.vec_epi && m0 + BM <= p.M && n0 + BN <= p.cout) {
    constexpr int TP = 36;
    __syncthreads();
    float* tbuf = reinterpret_cast<float*>(smem16) + wave * (32 * TP);
    const int prow = lane >> 2, c8 = (lane & 3) * 8;
    unsigned short* __restrict__ yb = reinterpret_cast<unsigned short*>(p.y);
    const unsigned short* __restrict__ rb = reinterpret_cast<const unsigned short*>(p.res);
    auto block_out = [&](auto ic, auto jc) {
      constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
      const int cb = n0 + wc * TN + j * 32;
      const float sc = p.scale[cb + r32], sh = p.shift[cb + r32];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v = acc[i][j][e] * sc + sh;
        tbuf[((e & 3) + 8 * (e >> 2) + 4 * hh) * TP + r32] = fmaxf(v, v * slope);  // slope 0.1: leaky; 1: linear
      }
      // (no wait between the transpose's writes and reads: the LDS executes one wave's operations in order, and the
      // compiler orders them by the aliasing pointers - the explicit s_waitcnt pairs of round 1 serialised every block)
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int row = pass * 16 + prow;
        const float4 lo = *reinterpret_cast<const float4*>(tbuf + row * TP + c8);
        const float4 hi = *reinterpret_cast<const float4*>(tbuf + row * TP + c8 + 4);
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const long long m = m0 + wr * TM + i * 32 + row;
        if (rb) {
          const uint4 r = *reinterpret_cast<const uint4*>(rb + m * p.res_pitch + cb + c8);
          const unsigned rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            v[2 * k] += H16<F16>::from(rr[k] & 0xffffu);
            v[2 * k + 1] += H16<F16>::from(rr[k] >> 16);
          }
        }
        uint4 o;
        o.x = pack2<F16>(v[0], v[1]);
        o.y = pack2<F16>(v[2], v[3]);
        o.z = pack2<F16>(v[4], v[5]);
        o.w = pack2<F16>(v[6], v[7]);
        me::store16(yb + m * p.y_pitch + cb + c8, o, p.store_mode);
      }
    };
    static_for([&](auto jc) { static_for([&](auto ic) { block_out(ic, jc); }, std::make_integer_sequence<int, MT>{}); },
               std::make_integer_sequence<int, NT>{});
    return;
  }
  // (i, j) are expanded by static_for (a 128-element pragma-unrolled nest is refused as "too large" for the 128x64 and
  // 128x128 wave tiles, and a rolled loop would put the accumulators in scratch).  Fast path: plain store, leaky / linear
  // activation (slope 0.1 / 1.0), residual and output type hoisted out of the element loop.
  const bool fast = p.ups == 1 && p.act != ME_ACT_SIGMOID;
  auto tile_out = [&](auto ic, auto jc) {
    constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
    const int co = n0 + wc * TN + j * 32 + r32;
    if (co >= p.cout) return;
    const float sc = p.scale[co], sh = p.shift[co];
    const int mb = m0 + wr * TM + i * 32 + 4 * hh;
    if (fast) {
      auto run = [&](auto f32c, auto resc) {
        constexpr bool F32 = decltype(f32c)::value, RES = decltype(resc)::value;
        using T = std::conditional_t<F32, float, unsigned short>;
        T* __restrict__ y = reinterpret_cast<T*>(p.y) + co;
        const T* __restrict__ r = reinterpret_cast<const T*>(p.res) + co;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = mb + (e & 3) + 8 * (e >> 2);
          if (m >= p.M) continue;
          float v = acc[i][j][e] * sc + sh;
          v = v > 0.f ? v : v * slope;
          if constexpr (RES) {
            if constexpr (F32) v += r[(long long)m * p.res_pitch];
            else v += H16<F16>::from(r[(long long)m * p.res_pitch]);
          }
          if constexpr (F32) y[(long long)m * p.y_pitch] = v;
          else y[(long long)m * p.y_pitch] = H16<F16>::to(v);
        }
      };
      if (p.y_f32) {
        if (p.res) run(std::true_type{}, std::true_type{});
        else run(std::true_type{}, std::false_type{});
      } else {
        if (p.res) run(std::false_type{}, std::true_type{});
        else run(std::false_type{}, std::false_type{});
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = mb + (e & 3) + 8 * (e >> 2);
        if (m < p.M) store_out<F16>(p, m, co, act16(acc[i][j][e] * sc + sh, p.act), hw);
      }
    }
  };
  static_for([&](auto jc) { static_for([&](auto ic) { tile_out(ic, jc); }, std::make_integer_sequence<int, MT>{}); },
             std::make_integer_sequence<int, NT>{});
}

template <int F16>
__global__ __launch_bounds__(256) void conv_splitk_reduce_h16(Conv16P p) {
  const long long total = (long long)p.M * p.cout;
  const int hw = p.ho * p.wo;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int co = (int)(idx % p.cout);
    const int m = (int)(idx / p.cout);
    float a = p.partial[idx];
    for (int k = 1; k < p.splitk; ++k) a += p.partial[(long long)k * total + idx];
    store_out<F16>(p, m, co, act16(a * p.scale[co] + p.shift[co], p.act), hw);
  }
}

// ---------------------------------------------------------------------------------------------
// stem (cin == 3): fp32 frames (NCHW as the caller hands them, or NHWC) and fp32 weights in, bf16 NHWC out.
// Same arithmetic and k order as conv_stem3_f32 (conv.hip); one thread = one pixel x CT output channels.
// ---------------------------------------------------------------------------------------------
template <int CT, int F16>
__global__ __launch_bounds__(256) void conv_stem3_h16(Conv16P p) {
  constexpr int K = 27;
  const int m = blockIdx.x * 256 + threadIdx.x;
  const int co0 = blockIdx.y * CT;
  if (m >= p.M) return;
  const int hw = p.ho * p.wo;
  const int nimg = m / hw;
  const int rem = m - nimg * hw;
  const int oy = rem / p.wo, ox = rem - oy * p.wo;
  const float* xf = reinterpret_cast<const float*>(p.x);
  float xin[K];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
      const bool ok = ((unsigned)iy < (unsigned)p.h) && ((unsigned)ix < (unsigned)p.w);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float v = 0.f;
        if (ok)
          v = p.x_nchw ? xf[(((long long)nimg * 3 + c) * p.h + iy) * p.w + ix]
                       : xf[((long long)(nimg * p.h + iy) * p.w + ix) * p.x_pitch + c];
        xin[(ky * 3 + kx) * 3 + c] = H16<F16>::from(H16<F16>::to(v));  // the frame is rounded to the storage type
      }
    }
  const float* __restrict__ wg = reinterpret_cast<const float*>(p.wgt) + (long long)co0 * K;  // uniform -> s_load
  unsigned packed[CT / 2];
#pragma unroll
  for (int j = 0; j < CT; j += 2) {
    float v[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) a = fmaf(xin[k], wg[(j + t) * K + k], a);  // (the weights come pre-rounded from the host)
      v[t] = act16(a * p.scale[co0 + j + t] + p.shift[co0 + j + t], p.act);
    }
    packed[j / 2] = pack2<F16>(v[0], v[1]);
  }
  unsigned short* yrow = reinterpret_cast<unsigned short*>(p.y) + (long long)m * p.y_pitch + co0;
#pragma unroll
  for (int j = 0; j < CT / 2; j += 4)
    *reinterpret_cast<uint4*>(yrow + 2 * j) = make_uint4(packed[j], packed[j + 1], packed[j + 2], packed[j + 3]);
}

// ---------------------------------------------------------------------------------------------
// HBM-bound NHWC helpers on bf16: 8 channels (16 bytes) per lane, channel-fastest
// ---------------------------------------------------------------------------------------------
constexpr int kThreads = 256;
inline unsigned grid_for(long long work) {
  long long b = (work + kThreads - 1) / kThreads;
  const long long cap = 256ll * 32;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

struct Pool16P {
  const unsigned short* x;
  unsigned short* y;
  long long x_pitch, y_pitch;
  int n, h, w, c, size, stride, pad, zero_ext, ho, wo;
};

template <int F16>
__device__ __forceinline__ unsigned max_pair(unsigned a, unsigned b) {  // two packed 16-bit floats, fmaxf per half (exact)
  const float lo = fmaxf(H16<F16>::from(a & 0xffffu), H16<F16>::from(b & 0xffffu));
  const float hi = fmaxf(H16<F16>::from(a >> 16), H16<F16>::from(b >> 16));
  return pack2<F16>(lo, hi);
}

template <int F16>
__global__ __launch_bounds__(kThreads) void maxpool_h16_kernel(Pool16P d) {
  const int cv = d.c / 8;
  const long long total = (long long)d.n * d.ho * d.wo * cv;
  const unsigned ninf = F16 ? 0xfc00fc00u : 0xff80ff80u;  // (-inf, -inf)
  for (long long idx = (long long)blockIdx.x * kThreads + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * kThreads) {
    const int c = (int)(idx % cv) * 8;
    long long pix = idx / cv;
    const int ox = (int)(pix % d.wo);
    pix /= d.wo;
    const int oy = (int)(pix % d.ho);
    const int nimg = (int)(pix / d.ho);
    uint4 best = make_uint4(ninf, ninf, ninf, ninf);
    for (int ky = 0; ky < d.size; ++ky) {
      const int iy = oy * d.stride - d.pad + ky;
      for (int kx = 0; kx < d.size; ++kx) {
        const int ix = ox * d.stride - d.pad + kx;
        if ((unsigned)iy < (unsigned)d.h && (unsigned)ix < (unsigned)d.w) {
          const uint4 t = *reinterpret_cast<const uint4*>(d.x + ((long long)(nimg * d.h + iy) * d.w + ix) * d.x_pitch + c);
          best.x = max_pair<F16>(best.x, t.x);
          best.y = max_pair<F16>(best.y, t.y);
          best.z = max_pair<F16>(best.z, t.z);
          best.w = max_pair<F16>(best.w, t.w);
        } else if (d.zero_ext && iy >= 0 && ix >= 0 && iy <= d.h && ix <= d.w) {
          best.x = max_pair<F16>(best.x, 0u);  // ZeroPad2d((0,1,0,1)): zeros take part in the max (quirk q16)
          best.y = max_pair<F16>(best.y, 0u);
          best.z = max_pair<F16>(best.z, 0u);
          best.w = max_pair<F16>(best.w, 0u);
        }
      }
    }
    *reinterpret_cast<uint4*>(d.y + ((long long)(nimg * d.ho + oy) * d.wo + ox) * d.y_pitch + c) = best;
  }
}

__global__ __launch_bounds__(kThreads) void upsample_h16_kernel(const unsigned short* x, long long xp, unsigned short* y,
                                                                 long long yp, int n, int h, int w, int c, int f) {
  const int cv = c / 8;
  const int ho = h * f, wo = w * f;
  const long long total = (long long)n * ho * wo * cv;
  for (long long idx = (long long)blockIdx.x * kThreads + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * kThreads) {
    const int cc = (int)(idx % cv) * 8;
    long long pix = idx / cv;
    const int ox = (int)(pix % wo);
    pix /= wo;
    const int oy = (int)(pix % ho);
    const int nimg = (int)(pix / ho);
    *reinterpret_cast<uint4*>(y + ((long long)(nimg * ho + oy) * wo + ox) * yp + cc) =
        *reinterpret_cast<const uint4*>(x + ((long long)(nimg * h + oy / f) * w + ox / f) * xp + cc);
  }
}

template <int F16>
__device__ __forceinline__ unsigned add_pair(unsigned a, unsigned b) {
  return pack2<F16>(H16<F16>::from(a & 0xffffu) + H16<F16>::from(b & 0xffffu), H16<F16>::from(a >> 16) + H16<F16>::from(b >> 16));
}

template <bool ADD, int F16>
__global__ __launch_bounds__(kThreads) void addcopy_h16_kernel(const unsigned short* a, long long ap,
                                                                const unsigned short* b, long long bp, unsigned short* y,
                                                                long long yp, long long pixels, int c) {
  const int cv = c / 8;
  const long long total = pixels * cv;
  for (long long idx = (long long)blockIdx.x * kThreads + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * kThreads) {
    const int cc = (int)(idx % cv) * 8;
    const long long pix = idx / cv;
    uint4 u = *reinterpret_cast<const uint4*>(a + pix * ap + cc);
    if (ADD) {
      const uint4 t = *reinterpret_cast<const uint4*>(b + pix * bp + cc);
      u.x = add_pair<F16>(u.x, t.x);
      u.y = add_pair<F16>(u.y, t.y);
      u.z = add_pair<F16>(u.z, t.z);
      u.w = add_pair<F16>(u.w, t.w);
    }
    *reinterpret_cast<uint4*>(y + pix * yp + cc) = u;
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
constexpr int kMaxSplit16 = 16;

bool addressable16(const Conv16P& p, int bm) {
  const long long hw = (long long)p.ho * p.wo;
  const long long span_imgs = (bm - 1) / hw + 2;
  const long long img_bytes = (long long)p.h * p.w * p.x_pitch * 2;
  const long long tap_bytes = ((long long)p.ks * p.w + p.ks) * p.x_pitch * 2 + (long long)p.cin * 2;
  const long long a_max = span_imgs * img_bytes + 2 * tap_bytes;
  const long long b_max = 256ll * p.ktot * 2 + (long long)p.ktot * 2;
  return a_max < (1ll << 31) && b_max < (1ll << 31);
}

template <int BM, int BN, int WR, int WC, int KSUB, int MINW = 1, int ABL = 0, int F16 = 0, int MASKED = 0>
int launch16(Conv16P& p, hipStream_t stream) {
  static_assert(BM <= 256, "addressable16 / the descriptor window assume tiles of at most 256 rows");
  ME_REQUIRE(addressable16(p, BM), ME_E_TOOBIG,
             "me_conv2d_h16: one tile's input window exceeds the 2 GiB buffer-descriptor range");
  p.cs = p.cin / (32 * KSUB);
  p.stages = p.ks * p.ks * p.cs;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.cout + BN - 1) / BN;
  if (p.splitk > p.stages) p.splitk = p.stages;
  p.sps = (p.stages + p.splitk - 1) / p.splitk;
  while (p.splitk > 1 && (p.splitk - 1) * p.sps >= p.stages) --p.splitk;
  constexpr int NW = WR * WC;
  constexpr int LPW = ((BM + BN) / 16 + NW - 1) / NW;
  const size_t lds = (size_t)3 * KSUB * LPW * NW * 1024;
  if constexpr (MASKED) {
    ME_REQUIRE(p.splitk == 1 && p.mask_cols % BN == 0, ME_E_BADARG,
               "me_conv2d_h16: tap masks need whole tiles (split_k <= 1) and a tile width (%d) that divides tap_mask_cols", BN);
  }
  auto kern = conv_igemm_buf_h16<BM, BN, WR, WC, KSUB, MINW, ABL, F16, MASKED>;
  if (lds > 64 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
      attr_set = true;
    }
  }
  const long long blocks = (long long)p.tiles_m * p.tiles_n;
  ME_REQUIRE(blocks < (1ll << 31), ME_E_TOOBIG, "me_conv2d_h16: grid too large");
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)p.splitk), dim3(64 * NW), lds, stream, p);
  int rc = me::check_launch("conv_igemm_buf_h16");
  if (rc || p.splitk == 1) return rc;
  long long rb = ((long long)p.M * p.cout + 255) / 256;
  if (rb > 256 * 16) rb = 256 * 16;
  hipLaunchKernelGGL(conv_splitk_reduce_h16<F16>, dim3((unsigned)rb), dim3(256), 0, stream, p);
  return me::check_launch("conv_splitk_reduce_h16");
}

int fill16(const me_conv16_desc* d, Conv16P& p) {
  p.store_mode = me::store_mode();
  ME_REQUIRE(d != nullptr, ME_E_NULLPTR, "me_conv2d_h16: null descriptor");
  ME_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0, ME_E_BADARG,
             "me_conv2d_h16: non-positive dimension");
  ME_REQUIRE(d->ksize >= 1 && d->stride >= 1 && d->pad >= 0, ME_E_BADARG, "me_conv2d_h16: bad ksize/stride/pad");
  const int ho = (d->h + 2 * d->pad - d->ksize) / d->stride + 1;
  const int wo = (d->w + 2 * d->pad - d->ksize) / d->stride + 1;
  ME_REQUIRE(ho == d->ho && wo == d->wo, ME_E_BADARG, "me_conv2d_h16: ho/wo (%d,%d) != derived (%d,%d)", d->ho, d->wo,
             ho, wo);
  ME_REQUIRE((long long)d->n * d->ho * d->wo < (1ll << 31), ME_E_TOOBIG, "me_conv2d_h16: too many output pixels");
  ME_REQUIRE(d->half_type == 0 || d->half_type == 1, ME_E_BADARG, "me_conv2d_h16: half_type must be 0 (bf16) or 1 (f16)");
  p.x = reinterpret_cast<const unsigned short*>(d->x);
  p.wgt = reinterpret_cast<const unsigned short*>(d->wgt);
  p.wgt_tiled = reinterpret_cast<const unsigned short*>(d->wgt_tiled);
  p.scale = d->scale; p.shift = d->shift; p.res = d->res; p.y = d->y;
  p.x_pitch = d->x_pitch; p.res_pitch = d->res_pitch; p.y_pitch = d->y_pitch;
  p.n = d->n; p.h = d->h; p.w = d->w; p.cin = d->cin; p.cout = d->cout; p.ks = d->ksize;
  p.stride = d->stride; p.pad = d->pad; p.ho = d->ho; p.wo = d->wo; p.act = d->act; p.ups = d->upsample;
  p.y_f32 = d->y_f32; p.x_nchw = d->x_nchw; p.f16 = d->half_type;
  p.M = d->n * d->ho * d->wo;
  p.ktot = d->ksize * d->ksize * d->cin;
  magic16((unsigned)(d->ho * d->wo), &p.hw_m, &p.hw_s);
  magic16((unsigned)d->wo, &p.wo_m, &p.wo_s);
  p.mask_cols = d->tap_mask_cols > 0 ? d->tap_mask_cols : 0;
  for (int i = 0; i < 4; ++i) p.tapmask[i] = d->tap_mask[i];
  p.cs = p.stages = p.tiles_m = p.tiles_n = 0;
  p.partial = nullptr;
  p.partial_bytes = 0;
  p.splitk = 1;
  p.sps = 0;
  p.vec_epi = !d->y_f32 && d->upsample == 1 && d->act != ME_ACT_SIGMOID && d->y_pitch % 8 == 0 && me::aligned16(d->y) &&
              (!d->res || (d->res_pitch % 8 == 0 && me::aligned16(d->res)));
  return 0;
}

// default (tile, split) when the caller does not force one: the engine's autotuner measures the candidates; this is
// the cold-start guess - the largest tile that still gives every CU >= 2 workgroups, split-K below that.
void plan16(const Conv16P& p, int max_split, int* tile, int* split) {
  static const int bm[] = {0, 128, 128, 64, 256}, bn[] = {0, 128, 64, 64, 128};
  int best = 3;
  for (int t : {4, 1, 2, 3}) {
    const long long blocks = (long long)((p.M + bm[t] - 1) / bm[t]) * ((p.cout + bn[t] - 1) / bn[t]);
    if (blocks >= 512) {
      best = t;
      break;
    }
  }
  const long long blocks = (long long)((p.M + bm[best] - 1) / bm[best]) * ((p.cout + bn[best] - 1) / bn[best]);
  int s = 1;
  const int stages = p.ks * p.ks * (p.cin / 32);
  while (s < max_split && blocks * s < 512 && stages / (s * 2) >= 8) s *= 2;
  *tile = best;
  *split = s;
}

}  // namespace

extern "C" {

int64_t me_conv2d_h16_workspace_bytes(const me_conv16_desc* d) {
  Conv16P p;
  if (!d || fill16(d, p) != 0 || d->cin <= 4) return 0;
  int tile, split;
  if (d->tile >= 100) return me16::p8_workspace_bytes(p, d->tile, d->split_k);  // patch tiles: split only when asked to
  if (d->tile == 40 || d->tile == 41) return 0;  // the K split lives inside the workgroup: no slabs
  plan16(p, d->split_k > 0 ? d->split_k : kMaxSplit16, &tile, &split);
  if (d->split_k > 0) split = d->split_k;
  return split > 1 ? (int64_t)split * p.M * p.cout * (int64_t)sizeof(float) : 0;
}

int me_conv2d_h16(const me_conv16_desc* d, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  Conv16P p;
  int rc = fill16(d, p);
  if (rc) return rc;
  ME_REQUIRE(d->x && d->wgt && d->scale && d->shift && d->y, ME_E_NULLPTR, "me_conv2d_h16: null tensor pointer");
  ME_REQUIRE(d->upsample == 1 || d->upsample == 2, ME_E_BADARG, "me_conv2d_h16: upsample must be 1 or 2");
  ME_REQUIRE(d->act >= 0 && d->act <= 2, ME_E_BADARG, "me_conv2d_h16: unknown activation %d", d->act);
  ME_REQUIRE(d->y_pitch >= d->cout, ME_E_BADARG, "me_conv2d_h16: y_pitch < cout");

  if (d->cin <= 4) {  // stem: fp32 frames + fp32 weights -> bf16
    ME_REQUIRE(d->cin == 3 && d->ksize == 3, ME_E_BADARG, "me_conv2d_h16: the stem kernel needs cin 3, ksize 3");
    ME_REQUIRE(d->cout % 16 == 0 && d->y_pitch % 8 == 0 && !d->y_f32, ME_E_BADARG,
               "me_conv2d_h16: stem needs cout %% 16 == 0, y_pitch %% 8 == 0, bf16 output");
    ME_REQUIRE(d->res == nullptr && d->upsample == 1, ME_E_BADARG, "me_conv2d_h16: stem has no residual/upsample epilogue");
    ME_REQUIRE(d->x_nchw || d->x_pitch >= d->cin, ME_E_BADARG, "me_conv2d_h16: x_pitch < cin");
    ME_REQUIRE(me::aligned16(d->y), ME_E_ALIGN, "me_conv2d_h16: y not 16-byte aligned");
    if (d->tile != 1 && me16::stem_mfma_eligible(p)) return me16::launch_stem_mfma(p, stream);  // tile 1 forces the VALU stem
    const unsigned mb = (unsigned)((p.M + 255) / 256);
    if (d->cout % 32 == 0) {
      if (p.f16) hipLaunchKernelGGL((conv_stem3_h16<32, 1>), dim3(mb, d->cout / 32), dim3(256), 0, stream, p);
      else hipLaunchKernelGGL((conv_stem3_h16<32, 0>), dim3(mb, d->cout / 32), dim3(256), 0, stream, p);
    } else {
      if (p.f16) hipLaunchKernelGGL((conv_stem3_h16<16, 1>), dim3(mb, d->cout / 16), dim3(256), 0, stream, p);
      else hipLaunchKernelGGL((conv_stem3_h16<16, 0>), dim3(mb, d->cout / 16), dim3(256), 0, stream, p);
    }
    return me::check_launch("conv_stem3_h16");
  }

  ME_REQUIRE(!d->x_nchw, ME_E_BADARG, "me_conv2d_h16: NCHW input only for the stem");
  ME_REQUIRE(d->cin % 32 == 0, ME_E_BADARG, "me_conv2d_h16: cin %% 32 != 0 (cin=%d): pad the channels", d->cin);
  ME_REQUIRE(d->x_pitch >= d->cin && d->x_pitch % 8 == 0, ME_E_ALIGN, "me_conv2d_h16: x_pitch must be >= cin, %% 8");
  ME_REQUIRE(me::aligned16(d->x) && me::aligned16(d->wgt), ME_E_ALIGN, "me_conv2d_h16: x / wgt not 16-byte aligned");
  ME_REQUIRE(!d->res || d->res_pitch >= d->cout, ME_E_BADARG, "me_conv2d_h16: res_pitch < cout");
  ME_REQUIRE(d->split_k >= 0 && d->split_k <= 64, ME_E_BADARG, "me_conv2d_h16: split_k out of range");
  ME_REQUIRE(d->ksize * d->ksize <= 32, ME_E_TOOBIG, "me_conv2d_h16: filters with more than 32 taps are not supported");

  const long long slab = (long long)p.M * p.cout * (long long)sizeof(float);
  int max_split = 1;
  if (d->workspace && d->workspace_bytes >= 2 * slab) {
    const long long fit = d->workspace_bytes / slab;
    max_split = fit < kMaxSplit16 ? (int)fit : kMaxSplit16;
  }
  ME_REQUIRE(p.mask_cols == 0 || d->tile == 1 || d->tile == 2 || d->tile == 3 || d->tile == 11 || d->tile == 12 || d->tile == 13,
             ME_E_BADARG, "me_conv2d_h16: tap masks need an explicit per-tap tile id 1 / 2 / 3 / 11 / 12 / 13 (got %d)", d->tile);
  if (d->tile >= 100) {  // patch-resident tiles: K split only on request (compact slabs, checked by the launcher)
    p.splitk = d->split_k > 1 ? d->split_k : 1;
    p.partial = reinterpret_cast<float*>(d->workspace);
    p.partial_bytes = d->workspace ? d->workspace_bytes : 0;
    return me16::launch_p8_tile(p, d->tile, stream);
  }
  int tile = d->tile, split = 1;
  if (tile == 0 || d->split_k == 0) {
    int t0, s0;
    plan16(p, max_split, &t0, &s0);
    if (tile == 0) tile = t0;
    split = s0;
  }
  if (d->split_k > 0) {
    ME_REQUIRE(d->split_k == 1 || (d->workspace && d->workspace_bytes >= d->split_k * slab), ME_E_BADARG,
               "me_conv2d_h16: split_k=%d needs a workspace of %lld bytes", d->split_k, d->split_k * slab);
    split = d->split_k;
  }
  p.splitk = split;
  p.partial = reinterpret_cast<float*>(d->workspace);
  if (tile >= 100) return me16::launch_p8_tile(p, tile, stream);
  if (tile == 40 || tile == 41) return me16::launch_kw(p, tile, stream);
  if (tile == 50) return me16::launch_ws1x1(p, stream);
  if (tile == 60) return me16::launch_ws3x3(p, stream);
  const bool k2 = d->cin % 64 == 0;  // two 32-channel sub-stages per pipeline stage when the channel count allows
  if (p.mask_cols > 0) {  // column-class tap masks (ABI 13): the masked instances of the per-tap tiles
    const int taps = d->ksize * d->ksize;
    ME_REQUIRE(d->cout % p.mask_cols == 0 && d->cout / p.mask_cols <= 4, ME_E_BADARG,
               "me_conv2d_h16: tap_mask_cols %d must cut cout %d into at most 4 classes", p.mask_cols, d->cout);
    for (int i = 0; i < d->cout / p.mask_cols; ++i)
      ME_REQUIRE((d->tap_mask[i] & ((1u << taps) - 1u)) != 0, ME_E_BADARG, "me_conv2d_h16: tap_mask[%d] selects no tap", i);
    ME_REQUIRE(d->split_k <= 1, ME_E_BADARG, "me_conv2d_h16: tap masks and a K split do not combine");
    p.splitk = 1;
    const bool two = (tile == 1 || tile == 2 || tile == 3) && k2;
#define ME_MASKED16(BM_, BN_) \
  (p.f16 ? (two ? launch16<BM_, BN_, 2, 2, 2, 1, 0, 1, 1>(p, stream) : launch16<BM_, BN_, 2, 2, 1, 1, 0, 1, 1>(p, stream)) \
         : (two ? launch16<BM_, BN_, 2, 2, 2, 1, 0, 0, 1>(p, stream) : launch16<BM_, BN_, 2, 2, 1, 1, 0, 0, 1>(p, stream)))
    switch (tile) {
      case 1: case 11: return ME_MASKED16(128, 128);
      case 2: case 12: return ME_MASKED16(128, 64);
      case 3: case 13: return ME_MASKED16(64, 64);
      default: ME_REQUIRE(false, ME_E_BADARG, "me_conv2d_h16: tap masks need tile 1 / 2 / 3 / 11 / 12 / 13 (got %d)", tile);
    }
#undef ME_MASKED16
  }
  if (p.f16) {
    switch (tile) {
      case 1: return k2 ? launch16<128, 128, 2, 2, 2, 1, 0, 1>(p, stream) : launch16<128, 128, 2, 2, 1, 1, 0, 1>(p, stream);
      case 2: return k2 ? launch16<128, 64, 2, 2, 2, 1, 0, 1>(p, stream) : launch16<128, 64, 2, 2, 1, 1, 0, 1>(p, stream);
      case 3: return k2 ? launch16<64, 64, 2, 2, 2, 1, 0, 1>(p, stream) : launch16<64, 64, 2, 2, 1, 1, 0, 1>(p, stream);
      case 4: return k2 ? launch16<256, 128, 4, 2, 2, 2, 0, 1>(p, stream) : launch16<256, 128, 4, 2, 1, 2, 0, 1>(p, stream);
      case 11: return launch16<128, 128, 2, 2, 1, 1, 0, 1>(p, stream);
      case 12: return launch16<128, 64, 2, 2, 1, 1, 0, 1>(p, stream);
      case 13: return launch16<64, 64, 2, 2, 1, 1, 0, 1>(p, stream);
      case 14: return launch16<256, 128, 4, 2, 1, 2, 0, 1>(p, stream);
      case 5: case 15: return launch16<192, 128, 2, 2, 1, 2, 0, 1>(p, stream);
      default: ME_REQUIRE(false, ME_E_BADARG, "me_conv2d_h16: unknown tile id %d (half_type 1)", tile);
    }
  }
  switch (tile) {
    case 1: return k2 ? launch16<128, 128, 2, 2, 2>(p, stream) : launch16<128, 128, 2, 2, 1>(p, stream);
    case 2: return k2 ? launch16<128, 64, 2, 2, 2>(p, stream) : launch16<128, 64, 2, 2, 1>(p, stream);
    case 3: return k2 ? launch16<64, 64, 2, 2, 2>(p, stream) : launch16<64, 64, 2, 2, 1>(p, stream);
    case 4: return k2 ? launch16<256, 128, 4, 2, 2, 2>(p, stream) : launch16<256, 128, 4, 2, 1, 2>(p, stream);
    // forced ids (tuning): single sub-stage variants
    case 11: return launch16<128, 128, 2, 2, 1>(p, stream);
    case 12: return launch16<128, 64, 2, 2, 1>(p, stream);
    case 13: return launch16<64, 64, 2, 2, 1>(p, stream);
    case 14: return launch16<256, 128, 4, 2, 1, 2>(p, stream);
    // 192 x 128 (4 waves, wave tile 96 x 64): a row count between the 128- and 256-row tiles for the layers whose tile count
    // falls just above a multiple of the 512 resident workgroups (one sub-stage only: 61 KB of LDS, 2 workgroups per CU)
    case 5: case 15: return launch16<192, 128, 2, 2, 1, 2>(p, stream);
    case 71: return launch16<128, 128, 2, 2, 1, 1, 3>(p, stream);
    case 73: return launch16<64, 64, 2, 2, 1, 1, 3>(p, stream);
    case 74: return launch16<256, 128, 4, 2, 1, 2, 3>(p, stream);
    // ablations (wrong results): 8x = no memory traffic, 9x = no MFMAs
    case 81: return launch16<128, 128, 2, 2, 1, 1, 1>(p, stream);
    case 83: return launch16<64, 64, 2, 2, 1, 1, 1>(p, stream);
    case 84: return launch16<256, 128, 4, 2, 1, 2, 1>(p, stream);
    case 91: return launch16<128, 128, 2, 2, 1, 1, 2>(p, stream);
    case 93: return launch16<64, 64, 2, 2, 1, 1, 2>(p, stream);
    case 94: return launch16<256, 128, 4, 2, 1, 2, 2>(p, stream);
    default: ME_REQUIRE(false, ME_E_BADARG, "me_conv2d_h16: unknown tile id %d", tile);
  }
  return 0;
}

int me_maxpool_h16(const void* x, int64_t x_pitch, void* y, int64_t y_pitch, int32_t n, int32_t h, int32_t w, int32_t c,
                   int32_t size, int32_t stride, int32_t pad, int32_t zero_ext, int32_t ho, int32_t wo, int32_t half_type,
                   void* stream) {
  ME_REQUIRE(x && y, ME_E_NULLPTR, "me_maxpool_h16: null pointer");
  ME_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && size >= 1 && stride >= 1 && pad >= 0, ME_E_BADARG,
             "me_maxpool_h16: bad dimensions");
  ME_REQUIRE(c % 8 == 0 && x_pitch % 8 == 0 && y_pitch % 8 == 0 && x_pitch >= c && y_pitch >= c, ME_E_ALIGN,
             "me_maxpool_h16: channels and pitches must be multiples of 8");
  ME_REQUIRE(me::aligned16(x) && me::aligned16(y), ME_E_ALIGN, "me_maxpool_h16: pointers not 16-byte aligned");
  const int eh = (h + (zero_ext ? 1 : 0) + 2 * pad - size) / stride + 1, ew = (w + (zero_ext ? 1 : 0) + 2 * pad - size) / stride + 1;
  ME_REQUIRE(eh == ho && ew == wo, ME_E_BADARG, "me_maxpool_h16: ho/wo (%d,%d) != derived (%d,%d)", ho, wo, eh, ew);
  Pool16P d{reinterpret_cast<const unsigned short*>(x), reinterpret_cast<unsigned short*>(y), x_pitch, y_pitch, n, h, w, c,
            size, stride, pad, zero_ext, ho, wo};
  if (half_type)
    hipLaunchKernelGGL(maxpool_h16_kernel<1>, dim3(grid_for((long long)n * ho * wo * (c / 8))), dim3(kThreads), 0,
                       reinterpret_cast<hipStream_t>(stream), d);
  else
    hipLaunchKernelGGL(maxpool_h16_kernel<0>, dim3(grid_for((long long)n * ho * wo * (c / 8))), dim3(kThreads), 0,
                       reinterpret_cast<hipStream_t>(stream), d);
  return me::check_launch("maxpool_h16_kernel");
}

int me_upsample_h16(const void* x, int64_t x_pitch, void* y, int64_t y_pitch, int32_t n, int32_t h, int32_t w, int32_t c,
                     int32_t factor, void* stream) {
  ME_REQUIRE(x && y, ME_E_NULLPTR, "me_upsample_h16: null pointer");
  ME_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && factor >= 1, ME_E_BADARG, "me_upsample_h16: bad dimensions");
  ME_REQUIRE(c % 8 == 0 && x_pitch % 8 == 0 && y_pitch % 8 == 0 && me::aligned16(x) && me::aligned16(y), ME_E_ALIGN,
             "me_upsample_h16: channels / pitches %% 8, pointers 16-byte aligned");
  hipLaunchKernelGGL(upsample_h16_kernel, dim3(grid_for((long long)n * h * factor * w * factor * (c / 8))), dim3(kThreads),
                     0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const unsigned short*>(x), x_pitch,
                     reinterpret_cast<unsigned short*>(y), y_pitch, n, h, w, c, factor);
  return me::check_launch("upsample_h16_kernel");
}

int me_add_h16(const void* a, int64_t a_pitch, const void* b, int64_t b_pitch, void* y, int64_t y_pitch, int64_t pixels,
               int32_t c, int32_t half_type, void* stream) {
  ME_REQUIRE(a && b && y, ME_E_NULLPTR, "me_add_h16: null pointer");
  ME_REQUIRE(pixels > 0 && c > 0, ME_E_BADARG, "me_add_h16: bad dimensions");
  ME_REQUIRE(c % 8 == 0 && a_pitch % 8 == 0 && b_pitch % 8 == 0 && y_pitch % 8 == 0 && me::aligned16(a) &&
                 me::aligned16(b) && me::aligned16(y),
             ME_E_ALIGN, "me_add_h16: channels / pitches %% 8, pointers 16-byte aligned");
  auto* ap = reinterpret_cast<const unsigned short*>(a);
  auto* bp = reinterpret_cast<const unsigned short*>(b);
  auto* yp = reinterpret_cast<unsigned short*>(y);
  if (half_type)
    hipLaunchKernelGGL((addcopy_h16_kernel<true, 1>), dim3(grid_for(pixels * (c / 8))), dim3(kThreads), 0,
                       reinterpret_cast<hipStream_t>(stream), ap, a_pitch, bp, b_pitch, yp, y_pitch, pixels, c);
  else
    hipLaunchKernelGGL((addcopy_h16_kernel<true, 0>), dim3(grid_for(pixels * (c / 8))), dim3(kThreads), 0,
                       reinterpret_cast<hipStream_t>(stream), ap, a_pitch, bp, b_pitch, yp, y_pitch, pixels, c);
  return me::check_launch("addcopy_h16_kernel");
}

int me_copy_h16(const void* x, int64_t x_pitch, void* y, int64_t y_pitch, int64_t pixels, int32_t c, void* stream) {
  ME_REQUIRE(x && y, ME_E_NULLPTR, "me_copy_h16: null pointer");
  ME_REQUIRE(pixels > 0 && c > 0, ME_E_BADARG, "me_copy_h16: bad dimensions");
  ME_REQUIRE(c % 8 == 0 && x_pitch % 8 == 0 && y_pitch % 8 == 0 && me::aligned16(x) && me::aligned16(y), ME_E_ALIGN,
             "me_copy_h16: channels / pitches %% 8, pointers 16-byte aligned");
  hipLaunchKernelGGL((addcopy_h16_kernel<false, 0>), dim3(grid_for(pixels * (c / 8))), dim3(kThreads), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const unsigned short*>(x), x_pitch,
                     reinterpret_cast<const unsigned short*>(x), x_pitch, reinterpret_cast<unsigned short*>(y), y_pitch,
                     pixels, c);
  return me::check_launch("addcopy_h16_kernel");
}

}  // extern "C"
