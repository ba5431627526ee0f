// bneck_h16.hip - one launch for a Darknet bottleneck in the 16-bit storage modes (gfx950):
//     mid = leaky(bn1(conv1x1(x)))          cin -> cmid          (reference yolov3/models.py:22-41, a [convolutional] block)
//     y   = leaky(bn2(conv3x3(mid))) + x    cmid -> cout, pad 1  (the next [convolutional] block + the [shortcut], :258-260)
// The mid tensor never reaches HBM: a workgroup owns BM consecutive output positions of the 3x3 (all cout channels) and
//   phase 1  runs the 1x1 over those positions PLUS their halo (one padded row + one position on either side: rows =
//            BM + 2 * (Wp + 1)), x and W1 streamed through LDS in 64-byte channel chunks, fp32 accumulators -> bn1 + leaky
//            -> rounded to the storage type (the same rounding point the two-launch path has: mid is a stored activation)
//            -> LDS, in the patch layout of the patch-resident 3x3 kernel ([32-channel chunk][row][64 B], 16-byte pieces
//            XOR-swizzled by (row >> 2) & 3).  Pad positions of the padded-linear space and positions outside the batch
//            hold ZEROS (the 3x3 pads mid with zeros, not with leaky(bn1(0)));
//   phase 2  is the main loop of conv_p8_impl.h's register-pipelined variant (one barrier per (chunk, tap) stage, weight
//            slabs of BN rows x 64 B through a three-slot ring, fragments of stage s + 1 fetched behind the MFMAs of stage s)
//            with the patch already resident for EVERY chunk - no patch DMA at all;
//   epilogue bn2 + leaky + residual (x again: L2 / Infinity Cache hits) + 16-byte stores, as in conv_p8_impl.h.
// Phase 1 multiplies with the operand roles swapped (A = W1 rows, B = x rows): an accumulator lane then holds four
// consecutive mid channels of ONE position per register group, i.e. 8 bytes of a patch row -> ds_write_b64, no transpose.
// Index space, tile order over the XCDs, halo arithmetic: exactly conv_p8_h16.hip's padded-linear space (Wp = W + 1).
// Cost: the halo rows are recomputed by neighbouring tiles (rows / BM = 1.56 at 52 x 52 with BM = 192; the 1x1 is 10 % of the
// block's FLOPs); saved: the mid tensor's write and read, one launch, and the 1x1 kernel's HBM-bound pass over x.
#include <cstdlib>

#include "conv16_common.h"

namespace {
using namespace me_dma;

struct BnArgs {
  const void* x;
  const void* w1t;   // [cin / 32][cmid][32]
  const float* sc1;
  const float* sh1;
  const void* w2t;   // [9][cmid / 32][cout][32]
  const float* sc2;
  const float* sh2;
  const void* res;
  void* y;
  long long x_pitch, res_pitch, y_pitch;  // elements
  int n, h, w, cin, cmid, cout, act1, act2;
  int Wp, Ip, halo;
  long long Mp;
  unsigned ip_m, ip_s, wp_m, wp_s;
  int rows;        // BM + 2 * halo
  int tiles_m;
  int store_mode;
  int abl;         // tuning only (MILLIEYE_BNECK_ABL, wrong results): 1 = no phase-1 K loop, 2 = no phase-2 loop, 4 = no epilogue
};

__device__ __forceinline__ unsigned udiv_magic(unsigned n, unsigned m, unsigned s) { return (__umulhi(n, m) + n) >> s; }

__device__ __forceinline__ void dma1(unsigned v, u32x4 r, unsigned s, unsigned dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v], %[r], %[s] offen lds\n\t"
               "s_mov_b32 m0, %[k]"
               : [k] "=&s"(keep)
               : [d] "s"(dst), [r] "s"(r), [s] "s"(s), [v] "v"(v)
               : "memory", "scc");
}

using me::store16;

// WR x WC waves (8), wave tile 32 MT x 32 NT of the 3x3's BM x BN output tile (BN == cout);  CMC = cmid / 32 chunks;
// phase 1: the 8 waves as (8 / CMC) x CMC over [rows] x [cmid], P1MT row blocks of 32 per wave (ROWS_PAD = 32 P1MT 8 / CMC)
template <int F16, int WR, int WC, int MT, int NT, int CMC, int P1MT, int NBUF>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void bneck_kernel(BnArgs a) {
  using HT = H16<F16>;
  using frag = typename HT::v8;
  static_assert(WR * WC == 8, "8 waves");
  static_assert(CMC == 2 || CMC == 4 || CMC == 8, "cmid = 64, 128 or 256");
  constexpr int TM = 32 * MT, TN = 32 * NT, BM = TM * WR, BN = TN * WC;
  constexpr int CMID = 32 * CMC;
  constexpr int P1WC = CMC, P1WR = 8 / CMC;
  constexpr int ROWS_PAD = 32 * P1MT * P1WR;
  constexpr int LPB = BN / 16 / 8;                  // weight-slab DMA instructions per wave and stage
  static_assert(LPB >= 1, "BN >= 128");
  constexpr unsigned B_SLOT = BN * 64u;             // one (tap, chunk) slab of the 3x3's weights
  constexpr unsigned MID_BASE = 3u * B_SLOT;        // behind the three-slot ring
  constexpr unsigned MID_CH = ROWS_PAD * 64u;       // one 32-channel chunk of the mid patch
  // phase 1 staging: NBUF x buffers (ROWS_PAD rows x 64 B each) where the patch will be, NBUF W1 buffers behind the patch region
  static_assert(NBUF >= 2 && NBUF <= 4, "2..4 staging buffers");
  constexpr unsigned PATCH_BYTES = (CMC > NBUF ? CMC : NBUF) * MID_CH;
  constexpr unsigned XS = MID_BASE, WS = MID_BASE + PATCH_BYTES;
  constexpr unsigned W1_BYTES = CMID * 64u;
  constexpr int NXI = ROWS_PAD / 16;                // x-slab DMA instructions per chunk (16 rows each)
  constexpr int LPX = (NXI + 7) / 8;
  constexpr int NWI = CMID / 16;                    // W1-slab DMA instructions per chunk (<= 8 for cmid <= 128, else 2 per wave)
  constexpr int LPW1 = (NWI + 7) / 8;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC;
  const int pr = wave / P1WC, pc = wave % P1WC;
  const int r32 = lane & 31, hh = lane >> 5;
  const int swb = (r32 >> 2) & 3;

  int tile_m;
  {  // contiguous tile ranges per XCD (workgroup ids are dealt round-robin over the eight XCDs)
    const int nwg = a.tiles_m;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    tile_m = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int W = a.w, H = a.h;
  const long long q0 = (long long)tile_m * BM;
  const long long pq0 = q0 - a.halo;
  const int nb = pq0 > 0 ? (int)udiv_magic((unsigned)pq0, a.ip_m, a.ip_s) : 0;
  const u32x4 rsrc_x = make_rsrc(reinterpret_cast<const unsigned char*>(a.x) + (long long)nb * H * W * a.x_pitch * 2);
  const u32x4 rsrc_w1 = make_rsrc(a.w1t);
  const u32x4 rsrc_w2 = make_rsrc(a.w2t);

  // ---- per-lane DMA offsets
  unsigned v_x[LPX];
#pragma unroll
  for (int j = 0; j < LPX; ++j) {
    const int r = (wave + 8 * j) * 16 + (lane >> 2);
    const int qd = (lane & 3) ^ ((r >> 2) & 3);
    const long long pq = pq0 + r;
    unsigned off = kOobOffset;
    if (r < a.rows && pq >= 0 && pq < a.Mp) {
      const unsigned u = (unsigned)pq;
      const unsigned n = udiv_magic(u, a.ip_m, a.ip_s);
      const unsigned rem = u - n * (unsigned)a.Ip;
      const unsigned y = udiv_magic(rem, a.wp_m, a.wp_s);
      const unsigned x = rem - y * (unsigned)a.Wp;
      if (y < (unsigned)H && x < (unsigned)W)
        off = (unsigned)((((long long)(n - nb) * H + y) * W + x) * a.x_pitch * 2) + 16u * qd;
    }
    v_x[j] = off;
  }
  unsigned v_w1[LPW1];
#pragma unroll
  for (int j = 0; j < LPW1; ++j) {
    const int row = (wave + 8 * j) * 16 + (lane >> 2);
    const int qd = (lane & 3) ^ ((row >> 2) & 3);
    v_w1[j] = (unsigned)row * 64u + 16u * qd;
  }
  unsigned v_b[LPB];
#pragma unroll
  for (int j = 0; j < LPB; ++j) {
    const int row = (wave + 8 * j) * 16 + (lane >> 2);
    const int qd = (lane & 3) ^ ((row >> 2) & 3);
    v_b[j] = (unsigned)row * 64u + 16u * qd;
  }
  const unsigned wave_lds = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)wave * 1024u);

  auto issue_b = [&](int chunk, int tap, unsigned ring) {   // the 3x3's weight slab of stage (chunk, tap) into ring slot `ring`
    const unsigned soff = ((unsigned)tap * (unsigned)CMC + (unsigned)chunk) * (unsigned)BN * 64u;
#pragma unroll
    for (int j = 0; j < LPB; ++j) dma1(v_b[j], rsrc_w2, soff, wave_lds + ring * B_SLOT + (unsigned)j * 8192u);
  };
  auto issue1 = [&](int kc, int buf) {   // phase 1: chunk kc of x (ROWS_PAD rows) and of W1 (cmid rows)
    const unsigned xdst = wave_lds + XS + (unsigned)buf * MID_CH;
#pragma unroll
    for (int j = 0; j < LPX; ++j)
      if (wave + 8 * j < NXI) dma1(v_x[j], rsrc_x, (unsigned)kc * 64u, xdst + (unsigned)j * 8192u);
    const unsigned wdst = wave_lds + WS + (unsigned)buf * W1_BYTES;
#pragma unroll
    for (int j = 0; j < LPW1; ++j)
      if (wave + 8 * j < NWI) dma1(v_w1[j], rsrc_w1, (unsigned)kc * (CMID * 64u), wdst + (unsigned)j * 8192u);
  };
  // DMA instructions of THIS wave per phase-1 chunk (the waits below count them)
  int per_chunk = 0;
#pragma unroll
  for (int j = 0; j < LPX; ++j) per_chunk += (wave + 8 * j < NXI) ? 1 : 0;
#pragma unroll
  for (int j = 0; j < LPW1; ++j) per_chunk += (wave + 8 * j < NWI) ? 1 : 0;
  per_chunk = __builtin_amdgcn_readfirstlane(per_chunk);
  auto wait_vm = [&](int n) {   // s_waitcnt takes an immediate: a uniform switch
    switch (n) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
      case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
      case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
      case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
      case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
      case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
      case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
      case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
      case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
      case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
      case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
      case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // (more than the table holds: wait for everything)
    }
  };

  // the first three weight slabs of phase 2 ride in front of phase 1 (oldest loads of the wave: they retire first)
  issue_b(0, 0, 0);
  issue_b(0, 1, 1);
  issue_b(0, 2, 2);

  // ================================================================ phase 1: mid patch = leaky(bn1(x W1)) -> LDS
  {
    f32x16 acc1[P1MT];
#pragma unroll
    for (int i = 0; i < P1MT; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc1[i][e] = 0.f;
    // this lane's 16 mid channels of the column block pc: pc * 32 + 8 g + 4 hh + k  (g = register group, k = 0..3)
    float4 s1[4], t1[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      s1[g] = *reinterpret_cast<const float4*>(a.sc1 + pc * 32 + 8 * g + 4 * hh);
      t1[g] = *reinterpret_cast<const float4*>(a.sh1 + pc * 32 + 8 * g + 4 * hh);
    }
    const unsigned offk0 = (unsigned)(((0 + hh) ^ swb) * 16), offk1 = (unsigned)(((2 + hh) ^ swb) * 16);
    const int kcs = (a.abl & 1) ? 0 : a.cin >> 5;
    // NBUF - 1 chunks in flight beyond the one being multiplied; one barrier per chunk: behind it every wave has left chunk
    // kc - 1, whose buffer (kc - 1) % NBUF = (kc + NBUF - 1) % NBUF takes chunk kc + NBUF - 1
#pragma unroll
    for (int c = 0; c < NBUF - 1; ++c)
      if (c < kcs) issue1(c, c);
    int buf = 0;
    for (int kc = 0; kc < kcs; ++kc) {
      const int ahead = kcs - 1 - kc < NBUF - 2 ? kcs - 1 - kc : NBUF - 2;   // chunks requested behind chunk kc at this point
      wait_vm(ahead * per_chunk);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kc + NBUF - 1 < kcs) issue1(kc + NBUF - 1, buf == 0 ? NBUF - 1 : buf - 1);
      const unsigned char* Wb = smem + WS + (unsigned)buf * W1_BYTES + (pc * 32 + r32) * 64;
      const unsigned char* Xb = smem + XS + (unsigned)buf * MID_CH + (unsigned)(pr * P1MT * 32 + r32) * 64u;
      const frag wa0 = *reinterpret_cast<const frag*>(Wb + offk0);
      const frag wa1 = *reinterpret_cast<const frag*>(Wb + offk1);
#pragma unroll
      for (int i = 0; i < P1MT; ++i) {
        const frag xb0 = *reinterpret_cast<const frag*>(Xb + i * 2048 + offk0);
        const frag xb1 = *reinterpret_cast<const frag*>(Xb + i * 2048 + offk1);
        acc1[i] = HT::mfma(wa0, xb0, acc1[i]);
        acc1[i] = HT::mfma(wa1, xb1, acc1[i]);
      }
      buf = buf + 1 == NBUF ? 0 : buf + 1;
    }
    __syncthreads();   // every wave is done with the staging buffers: the patch goes over the x staging
    const float slope1 = a.act1 == ME_ACT_LEAKY ? 0.1f : 1.0f;
#pragma unroll
    for (int i = 0; i < P1MT; ++i) {
      const int R = (pr * P1MT + i) * 32 + r32;
      const long long pq = pq0 + R;
      bool valid = false;
      if (R < a.rows && pq >= 0 && pq < a.Mp) {
        const unsigned u = (unsigned)pq;
        const unsigned n = udiv_magic(u, a.ip_m, a.ip_s);
        const unsigned rem = u - n * (unsigned)a.Ip;
        const unsigned y = udiv_magic(rem, a.wp_m, a.wp_s);
        const unsigned x = rem - y * (unsigned)a.Wp;
        valid = y < (unsigned)H && x < (unsigned)W;
      }
      unsigned char* dst = smem + MID_BASE + (unsigned)pc * MID_CH + (unsigned)R * 64u + 8u * hh;
      const unsigned sw = (unsigned)(R >> 2) & 3u;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v0 = acc1[i][4 * g + 0] * s1[g].x + t1[g].x;
        float v1 = acc1[i][4 * g + 1] * s1[g].y + t1[g].y;
        float v2 = acc1[i][4 * g + 2] * s1[g].z + t1[g].z;
        float v3 = acc1[i][4 * g + 3] * s1[g].w + t1[g].w;
        v0 = fmaxf(v0, v0 * slope1);
        v1 = fmaxf(v1, v1 * slope1);
        v2 = fmaxf(v2, v2 * slope1);
        v3 = fmaxf(v3, v3 * slope1);
        uint2 o;
        o.x = valid ? pack2<F16>(v0, v1) : 0u;
        o.y = valid ? pack2<F16>(v2, v3) : 0u;
        *reinterpret_cast<uint2*>(dst + (((unsigned)g ^ sw) * 16u)) = o;
      }
    }
  }

  // ================================================================ phase 2: 3x3 over the resident patch
  int tapoff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) tapoff[t] = __builtin_amdgcn_readfirstlane((t / 3) * a.Wp + (t % 3));
  unsigned rowbase[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) rowbase[i] = (unsigned)(wr * TM + i * 32 + r32);
  const unsigned char* b_frag = smem + (wc * TN + r32) * 64;
  const unsigned b_offk[2] = {(unsigned)(((0 + hh) ^ swb) * 16), (unsigned)(((2 + hh) ^ swb) * 16)};

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  frag afr[2][2][MT], bfr[2][2][NT];
  auto load_frags = [&](auto setc, auto tc, int chunk) {
    constexpr int SET = decltype(setc)::value, T = decltype(tc)::value;
    const unsigned char* Ab = smem + MID_BASE + (unsigned)chunk * MID_CH;
    const unsigned char* Bb = b_frag + (T % 3) * B_SLOT;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      unsigned rb0 = rowbase[i];
      asm volatile("" : "+v"(rb0));
      const unsigned R = rb0 + (unsigned)tapoff[T];
      const unsigned o0 = (R << 6) + (((R >> 2) ^ (unsigned)hh) & 3u) * 16u;
      afr[SET][0][i] = *reinterpret_cast<const frag*>(Ab + o0);
      afr[SET][1][i] = *reinterpret_cast<const frag*>(Ab + (o0 ^ 32u));
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < NT; ++j) bfr[SET][ks][j] = *reinterpret_cast<const frag*>(Bb + j * 32 * 64 + b_offk[ks]);
  };
  using Z = std::integral_constant<int, 0>;

  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (slabs 0 - 2 landed long ago) the patch is written
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  load_frags(Z{}, Z{}, 0);

  auto stage = [&](auto tc, auto parc, int chunk) {
    constexpr int T = decltype(tc)::value, PAR = decltype(parc)::value;
    constexpr int SET = (T + PAR) & 1;
    const bool more_chunks = chunk + 1 < CMC;
    const bool has_next = T < 8 || more_chunks;
    // stage s + 1's weights must have landed; the only younger loads are stage s + 2's
    if (T >= 7 && !more_chunks) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPB) : "memory");
    if (has_next) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    constexpr int TN1 = T < 8 ? T + 1 : 0;
    const int cn = T < 8 ? chunk : chunk + 1;
    const unsigned char* Ab = smem + MID_BASE + (unsigned)cn * MID_CH;
    const unsigned char* Bb = b_frag + (TN1 % 3) * B_SLOT;
    constexpr int NS = SET ^ 1;
    static_for(
        [&](auto gc) {
          constexpr int g = decltype(gc)::value;
          constexpr int ks = g / MT, i = g % MT;
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = HT::mfma(afr[SET][ks][i], bfr[SET][ks][j], acc[i][j]);
          if constexpr (g < MT) {
            if (has_next) {
              unsigned rb0 = rowbase[g];
              asm volatile("" : "+v"(rb0));
              const unsigned R = rb0 + (unsigned)tapoff[TN1];
              const unsigned o0 = (R << 6) + (((R >> 2) ^ (unsigned)hh) & 3u) * 16u;
              afr[NS][0][g] = *reinterpret_cast<const frag*>(Ab + o0);
              afr[NS][1][g] = *reinterpret_cast<const frag*>(Ab + (o0 ^ 32u));
            }
          }
          if constexpr (g == MT) {
            constexpr int T3 = (T + 3) % 9;
            const int c3 = chunk + (T + 3 >= 9 ? 1 : 0);
            if (c3 < CMC) issue_b(c3, T3, (unsigned)(T3 % 3));
          }
          if constexpr (g >= MT) {
            if (has_next) {
              constexpr int lo = (g - MT) * (2 * NT) / MT, hi = (g - MT + 1) * (2 * NT) / MT;
#pragma unroll
              for (int r = lo; r < hi; ++r)
                bfr[NS][r / NT][r % NT] = *reinterpret_cast<const frag*>(Bb + (r % NT) * 32 * 64 + b_offk[r / NT]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        },
        std::make_integer_sequence<int, 2 * MT>{});
  };
  {
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    static_assert(CMC % 2 == 0, "two chunks per trip");
    for (int chunk = 0; chunk < ((a.abl & 2) ? 0 : CMC); chunk += 2) {
      static_for([&](auto tc) { stage(tc, P0{}, chunk); }, std::make_integer_sequence<int, 9>{});
      static_for([&](auto tc) { stage(tc, P1{}, chunk + 1); }, std::make_integer_sequence<int, 9>{});
    }
  }

  // ================================================================ epilogue (conv_p8_impl.h's 16-bit epilogue)
  if (!(a.abl & 4)) {
    const float slope = a.act2 == ME_ACT_LEAKY ? 0.1f : 1.0f;
    constexpr int TP = 36;
    __syncthreads();
    float* tbuf = reinterpret_cast<float*>(smem) + wave * (2 * 32 * TP);
    const int prow = lane >> 2, c8 = (lane & 3) * 8;
    unsigned short* __restrict__ yb = reinterpret_cast<unsigned short*>(a.y);
    const unsigned short* __restrict__ rb = reinterpret_cast<const unsigned short*>(a.res);
    int mrow[MT][2];
  #pragma unroll
    for (int i = 0; i < MT; ++i)
  #pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const long long q = q0 + wr * TM + i * 32 + pass * 16 + prow;
        int m = -1;
        if (q < a.Mp) {
          const unsigned u = (unsigned)q;
          const unsigned n = udiv_magic(u, a.ip_m, a.ip_s);
          const unsigned rem = u - n * (unsigned)a.Ip;
          const unsigned y = udiv_magic(rem, a.wp_m, a.wp_s);
          const unsigned x = rem - y * (unsigned)a.Wp;
          if (y < (unsigned)H && x < (unsigned)W) m = (int)((n * (unsigned)H + y) * (unsigned)W + x);
        }
        mrow[i][pass] = m;
      }
    constexpr int RD = MT <= 2 ? MT : 2;
    uint4 rres[RD][NT][2];
    auto fetch_res = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if (!rb) return;
  #pragma unroll
      for (int j = 0; j < NT; ++j)
  #pragma unroll
        for (int pass = 0; pass < 2; ++pass)
          rres[i % RD][j][pass] = mrow[i][pass] >= 0
                                      ? *reinterpret_cast<const uint4*>(rb + (long long)mrow[i][pass] * a.res_pitch + wc * TN + j * 32 + c8)
                                      : make_uint4(0u, 0u, 0u, 0u);
    };
    
    // (requesting these pieces in front of the last two chunks of the main loop instead: no gain, 78 - 80 us either way -
    //  every workgroup is in its epilogue at the same time and the chip's memory system is what they wait for;
    //  profiles/r06_micro_bneck_resprefetch.txt)
    static_for(fetch_res, std::make_integer_sequence<int, RD>{});
    auto block_out = [&](auto ic, auto jc) {
      constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
      float* tb = tbuf + ((i * NT + j) & 1) * (32 * TP);
      const int cb = wc * TN + j * 32;
      const float sc = a.sc2[cb + r32], sh = a.sh2[cb + r32];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = acc[i][j][e] * sc + sh;
        v = fmaxf(v, v * slope);
        tb[((e & 3) + 8 * (e >> 2) + 4 * hh) * TP + r32] = v;
      }
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int row = pass * 16 + prow;
        const float4 lo = *reinterpret_cast<const float4*>(tb + row * TP + c8);
        const float4 hi = *reinterpret_cast<const float4*>(tb + row * TP + c8 + 4);
        const long long m = mrow[i][pass];
        if (m >= 0) {
          float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
          if (rb) {
            const uint4 r4 = rres[i % RD][j][pass];
            const unsigned rr[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              v[2 * k] += HT::from(rr[k] & 0xffffu);
              v[2 * k + 1] += HT::from(rr[k] >> 16);
            }
          }
          uint4 o;
          o.x = pack2<F16>(v[0], v[1]);
          o.y = pack2<F16>(v[2], v[3]);
          o.z = pack2<F16>(v[4], v[5]);
          o.w = pack2<F16>(v[6], v[7]);
          store16(yb + m * a.y_pitch + cb + c8, o, a.store_mode);
        }
      }
    };
    static_for(
        [&](auto ic) {
          constexpr int i = decltype(ic)::value;
          static_for([&](auto jc) { block_out(ic, jc); }, std::make_integer_sequence<int, NT>{});
          if constexpr (i + RD < MT) fetch_res(std::integral_constant<int, i + RD>{});
        },
        std::make_integer_sequence<int, MT>{});
  }
}

void magic_u32(unsigned d, unsigned* m, unsigned* s) {
  unsigned sh = 0;
  while ((1ull << sh) < d) ++sh;
  *s = sh;
  *m = (unsigned)(((1ull << 32) * ((1ull << sh) - d)) / d + 1);
}

template <int F16, int WR, int WC, int MT, int NT, int CMC, int P1MT, int NBUF>
int launch_bneck(const me_bneck16_desc* d, hipStream_t stream) {
  constexpr int BM = 32 * MT * WR, BN = 32 * NT * WC;
  constexpr int ROWS_PAD = 32 * P1MT * (8 / CMC);
  BnArgs a = {};
  a.x = d->x; a.w1t = d->w1_tiled; a.sc1 = d->scale1; a.sh1 = d->shift1;
  a.w2t = d->w2_tiled; a.sc2 = d->scale2; a.sh2 = d->shift2; a.res = d->res; a.y = d->y;
  a.x_pitch = d->x_pitch; a.res_pitch = d->res_pitch; a.y_pitch = d->y_pitch;
  a.n = d->n; a.h = d->h; a.w = d->w; a.cin = d->cin; a.cmid = d->cmid; a.cout = d->cout; a.act1 = d->act1; a.act2 = d->act2;
  a.Wp = d->w + 1;
  a.Ip = (d->h + 1) * a.Wp;
  a.halo = a.Wp + 1;
  a.Mp = (long long)d->n * a.Ip;
  a.rows = BM + 2 * a.halo;
  ME_REQUIRE(a.rows <= ROWS_PAD, ME_E_TOOBIG, "me_bneck_h16: tile %d holds %d patch rows, a %d-wide map needs %d", d->tile, ROWS_PAD, d->w, a.rows);
  ME_REQUIRE(a.Mp < (1ll << 31), ME_E_TOOBIG, "me_bneck_h16: too many padded positions");
  magic_u32((unsigned)a.Ip, &a.ip_m, &a.ip_s);
  magic_u32((unsigned)a.Wp, &a.wp_m, &a.wp_s);
  a.tiles_m = (int)((a.Mp + BM - 1) / BM);
  a.store_mode = me::store_mode();
  {
    const char* e = getenv("MILLIEYE_BNECK_ABL");
    a.abl = e ? atoi(e) : 0;
  }
  size_t lds = 3 * (size_t)BN * 64 + (size_t)(CMC > NBUF ? CMC : NBUF) * ROWS_PAD * 64 + (size_t)NBUF * CMC * 32 * 64;
  const size_t epi = 8 * 2 * 32 * 36 * sizeof(float);
  if (lds < epi) lds = epi;
  ME_REQUIRE(lds <= 160 * 1024, ME_E_TOOBIG, "me_bneck_h16: tile %d needs %zu bytes of LDS", d->tile, lds);
  auto kern = bneck_kernel<F16, WR, WC, MT, NT, CMC, P1MT, NBUF>;
  static bool attr_set = false;
  if (!attr_set) {
    ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)a.tiles_m), dim3(512), lds, stream, a);
  return me::check_launch("bneck_h16");
}

}  // namespace

extern "C" {

// tile ids (me_bneck16_desc.tile): BM x (cin -> cmid -> cout)
//   1: 192 x (* -> 128 -> 256)   52 x 52 blocks of Darknet-53 (patch rows <= 320: maps up to 62 wide)
//   3: 512 x (* ->  64 -> 128)   104 x 104 blocks (patch rows <= 768: maps up to 126 wide)
//   4: 256 x (* ->  64 -> 128)   (patch rows <= 512: maps up to 126 wide)
int me_bneck_h16_supported(const me_bneck16_desc* d) {
  if (!d) return 0;
  if (d->cin % 32 || d->cin < 32) return 0;
  const int halo2 = 2 * (d->w + 2);
  int bm, pad;
  if (d->cmid == 128 && d->cout == 256) {
    if (d->tile == 1) { bm = 192; pad = 320; }
    else return 0;
  } else if (d->cmid == 64 && d->cout == 128) {
    if (d->tile == 3) { bm = 512; pad = 768; }
    else if (d->tile == 4) { bm = 256; pad = 512; }
    else return 0;
  } else {
    return 0;
  }
  if (bm + halo2 > pad) return 0;
  if ((long long)d->n * (d->h + 1) * (d->w + 1) >= (1ll << 31)) return 0;
  const long long img_bytes = (long long)d->h * d->w * d->x_pitch * 2;
  const long long span = (1024 / ((long long)(d->h + 1) * (d->w + 1))) + 2;
  return span * img_bytes < (1ll << 31) ? 1 : 0;
}

int me_bneck_h16(const me_bneck16_desc* d, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(d != nullptr, ME_E_NULLPTR, "me_bneck_h16: null descriptor");
  ME_REQUIRE(d->x && d->w1_tiled && d->w2_tiled && d->scale1 && d->shift1 && d->scale2 && d->shift2 && d->y, ME_E_NULLPTR,
             "me_bneck_h16: null tensor pointer");
  ME_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0, ME_E_BADARG, "me_bneck_h16: non-positive dimension");
  ME_REQUIRE(d->half_type == 0 || d->half_type == 1, ME_E_BADARG, "me_bneck_h16: half_type must be 0 (bf16) or 1 (f16)");
  ME_REQUIRE((d->act1 == ME_ACT_LEAKY || d->act1 == ME_ACT_LINEAR) && (d->act2 == ME_ACT_LEAKY || d->act2 == ME_ACT_LINEAR), ME_E_BADARG,
             "me_bneck_h16: activations must be leaky or linear");
  ME_REQUIRE(d->x_pitch >= d->cin && d->x_pitch % 8 == 0 && d->y_pitch >= d->cout && d->y_pitch % 8 == 0, ME_E_ALIGN,
             "me_bneck_h16: pitches must cover the channels and be multiples of 8");
  ME_REQUIRE(!d->res || (d->res_pitch >= d->cout && d->res_pitch % 8 == 0 && me::aligned16(d->res)), ME_E_ALIGN,
             "me_bneck_h16: residual pitch / alignment");
  ME_REQUIRE(me::aligned16(d->x) && me::aligned16(d->y) && me::aligned16(d->w1_tiled) && me::aligned16(d->w2_tiled) &&
                 me::aligned16(d->scale1) && me::aligned16(d->shift1), ME_E_ALIGN, "me_bneck_h16: 16-byte alignment");
  ME_REQUIRE(me_bneck_h16_supported(d), ME_E_BADARG,
             "me_bneck_h16: no instance for tile %d, %d -> %d -> %d channels on a %d x %d map", d->tile, d->cin, d->cmid, d->cout,
             d->h, d->w);
#define ME_BN(WR, WC, MT, NT, CMC, P1MT, NBUF) \
  (d->half_type ? launch_bneck<1, WR, WC, MT, NT, CMC, P1MT, NBUF>(d, stream) : launch_bneck<0, WR, WC, MT, NT, CMC, P1MT, NBUF>(d, stream))
  switch (d->tile) {
    case 1: return ME_BN(2, 4, 3, 2, 4, 5, 4);   // 48 + 80 + 32 KB of LDS
    case 3: return ME_BN(4, 2, 4, 2, 2, 6, 2);   // 24 + 96 + 8
    case 4: return ME_BN(4, 2, 2, 2, 2, 4, 3);   // 24 + 96 + 12
    default: break;
  }
#undef ME_BN
  ME_REQUIRE(false, ME_E_BADARG, "me_bneck_h16: unknown tile id %d", d->tile);
  return 0;
}

}  // extern "C"
