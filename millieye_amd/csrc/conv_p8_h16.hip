// conv_p8_h16.hip - patch-resident, big-tile generation of the 16-bit 3x3 / stride-1 convolution (gfx950).
//
// Why (profiles/r01_h16_*: the per-tap kernel of conv_h16.hip pulls ~600 MB of operand re-reads through L2 per launch and
// issues 0.46 DMA + 1.18 LDS instructions per MFMA; with no DMA at all the same loop runs at 1.4 PFLOP/s): the nine taps of a
// 3x3 filter read the same input pixels shifted by one, so a tile's input is brought into LDS ONCE per 32-channel chunk and
// the taps are walked by LDS row offset; only the weights stream per tap (3-slot ring).
//
// Index space.  Output pixels are numbered in a PADDED-LINEAR space: row pitch Wp = W + 1 (one zero column behind every row),
// image pitch Ip = (H + 1) * Wp (one zero row behind every image), q = n * Ip + y * Wp + x.  The neighbour of q for tap
// (dy, dx) is q + dy * Wp + dx for EVERY q - the zero column is the left neighbour of the next row and the right neighbour of
// this one, the zero row is "above" the next image and "below" this one - so a tile of BM consecutive positions needs the
// contiguous range [q0 - Wp - 1, q0 + BM + Wp + 1) whatever it contains: tiles may start anywhere and cross rows and images
// (the first patch kernel, tile 41, had to stop at image borders: 22-34 % idle rows at 26x26 / 13x13).  Pad positions cost
// MFMA work ((H+1)(W+1)/(HW) - 1: 4 % at 52x52, 16 % at 13x13) and are neither loaded (out-of-range DMA lanes zero-fill LDS)
// nor stored.  HBM layouts do not change: dense NHWC in, dense NHWC out.
//
// Workgroup = 8 waves (WR x WC), tile BM x BN = (32 MT WR) x (32 NT WC), one workgroup per CU for the big tiles.
// LDS: [weight ring: 3 x BN x 64 B][patch slot 0][patch slot 1], patch rows are 64 B (32 channels), XOR-swizzled by
// (row >> 2) & 3 like every other tile of this library (ds_read_b128 conflict-free).
// Stage = (chunk, tap): ONE barrier per 2 * MT * NT MFMAs per wave (the per-tap kernel: per 8), DMA instructions per wave and
// stage: BN / 128 for the weights + the next chunk's patch once per nine stages.
#include <cstdlib>

#include "conv16_common.h"
#include "conv_p8_impl.h"

namespace {
using namespace me_p8;

// storage-type traits of the generic kernel: 64-byte chunk = 32 channels, one ds_read_b128 = one MFMA operand
template <int F16>
struct DT16 {
  using frag = typename H16<F16>::v8;
  using elem = std::conditional_t<F16 != 0, _Float16, __bf16>;
  static constexpr int kBytes = 2, kChunk = 32;
  static __device__ __forceinline__ void mfma(const frag& a, const frag& b, f32x16& c) { c = H16<F16>::mfma(a, b, c); }
  static __device__ __forceinline__ frag fill(float v) {
    frag f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (elem)(v + 0.01f * e);
    return f;
  }
  static __device__ __forceinline__ float from16(unsigned b) { return H16<F16>::from(b); }
  static __device__ __forceinline__ unsigned pack2(float lo, float hi) { return ::pack2<F16>(lo, hi); }
};

void magic_u32(unsigned d, unsigned* m, unsigned* s) {
  unsigned sh = 0;
  while ((1ull << sh) < d) ++sh;
  *s = sh;
  *m = (unsigned)(((1ull << 32) * ((1ull << sh) - d)) / d + 1);
}

// K split, second pass: slabs [tile][split][BM][BN] of raw accumulators -> fixed-order sum, then the epilogue of the one-pass
// kernel (affine, leaky, + residual, round to the storage type; one thread = 8 consecutive channels of one position = one
// 16-byte store).  Positions are padded-linear: pad positions and positions behind the last image are skipped.
template <int F16>
__global__ __launch_bounds__(256) void conv3x3_p8_reduce_h16(P8Args a, int bm, int bn) {
  const P8Conv& p = a.c;
  const int cols8 = bn / 8, tile_elems = bm * bn;
  const long long total = (long long)p.tiles_m * p.tiles_n * bm * cols8;
  const float* partial = reinterpret_cast<const float*>(p.partial);
  const float slope = p.act == ME_ACT_LEAKY ? 0.1f : 1.0f;
  unsigned short* __restrict__ yb = reinterpret_cast<unsigned short*>(p.y);
  const unsigned short* __restrict__ rb = reinterpret_cast<const unsigned short*>(p.res);
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c8 = (int)(idx % cols8) * 8;
    const long long t2 = idx / cols8;
    const int lm = (int)(t2 % bm);
    const int tile = (int)(t2 / bm);
    const long long q = (long long)(tile / p.tiles_n) * bm + lm;
    if (q >= a.Mp) continue;
    const unsigned u = (unsigned)q;
    const unsigned n = udiv_magic(u, a.ip_m, a.ip_s);
    const unsigned rem = u - n * (unsigned)a.Ip;
    const unsigned y = udiv_magic(rem, a.wp_m, a.wp_s);
    const unsigned x = rem - y * (unsigned)a.Wp;
    if (y >= (unsigned)p.h || x >= (unsigned)p.w) continue;
    const long long m = ((long long)n * p.h + y) * p.w + x;
    const int co = (tile % p.tiles_n) * bn + c8;
    const float* src = partial + (long long)tile * p.splitk * tile_elems + (long long)lm * bn + c8;
    float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
    for (int k = 1; k < p.splitk; ++k) {
      const float4 l2 = *reinterpret_cast<const float4*>(src + (long long)k * tile_elems);
      const float4 h2 = *reinterpret_cast<const float4*>(src + (long long)k * tile_elems + 4);
      lo.x += l2.x; lo.y += l2.y; lo.z += l2.z; lo.w += l2.w;
      hi.x += h2.x; hi.y += h2.y; hi.z += h2.z; hi.w += h2.w;
    }
    float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float t = v[k] * p.scale[co + k] + p.shift[co + k];
      v[k] = fmaxf(t, t * slope);
    }
    if (rb) {
      const uint4 r4 = *reinterpret_cast<const uint4*>(rb + m * p.res_pitch + co);
      const unsigned rr[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[2 * k] += DT16<F16>::from16(rr[k] & 0xffffu);
        v[2 * k + 1] += DT16<F16>::from16(rr[k] >> 16);
      }
    }
    uint4 o;
    o.x = DT16<F16>::pack2(v[0], v[1]);
    o.y = DT16<F16>::pack2(v[2], v[3]);
    o.z = DT16<F16>::pack2(v[4], v[5]);
    o.w = DT16<F16>::pack2(v[6], v[7]);
    *reinterpret_cast<uint4*>(yb + m * p.y_pitch + co) = o;
  }
}

template <int WR, int WC, int MT, int NT, int PIPE, int MINB, int F16, int ABL = 0, int TG = 1, int DS = 0, int RB = 3>
int launch_p8(const Conv16P& p, hipStream_t stream) {
  constexpr int BM = 32 * MT * WR, BN = 32 * NT * WC;
  P8Args a = {};
  a.c.x = p.x; a.c.wgt_tiled = p.wgt_tiled; a.c.scale = p.scale; a.c.shift = p.shift; a.c.res = p.res; a.c.y = p.y;
  a.c.x_pitch = p.x_pitch; a.c.res_pitch = p.res_pitch; a.c.y_pitch = p.y_pitch;
  a.c.n = p.n; a.c.h = p.h; a.c.w = p.w; a.c.cin = p.cin; a.c.cout = p.cout; a.c.act = p.act;
  a.c.partial = p.partial;
  a.Wp = p.w + 1;
  a.Ip = (p.h + 1) * a.Wp;
  a.halo = a.Wp + 1;
  a.Mp = (long long)p.n * a.Ip;
  a.rows = BM + 2 * a.halo;
  constexpr int NWAVES = WR * WC;
  constexpr int NWA = DS ? NWAVES / 2 : NWAVES;   // waves that fetch the patch (DS: the upper half, see conv_p8_impl.h; DS = 4: the four loader waves)
  a.lpa = ((a.rows + 15) / 16 + NWA - 1) / NWA;
  if (DS == 4) {
    ME_REQUIRE(a.lpa <= kLpaMax, ME_E_TOOBIG, "me_conv2d_h16: patch of %d rows does not fit (map too wide for this tile)", a.rows);
    a.lpa = kLpaMax;   // static DMA bookkeeping: every loader issues seven pieces per chunk (the surplus ones out of range)
  }
  ME_REQUIRE(a.lpa <= kLpaMax, ME_E_TOOBIG, "me_conv2d_h16: patch of %d rows does not fit (map too wide for this tile)", a.rows);
  ME_REQUIRE(a.Mp < (1ll << 31), ME_E_TOOBIG, "me_conv2d_h16: too many padded positions");
  magic_u32((unsigned)a.Ip, &a.ip_m, &a.ip_s);
  magic_u32((unsigned)a.Wp, &a.wp_m, &a.wp_s);
  a.c.tiles_m = (int)((a.Mp + BM - 1) / BM);
  a.c.tiles_n = p.cout / BN;
  size_t lds = (size_t)RB * TG * BN * 64 + 2 * (size_t)a.lpa * NWA * 1024;
  const size_t epi = NWAVES * 2 * 32 * 36 * sizeof(float);
  if (lds < epi) lds = epi;
  ME_REQUIRE(lds <= (MINB == 2 ? 80 : 160) * 1024, ME_E_TOOBIG,
             "me_conv2d_h16: this tile needs %zu bytes of LDS for a %d-wide map", lds, p.w);
  auto kern = conv3x3_p8_kernel<WR, WC, MT, NT, PIPE, MINB, DT16<F16>, ABL, false, TG, DS, RB>;
  auto kern_sk = conv3x3_p8_kernel<WR, WC, MT, NT, PIPE, MINB, DT16<F16>, ABL == 0 ? 0 : ABL, ABL == 0, TG, DS, RB>;  // K-split instance
  static bool attr_set = false;
  if (!attr_set) {
    ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern_sk), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  // K split (p.splitk > 1): every tile is cut into splitk workgroups along the 32-channel chunks - for the layers whose tile
  // count leaves CUs idle or with a lone workgroup (13x13: 176 tiles of 256 x 128); slabs + a second launch
  const int cs = p.cin / 32;
  a.c.store_mode = me::store_mode();
  if (ABL == 9) {
    const char* e = getenv("MILLIEYE_STAMP_WAVE");
    a.stamp_wave = e ? atoi(e) : 0;
  }
  a.c.splitk = p.splitk > cs ? cs : (p.splitk < 1 ? 1 : p.splitk);
  a.c.cps = (cs + a.c.splitk - 1) / a.c.splitk;
  while (a.c.splitk > 1 && (a.c.splitk - 1) * a.c.cps >= cs) --a.c.splitk;
  const long long tiles = (long long)a.c.tiles_m * a.c.tiles_n;
  const long long blocks = tiles * a.c.splitk;
  ME_REQUIRE(blocks < (1ll << 31), ME_E_TOOBIG, "me_conv2d_h16: grid too large");
  if (a.c.splitk > 1) {
    const long long need = blocks * BM * BN * (long long)sizeof(float);
    ME_REQUIRE(ABL == 0 && p.partial && p.partial_bytes >= need, ME_E_BADARG,
               "me_conv2d_h16: tile with split_k=%d needs a workspace of %lld bytes", a.c.splitk, need);
  }
  static const int nmap_env = [] {
    const char* e = getenv("MILLIEYE_P8_NMAP");
    return e ? atoi(e) : 0;
  }();
  a.c.nmap = (nmap_env && a.c.splitk == 1 && a.c.tiles_n <= 8 && 8 % a.c.tiles_n == 0) ? 1 : 0;
  if (a.c.splitk > 1)
    hipLaunchKernelGGL(kern_sk, dim3((unsigned)blocks), dim3(64 * (NWAVES + (DS == 4 ? 4 : 0))), lds, stream, a);
  else if (a.c.nmap) {
    const int G = 8 / a.c.tiles_n;
    const unsigned per_xcd = (unsigned)((a.c.tiles_m + G - 1) / G);
    hipLaunchKernelGGL(kern, dim3(8u * per_xcd), dim3(64 * (NWAVES + (DS == 4 ? 4 : 0))), lds, stream, a);
  } else
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * (NWAVES + (DS == 4 ? 4 : 0))), lds, stream, a);
  int rc = me::check_launch("conv3x3_p8_h16");
  if (rc || a.c.splitk == 1) return rc;
  long long rb = (tiles * BM * (BN / 8) + 255) / 256;
  if (rb > 256 * 16) rb = 256 * 16;
  hipLaunchKernelGGL(conv3x3_p8_reduce_h16<F16>, dim3((unsigned)rb), dim3(256), 0, stream, a, BM, BN);
  return me::check_launch("conv3x3_p8_reduce_h16");
}

}  // namespace

namespace me16 {

// tile ids 100 + ...: patch-resident big tiles (3x3, stride 1, pad 1, 16-byte epilogue, cout % BN == 0)
bool p8_eligible(const Conv16P& p, int tile) {
  if (p.ks != 3 || p.stride != 1 || p.pad != 1 || p.ups != 1 || p.x_nchw || p.cin % 32 || !p.vec_epi) return false;
  if (!p.wgt_tiled || !me::aligned16(p.wgt_tiled)) return false;
  const int bn = (tile % 10 == 1) ? 128 : 256;  // ids ...1: 128 output channels per tile, else 256
  if (p.cout % bn) return false;
  if ((long long)p.n * (p.h + 1) * (p.w + 1) >= (1ll << 31)) return false;
  // descriptor windows: a tile spans at most a few images / 256 weight rows
  const long long img_bytes = (long long)p.h * p.w * p.x_pitch * 2;
  const long long span = (1024 / ((long long)(p.h + 1) * (p.w + 1))) + 2;
  return span * img_bytes < (1ll << 31) && 256ll * p.ktot * 2 < (1ll << 31);
}

// BM x BN of a patch tile id (0 x 0: unknown id)
void p8_tile_shape(int tile, int* bm, int* bn) {
  static const int ids[][3] = {{100, 128, 256}, {110, 192, 256}, {120, 256, 256}, {101, 128, 128}, {121, 256, 128},
                               {131, 384, 128}, {141, 512, 128}, {200, 128, 256}, {201, 128, 128}, {221, 256, 128},
                               {301, 128, 128}, {311, 192, 128}, {321, 256, 128}, {331, 256, 128},
                               {421, 256, 128}, {431, 384, 128}, {441, 512, 128},
                               {621, 256, 128},
                               {721, 256, 128}, {731, 384, 128},
                               {810, 192, 256}, {820, 256, 256}, {821, 256, 128}, {831, 384, 128},
                               {841, 512, 128}, {1210, 192, 256}, {1221, 256, 128}, {1231, 384, 128}};
  *bm = *bn = 0;
  for (const auto& t : ids)
    if (t[0] == tile) {
      *bm = t[1];
      *bn = t[2];
    }
}

// scratch of a K-split patch tile: compact slabs of every workgroup
long long p8_workspace_bytes(const Conv16P& p, int tile, int split) {
  int bm, bn;
  p8_tile_shape(tile, &bm, &bn);
  if (!bm || split <= 1 || p.cout % bn) return 0;
  const long long mp = (long long)p.n * (p.h + 1) * (p.w + 1);
  const int cs = p.cin / 32;
  if (split > cs) split = cs;
  return ((mp + bm - 1) / bm) * (p.cout / bn) * split * bm * bn * (long long)sizeof(float);
}

int launch_p8_tile(const Conv16P& p, int tile, hipStream_t stream) {
  ME_REQUIRE(p8_eligible(p, tile), ME_E_BADARG,
             "me_conv2d_h16: tile %d needs a 3x3 / stride 1 / pad 1 layer, cin %% 32 == 0, cout %% tile width == 0, 16-bit "
             "output with the 16-byte epilogue, and the tiled weight copy (wgt_tiled)", tile);
#define ME_P8(WR, WC, MT, NT, PIPE, MINB) \
  (p.f16 ? launch_p8<WR, WC, MT, NT, PIPE, MINB, 1>(p, stream) : launch_p8<WR, WC, MT, NT, PIPE, MINB, 0>(p, stream))
  switch (tile) {
    // one workgroup per CU, register-pipelined fragments
    case 100: return ME_P8(2, 4, 2, 2, 1, 1);   // 128 x 256
    case 110: return ME_P8(2, 4, 3, 2, 1, 1);   // 192 x 256
    case 120: return ME_P8(2, 4, 4, 2, 1, 1);   // 256 x 256
    case 101: return ME_P8(4, 2, 1, 2, 1, 1);   // 128 x 128
    case 121: return ME_P8(4, 2, 2, 2, 1, 1);   // 256 x 128
    case 131: return ME_P8(4, 2, 3, 2, 1, 1);   // 384 x 128
    case 141: return ME_P8(4, 2, 4, 2, 1, 1);   // 512 x 128
    // two workgroups per CU (<= 128 VGPRs, <= 80 KB LDS): their main loops and epilogues interleave
    case 200: return ME_P8(2, 4, 2, 2, 0, 2);   // 128 x 256
    case 201: return ME_P8(4, 2, 1, 2, 0, 2);   // 128 x 128
    case 221: return ME_P8(4, 2, 2, 2, 0, 2);   // 256 x 128
    // two independent 4-wave workgroups per CU (one wave per SIMD each, register-pipelined fragments, <= 256 VGPRs)
    case 301: return ME_P8(2, 2, 2, 2, 1, 2);   // 128 x 128
    case 311: return ME_P8(2, 2, 3, 2, 1, 2);   // 192 x 128
    case 321: return ME_P8(2, 2, 4, 2, 1, 2);   // 256 x 128
    case 331: return ME_P8(4, 1, 2, 4, 1, 2);   // 256 x 128, wave tile 64 x 128
    // tap groups (round 3): the one-workgroup tiles with a barrier per THREE taps (72 KB of weight slabs per chunk in LDS)
    case 421: return p.f16 ? launch_p8<4, 2, 2, 2, 1, 1, 1, 0, 3>(p, stream) : launch_p8<4, 2, 2, 2, 1, 1, 0, 0, 3>(p, stream);
    case 431: return p.f16 ? launch_p8<4, 2, 3, 2, 1, 1, 1, 0, 3>(p, stream) : launch_p8<4, 2, 3, 2, 1, 1, 0, 0, 3>(p, stream);
    case 441: return p.f16 ? launch_p8<4, 2, 4, 2, 1, 1, 1, 0, 3>(p, stream) : launch_p8<4, 2, 4, 2, 1, 1, 0, 0, 3>(p, stream);
    // DMA duty split (round 4): waves 0-3 fetch weight slabs only, waves 4-7 the patch only (conv_p8_impl.h, DS)
#define ME_P8D(WR, WC, MT, NT, PIPE, MINB, TG) \
  (p.f16 ? launch_p8<WR, WC, MT, NT, PIPE, MINB, 1, 0, TG, 1>(p, stream) : launch_p8<WR, WC, MT, NT, PIPE, MINB, 0, 0, TG, 1>(p, stream))
    case 621: return ME_P8D(4, 2, 2, 2, 0, 2, 1);   // 221
    case 721: return ME_P8D(4, 2, 2, 2, 1, 1, 3);   // 421
    case 731: return ME_P8D(4, 2, 3, 2, 1, 1, 3);   // 431
#undef ME_P8D
    // ping-pong (round 4, PIPE = 2): one 8-wave workgroup per CU whose halves run half a stage apart (conv_p8_impl.h)
#define ME_P8R(WR, WC, MT, NT, RB) \
  (p.f16 ? launch_p8<WR, WC, MT, NT, 2, 1, 1, 0, 1, 0, RB>(p, stream) : launch_p8<WR, WC, MT, NT, 2, 1, 0, 0, 1, 0, RB>(p, stream))
    case 810: return ME_P8R(2, 4, 3, 2, 3);   // 192 x 256
    case 820: return ME_P8R(2, 4, 4, 2, 3);   // 256 x 256
    case 821: return ME_P8R(4, 2, 2, 2, 3);   // 256 x 128
    case 831: return ME_P8R(4, 2, 3, 2, 3);   // 384 x 128
    case 841: return ME_P8R(4, 2, 4, 2, 3);   // 512 x 128
    // ... + four loader waves (DS = 4): the consumers' load segment is fragment reads only
#define ME_P8L(WR, WC, MT, NT) \
  (p.f16 ? launch_p8<WR, WC, MT, NT, 2, 1, 1, 0, 1, 4, 3>(p, stream) : launch_p8<WR, WC, MT, NT, 2, 1, 0, 0, 1, 4, 3>(p, stream))
    case 1210: return ME_P8L(2, 4, 3, 2);   // 192 x 256
    case 1221: return ME_P8L(4, 2, 2, 2);   // 256 x 128
    case 1231: return ME_P8L(4, 2, 3, 2);   // 384 x 128
#undef ME_P8L
#undef ME_P8R
    default: break;
  }
  // Ablation / instrumented instances of the tuning tools (tools/p8_timeline.py, tools/p8_bench.py --ablate): they skip parts
  // of the kernel and produce WRONG results, so the public ``tile`` field only reaches them when the process opted in.
  static const bool ablation_ok = [] {
    const char* e = getenv("MILLIEYE_ABLATION");
    return e && e[0] == '1';
  }();
  ME_REQUIRE(ablation_ok || !((tile >= 180 && tile <= 199) || (tile >= 280 && tile <= 299) || (tile >= 680 && tile <= 699) || (tile >= 880 && tile <= 899) || (tile >= 1080 && tile <= 1099)), ME_E_BADARG,
             "me_conv2d_h16: tile id %d is an ablation instance with wrong results (tuning tools only: MILLIEYE_ABLATION=1)",
             tile);
  if (tile == 199 || tile == 299 || tile == 899)  // 2048 x 8 bytes of time stamps go to the workspace
    ME_REQUIRE(p.partial && p.partial_bytes >= 16384, ME_E_BADARG,
               "me_conv2d_h16: the time-stamp instances need a workspace of at least 16384 bytes");
  if (tile == 196 || tile == 296 || tile == 696 || tile == 896) {  // six 8-byte words per workgroup
    int bm, bn;
    p8_tile_shape(tile == 196 || tile == 896 ? 131 : 221, &bm, &bn);  // (696: same tile shape as 296)
    const long long mp = (long long)p.n * (p.h + 1) * (p.w + 1);
    const long long need = ((mp + bm - 1) / bm) * (p.cout / bn) * 48;
    ME_REQUIRE(p.partial && p.partial_bytes >= need, ME_E_BADARG,
               "me_conv2d_h16: the per-workgroup stamp instances need a workspace of %lld bytes", need);
  }
  switch (tile) {
    case 180: return launch_p8<2, 4, 4, 2, 1, 1, 0, 1>(p, stream);
    case 190: return launch_p8<2, 4, 4, 2, 1, 1, 0, 3>(p, stream);
    case 191: return launch_p8<2, 4, 4, 2, 1, 1, 0, 4>(p, stream);
    case 192: return launch_p8<2, 4, 4, 2, 1, 1, 0, 5>(p, stream);
    case 193: return launch_p8<2, 4, 4, 2, 1, 1, 0, 6>(p, stream);
    case 293: return launch_p8<4, 2, 2, 2, 0, 2, 0, 6>(p, stream);
    case 290: return launch_p8<4, 2, 2, 2, 0, 2, 0, 3>(p, stream);
    case 280: return launch_p8<4, 2, 2, 2, 0, 2, 0, 1>(p, stream);
    case 299: return launch_p8<4, 2, 2, 2, 0, 2, 0, 9>(p, stream);   // 221 with time stamps
    case 199: return launch_p8<4, 2, 3, 2, 1, 1, 0, 9>(p, stream);   // 131 with time stamps
    case 296: return launch_p8<4, 2, 2, 2, 0, 2, 0, 10>(p, stream);  // 221 with per-workgroup stamps
    case 196: return launch_p8<4, 2, 3, 2, 1, 1, 0, 10>(p, stream);  // 131 with per-workgroup stamps
    case 297: return launch_p8<4, 2, 2, 2, 0, 2, 0, 7>(p, stream);
    case 298: return launch_p8<4, 2, 2, 2, 0, 2, 0, 8>(p, stream);
    case 696: return launch_p8<4, 2, 2, 2, 0, 2, 0, 10, 1, 1>(p, stream);  // 621 with per-workgroup stamps
    case 896: return launch_p8<4, 2, 3, 2, 2, 1, 0, 10>(p, stream);        // 831 with per-workgroup stamps
    case 899: return launch_p8<4, 2, 3, 2, 2, 1, 0, 9>(p, stream);         // 831 with time stamps
    case 881: return launch_p8<4, 2, 3, 2, 2, 1, 0, 1>(p, stream);         // 831, every DMA lane out of range
    case 887: return launch_p8<4, 2, 3, 2, 2, 1, 0, 7>(p, stream);         // 831, weights out of range
    case 888: return launch_p8<4, 2, 3, 2, 2, 1, 0, 8>(p, stream);         // 831, patch out of range
    case 890: return launch_p8<4, 2, 3, 2, 2, 1, 0, 3>(p, stream);         // 831 without DMA
    case 893: return launch_p8<4, 2, 3, 2, 2, 1, 0, 6>(p, stream);         // 831 without epilogue
    default: ME_REQUIRE(false, ME_E_BADARG, "me_conv2d_h16: unknown patch tile id %d", tile);
  }
#undef ME_P8
  return 0;
}

}  // namespace me16
