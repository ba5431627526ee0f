// conv3x3_ws_h16.hip - 3x3 convolutions with FEW input channels (32 / 64) in the 16-bit storage modes: weights in registers,
// 2-D input patches streamed through an LDS ring (gfx950).  Tile id 60 of me_conv2d_h16.
//
// Reference blocks: module3_our_dataset/yolov3/models.py:22-41 - the five 3x3 layers of Darknet-53 that live on the 416 / 208 /
// 104 maps (conv1 s2 32->64, conv3 32->64 + shortcut, conv5 s2 64->128, conv7 / conv10 64->128 + shortcut).  They are 51 GFLOP
// each like every other 3x3 layer, but their activations are 220 .. 530 MB per 32-frame batch: 30 .. 65 us of HBM time against
// 20 us of MFMAs.  The kernels that ran them were bound by neither: the per-tap implicit GEMM re-reads the input once per
// filter tap from L2 (9 x 88 MB on the 208 map: 150 us), the padded-linear patch kernel (conv_p8) needs W + 1 halo rows on both
// sides of a tile (4 x on the 208 map) and streams the weights through LDS with a barrier per tap.  Here:
//   * the whole filter of a wave's 32 output channels is K = 9 cin <= 576 deep: 72 / 144 VGPRs of B fragments, loaded once;
//   * a workgroup walks 2-D output tiles (TH x TW pixels; an MFMA row block = MBH x MBW = 32 pixels of it), whose input patch
//     ((TH - 1) s + 3) x ((TW - 1) s + 3) pixels is 1.2 - 1.6 x the tile's own input instead of 4 - 9 x, fetched by
//     buffer_load ... lds straight from the NHWC frame: zero padding and ragged tiles are lanes whose offset is out of range;
//   * one barrier per tile, NSLOT patches deep; the nine taps are LDS row shifts of the resident patch;
//   * the XOR swizzle of the channel chunks sits on the DMA's source side (conv1x1_ws_h16.hip explains the scheme);
//   * epilogue: affine + LeakyReLU, LDS transpose, fused shortcut (16-byte residual loads requested before the tile's MFMAs; the
//     patch refill of such layers is issued behind the epilogue so that the compiler's wait for them does not drain it),
//     16-byte stores.
#include <utility>

#include "conv16_common.h"

namespace {
using namespace me_dma;

struct K3Args {
  const unsigned short* x;
  const unsigned short* wgt;  // [cout][3][3][cin]
  const float* scale;
  const float* shift;
  const unsigned short* res;
  unsigned short* y;
  long long x_pitch, res_pitch, y_pitch;  // elements
  int n, h, w, ho, wo, cout, act;
  int tiles_y, tiles_x, tiles_total, grid_m, store_mode;
};

template <class F, int... J>
__device__ __forceinline__ void sfor3(F&& f, std::integer_sequence<int, J...>) {
  (f(std::integral_constant<int, J>{}), ...);
}

__device__ __forceinline__ void dma_one3(unsigned v, u32x4 r, unsigned dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v], %[r], 0 offen lds\n\t"
               "s_mov_b32 m0, %[k]"
               : [k] "=&s"(keep)
               : [d] "s"(dst), [r] "s"(r), [v] "v"(v)
               : "memory", "scc");
}

template <int N>
__device__ __forceinline__ void wait_vm3() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


template <int CIN, int WN, int WM, int MBH, int MBW, int TMY, int TMX, int S, int NSLOT, int F16>
__global__ __launch_bounds__(64 * WN * WM) void conv3x3_ws_kernel(K3Args a) {
  using v8 = typename H16<F16>::v8;
  static_assert(MBH * MBW == 32, "an MFMA row block is 32 pixels");
  constexpr int NW = WN * WM, TH = TMY * MBH, TW = TMX * MBW, NMB = TMY * TMX;
  static_assert(NMB % WM == 0, "row blocks must split evenly over the wave rows");
  constexpr int MPW = NMB / WM;                       // row blocks per wave and tile
  constexpr int PH = (TH - 1) * S + 3, PWP = (TW - 1) * S + 3, NPIX = PH * PWP;
  constexpr int PIXB = CIN * 2, CPP = CIN / 8;        // bytes / 16-byte chunks per pixel
  static_assert(CPP == 4 || CPP == 8, "cin 32 or 64");
  constexpr int PIECES = (NPIX * PIXB + 1023) / 1024;
  constexpr int ND = (PIECES + NW - 1) / NW;          // DMA instructions per wave and tile
  constexpr unsigned SLOTB = (unsigned)ND * NW * 1024u;
  constexpr int KPT = CIN / 16, KS = 9 * KPT;         // k-steps per tap / in total
  constexpr int TP = 36;
  constexpr int NR = 2 * MPW;                         // residual pieces per lane and tile
  static_assert(NSLOT >= 2 && (NSLOT - 2) * ND + NR <= 60, "ring depth");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave % WN, wm = wave / WN;
  const int r32 = lane & 31, hh = lane >> 5;
  const int bid = blockIdx.x;
  const int tile_n = bid / a.grid_m, b_m = bid - tile_n * a.grid_m;
  const int n0 = tile_n * (32 * WN) + wn * 32;

  auto swz = [](int pidx) { return CPP == 4 ? ((pidx >> 2) & 3) : ((pidx >> 1) & 7); };

  // ---- weights: lane (channel n0 + r32, k half hh) holds k = tap * CIN + sub * 16 + 8 hh .. + 8 of every k-step ---------------
  v8 wf[KS];
  {
    const unsigned short* wrow = a.wgt + (long long)(n0 + r32) * (9 * CIN) + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wf[ks] = *reinterpret_cast<const v8*>(wrow + 16 * ks);  // (tap, sub) = (ks / KPT, ks % KPT)
  }
  const float sc = a.scale[n0 + r32], sh = a.shift[n0 + r32];
  const float slope = a.act == ME_ACT_LEAKY ? 0.1f : 1.0f;

  // ---- DMA lanes: piece (wave * ND + i), lane l = chunk q of the slot = (patch pixel, chunk position) -----------------------
  unsigned pyx[ND];  // py | px << 12 | source chunk << 24 | valid << 31
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const int q = (wave * ND + i) * 64 + lane;
    const int pidx = q / CPP, pos = q % CPP;
    const int py = pidx / PWP, px = pidx - py * PWP;
    pyx[i] = (unsigned)py | ((unsigned)px << 12) | ((unsigned)(pos ^ swz(pidx)) << 24) | (pidx < NPIX ? 0x80000000u : 0u);
  }
  const unsigned pitchb = (unsigned)(a.x_pitch * 2);
  const unsigned wave_dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(wave * ND) * 1024u);
  const unsigned long long xbase = (unsigned long long)a.x;
  const int per_img = a.tiles_y * a.tiles_x;
  const unsigned img_bytes = (unsigned)a.h * (unsigned)a.w * pitchb;  // < 2^31 (checked by the launcher)

  auto issue = [&](int t, int slot) {
    u32x4 r;
    int iy0 = 0, ix0 = 0;
    unsigned long long b = xbase;
    unsigned recs = 0;  // tiles behind the end: every lane out of range
    if (t < a.tiles_total) {
      const int nimg = t / per_img, rem = t - nimg * per_img;
      const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
      iy0 = ty * (TH * S) - 1;
      ix0 = tx * (TW * S) - 1;
      b = xbase + (unsigned long long)nimg * img_bytes;
      recs = img_bytes;
    }
    r.x = __builtin_amdgcn_readfirstlane((unsigned)b);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
    r.z = __builtin_amdgcn_readfirstlane(recs);
    r.w = 0x00020000u;
    const unsigned dst = wave_dst + (unsigned)slot * SLOTB;
    sfor3(
        [&](auto ic) {
          constexpr int i = decltype(ic)::value;
          const unsigned e = pyx[i];
          const int iy = iy0 + (int)(e & 0xfffu), ix = ix0 + (int)((e >> 12) & 0xfffu);
          unsigned off = kOobOffset;
          if ((e & 0x80000000u) && (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w)
            off = ((unsigned)iy * (unsigned)a.w + (unsigned)ix) * pitchb + ((e >> 24) & 0xfu) * 16u;
          dma_one3(off, r, dst + i * 1024u);
        },
        std::make_integer_sequence<int, ND>{});
  };

  // ---- A fragments: row block mb = wm + j * WM; lane pixel (my, mx) of it; patch pixel of tap (dy, dx) = pb + dy * PWP + dx ----
  int pb[MPW];
#pragma unroll
  for (int j = 0; j < MPW; ++j) {
    const int mb = wm + j * WM;
    const int oy = (mb / TMX) * MBH + r32 / MBW, ox = (mb % TMX) * MBW + r32 % MBW;
    pb[j] = oy * S * PWP + ox * S;
  }
  float* tb = reinterpret_cast<float*>(smem3 + (unsigned)NSLOT * SLOTB) + wave * (32 * TP);
  const int prow = lane >> 2, c8 = (lane & 3) * 8;

  const int G = a.grid_m;
#pragma unroll
  for (int s = 0; s < NSLOT - 1; ++s) issue(b_m + s * G, s);

  int slot = 0;
  for (int t = b_m; t < a.tiles_total; t += G) {
    wait_vm3<(NSLOT - 2) * ND>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // ---- this tile's output pixels (transposed epilogue layout: lane -> pixel pass * 16 + prow of each row block) -----------
    const int nimg = t / per_img, rem = t - nimg * per_img;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    int mrow[MPW][2];  // dense pixel index (< 2^31: the launcher checks), -1 = outside the map
#pragma unroll
    for (int j = 0; j < MPW; ++j)
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int mb = wm + j * WM, q = pass * 16 + prow;
        const int oy = ty * TH + (mb / TMX) * MBH + q / MBW, ox = tx * TW + (mb % TMX) * MBW + q % MBW;
        mrow[j][pass] = (oy < a.ho && ox < a.wo) ? (nimg * a.ho + oy) * a.wo + ox : -1;
      }
    // ---- shortcut operand: plain loads requested before the tile's MFMAs (their latency hides behind them).  The compiler's
    // wait for them counts only the loads it knows, i.e. it is a vmcnt(0) that would also drain a patch refill issued in front
    // of it - so with a shortcut the refill is issued at the END of the iteration, behind the epilogue.  (A first version
    // requested the pieces by inline asm and waited by hand; the compiler is free to copy such registers between the two asm
    // statements - before the data has landed - and the detector's determinism stress test caught exactly that.)
    uint4 rres[NR];
    if (a.res) {
#pragma unroll
      for (int j = 0; j < MPW; ++j)
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
          const long long m = mrow[j][pass] >= 0 ? mrow[j][pass] : 0;  // (pixels outside the map: any readable address)
          rres[j * 2 + pass] = *reinterpret_cast<const uint4*>(a.res + m * a.res_pitch + n0 + c8);
        }
    } else {
      const int ps = slot == 0 ? NSLOT - 1 : slot - 1;
      issue(t + (NSLOT - 1) * G, ps);
    }
    const unsigned char* At = smem3 + (unsigned)slot * SLOTB;
    f32x16 acc[MPW];
#pragma unroll
    for (int j = 0; j < MPW; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      unsigned abase[MPW], asw[MPW];
#pragma unroll
      for (int j = 0; j < MPW; ++j) {
        const int pidx = pb[j] + (tap / 3) * PWP + (tap % 3);
        abase[j] = (unsigned)pidx * PIXB;
        asw[j] = (unsigned)((hh ^ swz(pidx)) * 16);  // chunk (2 sub + hh) ^ f = (2 sub) ^ (hh ^ f): hh is bit 0 of the chunk
      }
#pragma unroll
      for (int sub = 0; sub < KPT; ++sub)
#pragma unroll
        for (int j = 0; j < MPW; ++j) {
          const v8 af = *reinterpret_cast<const v8*>(At + abase[j] + (((unsigned)(sub * 32)) ^ asw[j]));
          acc[j] = H16<F16>::mfma(af, wf[tap * KPT + sub], acc[j]);
        }
    }
    // ---- epilogue ------------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < MPW; ++j) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = acc[j][e] * sc + sh;
        v = fmaxf(v, v * slope);
        tb[((e & 3) + 8 * (e >> 2) + 4 * hh) * TP + r32] = v;
      }
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int row = pass * 16 + prow;
        const float4 lo = *reinterpret_cast<const float4*>(tb + row * TP + c8);
        const float4 hi = *reinterpret_cast<const float4*>(tb + row * TP + c8 + 4);
        const long long m = mrow[j][pass];
        if (m >= 0) {
          float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
          if (a.res) {
            const uint4 r4 = rres[j * 2 + pass];
            const unsigned rr[4] = {r4.x, r4.y, r4.z, r4.w};
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
          me::store16(a.y + m * a.y_pitch + n0 + c8, o, a.store_mode);
        }
      }
    }
    if (a.res) {  // (see above: the refill of the slot tile t - G left, behind the shortcut's loads)
      const int ps = slot == 0 ? NSLOT - 1 : slot - 1;
      issue(t + (NSLOT - 1) * G, ps);
    }
    slot = slot + 1 == NSLOT ? 0 : slot + 1;
  }
  wait_vm3<0>();
}

template <int CIN, int WN, int WM, int MBH, int MBW, int TMY, int TMX, int S, int NSLOT>
int launch_ws3(const Conv16P& p, hipStream_t stream) {
  constexpr int NW = WN * WM, TH = TMY * MBH, TW = TMX * MBW;
  constexpr int PH = (TH - 1) * S + 3, PWP = (TW - 1) * S + 3;
  constexpr int PIECES = (PH * PWP * CIN * 2 + 1023) / 1024, ND = (PIECES + NW - 1) / NW;
  K3Args a;
  a.x = p.x; a.wgt = p.wgt; a.scale = p.scale; a.shift = p.shift;
  a.res = reinterpret_cast<const unsigned short*>(p.res);
  a.y = reinterpret_cast<unsigned short*>(p.y);
  a.x_pitch = p.x_pitch; a.res_pitch = p.res_pitch; a.y_pitch = p.y_pitch;
  a.n = p.n; a.h = p.h; a.w = p.w; a.ho = p.ho; a.wo = p.wo; a.cout = p.cout; a.act = p.act; a.store_mode = p.store_mode;
  a.tiles_y = (p.ho + TH - 1) / TH;
  a.tiles_x = (p.wo + TW - 1) / TW;
  const long long total = (long long)p.n * a.tiles_y * a.tiles_x;
  ME_REQUIRE(total < (1ll << 31) && (long long)p.n * p.ho * p.wo < (1ll << 31), ME_E_TOOBIG, "me_conv2d_h16: tile 60: too many tiles");
  a.tiles_total = (int)total;
  const int tiles_n = p.cout / (32 * WN);
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  const size_t lds = (size_t)NSLOT * ND * NW * 1024 + (size_t)NW * 32 * 36 * sizeof(float);
  ME_REQUIRE(lds <= 160 * 1024, ME_E_TOOBIG, "me_conv2d_h16: tile 60 needs %zu bytes of LDS", lds);
  int per_cu = (int)(160 * 1024 / lds);
  const int by_regs = 2 * 4 / NW;  // <= 256 VGPRs: two waves per SIMD
  if (per_cu > by_regs) per_cu = by_regs;
  if (per_cu < 1) per_cu = 1;
  int per_n = cus * per_cu / tiles_n;
  if (per_n < 1) per_n = 1;
  a.grid_m = a.tiles_total < per_n ? a.tiles_total : per_n;
  const dim3 grid((unsigned)(a.grid_m * tiles_n)), block(64 * NW);
  if (p.f16) {
    auto kern = conv3x3_ws_kernel<CIN, WN, WM, MBH, MBW, TMY, TMX, S, NSLOT, 1>;
    static bool attr = false;
    if (!attr) {
      ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr = true;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
  } else {
    auto kern = conv3x3_ws_kernel<CIN, WN, WM, MBH, MBW, TMY, TMX, S, NSLOT, 0>;
    static bool attr = false;
    if (!attr) {
      ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr = true;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
  }
  return me::check_launch("conv3x3_ws_h16");
}

}  // namespace

namespace me16 {

bool ws3x3_eligible(const Conv16P& p) {
  if (p.ks != 3 || p.pad != 1 || (p.stride != 1 && p.stride != 2) || p.ups != 1 || p.x_nchw || !p.vec_epi || p.y_f32) return false;
  if (p.x_pitch % 8 || !me::aligned16(p.x) || !me::aligned16(p.wgt)) return false;
  if ((long long)p.h * p.w * p.x_pitch * 2 >= (1ll << 31)) return false;
  if (p.h >= 4096 || p.w >= 4096) return false;
  const int c = p.cin, o = p.cout;
  return (c == 32 && o == 64) || (c == 64 && o == 128);
}

int launch_ws3x3(const Conv16P& p, hipStream_t stream) {
  ME_REQUIRE(ws3x3_eligible(p), ME_E_BADARG,
             "me_conv2d_h16: tile 60 (weight-stationary 3x3) needs a 3x3 / pad 1 / stride 1 or 2 layer with (cin, cout) = "
             "(32, 64) or (64, 128), 16-bit output with the 16-byte epilogue; got %d -> %d stride %d", p.cin, p.cout, p.stride);
  static const int variant = [] {
    const char* e = getenv("MILLIEYE_WS3_VARIANT");
    return e ? atoi(e) : 0;
  }();
  if (variant == 1 && p.cin == 64) {  // tuning: 4-wave workgroups, two per CU
    if (p.stride == 1) return launch_ws3<64, 4, 1, 4, 8, 2, 1, 1, 3>(p, stream);  // 8 x 8 tiles, 2 row blocks per wave
    return launch_ws3<64, 4, 1, 4, 8, 1, 1, 2, 3>(p, stream);                     // 4 x 8 tiles
  }
  if (variant == 2 && p.cin == 64) {
    if (p.stride == 1) return launch_ws3<64, 4, 2, 4, 8, 2, 1, 1, 4>(p, stream);  // 8 x 8 tiles, 8 waves, one workgroup per CU
    return launch_ws3<64, 4, 2, 4, 8, 2, 1, 2, 2>(p, stream);                     // 8 x 8 tiles, 8 waves
  }
  //                                     CIN WN WM MBH MBW TMY TMX S NSLOT
  if (p.cin == 32 && p.stride == 1) return launch_ws3<32, 2, 2, 2, 16, 4, 1, 1, 3>(p, stream);  //  8 x 16 tiles, patch 10 x 18
  if (p.cin == 32) return launch_ws3<32, 2, 2, 2, 16, 2, 1, 2, 3>(p, stream);                    //  4 x 16 tiles, patch  9 x 33
  if (p.stride == 1) return launch_ws3<64, 4, 1, 4, 8, 1, 1, 1, 4>(p, stream);                   //  4 x  8 tiles, patch  6 x 10, two
                                                                                                 //  4-wave workgroups per CU
  return launch_ws3<64, 4, 2, 4, 8, 1, 2, 2, 2>(p, stream);                                      //  4 x 16 tiles, patch  9 x 33
}

}  // namespace me16
