// conv_p8_f32.hip - the patch-resident big-tile 3x3 / stride-1 convolution (conv_p8_impl.h; design notes in
// conv_p8_h16.hip) instantiated for float32: v_mfma_f32_32x32x2_f32 - exact fp32 FMA chains, the numerics class of the CPU
// reference - on 64-byte chunks of 16 channels.  One ds_read_b128 (4 floats = 4 k values) feeds four MFMAs; A and B use the
// same permutation of the 16 channels of a chunk (lane half hh and chunk slot q pick channels 4 (2 q + hh) .. + 3), which a
// dot product does not see.  Why a second fp32 conv kernel: per (tile, 16-channel chunk) the input patch is brought into LDS
// once instead of nine times (the per-tap kernel of conv.hip moved 2.2x the algorithmic bytes: profiles/conv_traffic.json,
// round 1), one barrier per 8 MT NT MFMAs, and tiles are BM consecutive positions of the padded-linear index space - they
// cross rows and images, so there is no ragged last tile per image.
#include "conv32_common.h"
#include "conv_p8_impl.h"

namespace {
using namespace me_p8;

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct DT32 {
  using frag = f32x4;
  static constexpr int kBytes = 4, kChunk = 16;
  static __device__ __forceinline__ void mfma(const frag& a, const frag& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, c, 0, 0, 0);
  }
  static __device__ __forceinline__ frag fill(float v) { return frag{v, v + 0.01f, v + 0.02f, v + 0.03f}; }
  static __device__ __forceinline__ float from16(unsigned) { return 0.f; }      // 16-bit epilogue only
  static __device__ __forceinline__ unsigned pack2(float, float) { return 0u; }  // 16-bit epilogue only
};

void magic_u32(unsigned d, unsigned* m, unsigned* s) {
  unsigned sh = 0;
  while ((1ull << sh) < d) ++sh;
  *s = sh;
  *m = (unsigned)(((1ull << 32) * ((1ull << sh) - d)) / d + 1);
}

template <int WR, int WC, int MT, int NT, int PIPE, int MINB, int ABL = 0>
int launch_p8(const ConvP& p, hipStream_t stream) {
  constexpr int BM = 32 * MT * WR, BN = 32 * NT * WC, NWAVES = WR * WC;
  P8Args a = {};
  a.c.x = p.x; a.c.wgt_tiled = p.wgt_tiled; a.c.scale = p.scale; a.c.shift = p.shift; a.c.res = p.res; a.c.y = p.y;
  a.c.x_pitch = p.x_pitch; a.c.res_pitch = p.res_pitch; a.c.y_pitch = p.y_pitch;
  a.c.n = p.n; a.c.h = p.h; a.c.w = p.w; a.c.cin = p.cin; a.c.cout = p.cout; a.c.act = p.act;
  a.c.partial = p.partial;
  a.c.splitk = 1;  // (the K split of the patch tiles is built for the 16-bit epilogue only)
  a.c.cps = 0;
  a.Wp = p.w + 1;
  a.Ip = (p.h + 1) * a.Wp;
  a.halo = a.Wp + 1;
  a.Mp = (long long)p.n * a.Ip;
  a.rows = BM + 2 * a.halo;
  a.lpa = ((a.rows + 15) / 16 + NWAVES - 1) / NWAVES;
  ME_REQUIRE(a.lpa <= kLpaMax, ME_E_TOOBIG, "me_conv2d_f32: patch of %d rows does not fit (map too wide for this tile)", a.rows);
  ME_REQUIRE(a.Mp < (1ll << 31), ME_E_TOOBIG, "me_conv2d_f32: too many padded positions");
  magic_u32((unsigned)a.Ip, &a.ip_m, &a.ip_s);
  magic_u32((unsigned)a.Wp, &a.wp_m, &a.wp_s);
  a.c.tiles_m = (int)((a.Mp + BM - 1) / BM);
  a.c.tiles_n = p.cout / BN;
  const size_t lds = 3 * (size_t)BN * 64 + 2 * (size_t)a.lpa * NWAVES * 1024;
  ME_REQUIRE(lds <= (MINB == 2 ? 80 : 160) * 1024, ME_E_TOOBIG,
             "me_conv2d_f32: this tile needs %zu bytes of LDS for a %d-wide map", lds, p.w);
  auto kern = conv3x3_p8_kernel<WR, WC, MT, NT, PIPE, MINB, DT32, ABL>;
  static bool attr_set = false;
  if (!attr_set) {
    ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const long long blocks = (long long)a.c.tiles_m * a.c.tiles_n;
  ME_REQUIRE(blocks < (1ll << 31), ME_E_TOOBIG, "me_conv2d_f32: grid too large");
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * NWAVES), lds, stream, a);
  return me::check_launch("conv3x3_p8_f32");
}

bool p8_eligible(const ConvP& p, int tile) {
  if (p.ks != 3 || p.stride != 1 || p.pad != 1 || p.ups != 1 || p.x_nchw || p.cin % 16 || p.act == ME_ACT_SIGMOID) return false;
  if (!p.wgt_tiled || !me::aligned16(p.wgt_tiled) || p.splitk != 1) return false;
  const int bn = (tile % 10 == 1) ? 128 : 256;
  if (p.cout % bn) return false;
  if ((long long)p.n * (p.h + 1) * (p.w + 1) >= (1ll << 31)) return false;
  const long long img_bytes = (long long)p.h * p.w * p.x_pitch * 4;
  const long long span = (1024 / ((long long)(p.h + 1) * (p.w + 1))) + 2;
  return span * img_bytes < (1ll << 31);
}

}  // namespace

namespace me32 {

// tile ids (the same numbering as the 16-bit kernels): 1xx one 8-wave workgroup per CU with register-pipelined fragments,
// 2xx two 8-wave workgroups per CU, 3xx two independent 4-wave workgroups per CU; ...1 = 128 output channels per tile
int launch_p8_tile(const ConvP& p, int tile, hipStream_t stream) {
  ME_REQUIRE(p8_eligible(p, tile), ME_E_BADARG,
             "me_conv2d_f32: tile %d needs a 3x3 / stride 1 / pad 1 layer without upsample / sigmoid / split-K, cin %% 16 == 0, "
             "cout %% tile width == 0 and the tiled weight copy (wgt_tiled)", tile);
  switch (tile) {
    case 100: return launch_p8<2, 4, 2, 2, 1, 1>(p, stream);   // 128 x 256
    case 110: return launch_p8<2, 4, 3, 2, 1, 1>(p, stream);   // 192 x 256
    case 121: return launch_p8<4, 2, 2, 2, 1, 1>(p, stream);   // 256 x 128
    case 131: return launch_p8<4, 2, 3, 2, 1, 1>(p, stream);   // 384 x 128
    case 200: return launch_p8<2, 4, 2, 2, 0, 2>(p, stream);   // 128 x 256
    case 201: return launch_p8<4, 2, 1, 2, 0, 2>(p, stream);   // 128 x 128
    case 221: return launch_p8<4, 2, 2, 2, 0, 2>(p, stream);   // 256 x 128
    case 311: return launch_p8<2, 2, 3, 2, 1, 2>(p, stream);   // 192 x 128
    case 321: return launch_p8<2, 2, 4, 2, 1, 2>(p, stream);   // 256 x 128
    default: ME_REQUIRE(false, ME_E_BADARG, "me_conv2d_f32: unknown patch tile id %d", tile);
  }
  return 0;
}

}  // namespace me32
