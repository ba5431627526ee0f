// pack.hip - one launch that turns a convolution block's parameters into everything the fp32 kernels read (gfx950).
//
// A training step changes every weight, so the packed device copies of millieye_amd/engine.py:ConvWeights are rebuilt once
// per layer and step.  Through torch that was ~25 launches per layer (permute + contiguous, the tiled copy, the 180-degree
// rotation for the data gradient, six fp64 element-wise kernels for the BatchNorm fold, copies into the stable buffers):
// ~1900 launches and ~10 ms of a 42 ms Darknet-53 step (profiles/r03_bench_detector_train_b8_kernel_stats.txt).  Here:
//
//   ohwi   [cout][k][k][cin]            the B operand of the implicit GEMM (me_conv_desc.wgt)
//   tiled  [k*k][cin/16][cout][16]      me_conv_desc.wgt_tiled (cin % 16 == 0), optional
//   rot    [cin][k][k][cout]            weights of the data gradient: 180-degree rotated, channels transposed, optional
//   rott   [k*k][cout/16][cin][16]      tiled copy of rot (cout % 16 == 0), optional
//   scale, shift [cout]                 BatchNorm(eval) + bias folded in double precision exactly like the host code did:
//                                       scale = gamma / sqrt(var + eps), shift = beta - mean * scale (+ bias * scale);
//                                       without BatchNorm scale = 1, shift = bias (or 0)
//
// One workgroup = 16 output channels x 64 input channels x all taps, staged in LDS so that every destination is written in
// runs of >= 64 bytes.  Reference: the parameter layout of nn.Conv2d / nn.BatchNorm2d (module3_our_dataset/yolov3/models.py:22-41).
#include <math.h>

#include "common.h"

namespace {

struct PackArgs {
  const float* w;  // [cout][cin][k][k]
  const float* bias;
  const float* gamma;
  const float* beta;
  const float* mean;
  const float* var;
  float* ohwi;
  float* tiled;
  float* rot;
  float* rott;
  float* scale;
  float* shift;
  int cout, cin, k;
  float eps;
};

constexpr int PO = 16, PC = 64;

__global__ __launch_bounds__(256) void pack_conv_kernel(PackArgs a) {
  extern __shared__ float s_w[];  // [PO][PC][kk]
  const int kk = a.k * a.k;
  const int o0 = blockIdx.y * PO, c0 = blockIdx.x * PC;
  const int no = a.cout - o0 < PO ? a.cout - o0 : PO, nc = a.cin - c0 < PC ? a.cin - c0 : PC;
  const int t = threadIdx.x;
  // load: for each o the nc * kk floats of channels c0 .. c0 + nc are contiguous in OIHW
  const int run = nc * kk;
  for (int i = t; i < no * run; i += 256) {
    const int o = i / run, r = i - o * run;
    s_w[o * (PC * kk) + r] = a.w[((long long)(o0 + o) * a.cin + c0) * kk + r];
  }
  __syncthreads();
  // ohwi[o][tap][c]: runs of nc floats
  for (int i = t; i < no * kk * nc; i += 256) {
    const int c = i % nc, r = i / nc;
    const int tap = r % kk, o = r / kk;
    a.ohwi[((long long)(o0 + o) * kk + tap) * a.cin + c0 + c] = s_w[o * (PC * kk) + c * kk + tap];
  }
  if (a.tiled) {  // [tap][chunk][o][16]: runs of 16 floats per o, no * 16 contiguous per (tap, chunk)
    const int chunks = nc / 16;  // cin % 16 == 0 (checked by the launcher): nc is a multiple of 16
    for (int i = t; i < kk * chunks * no * 16; i += 256) {
      const int l = i & 15, r = i >> 4;
      const int o = r % no, r2 = r / no;
      const int ch = r2 % chunks, tap = r2 / chunks;
      a.tiled[(((long long)tap * (a.cin / 16) + c0 / 16 + ch) * a.cout + o0 + o) * 16 + l] =
          s_w[o * (PC * kk) + (ch * 16 + l) * kk + tap];
    }
  }
  if (a.rot) {  // rot[c][kk - 1 - tap][o]: runs of no floats
    for (int i = t; i < nc * kk * no; i += 256) {
      const int o = i % no, r = i / no;
      const int tap = r % kk, c = r / kk;
      a.rot[((long long)(c0 + c) * kk + (kk - 1 - tap)) * a.cout + o0 + o] = s_w[o * (PC * kk) + c * kk + tap];
    }
  }
  if (a.rott) {  // rott[kk - 1 - tap][o / 16][c][o % 16]: cout % 16 == 0 -> this block is one o-chunk: nc * 16 contiguous per tap
    for (int i = t; i < kk * nc * 16; i += 256) {
      const int l = i & 15, r = i >> 4;
      const int c = r % nc, tap = r / nc;
      a.rott[(((long long)(kk - 1 - tap) * (a.cout / 16) + o0 / 16) * a.cin + c0 + c) * 16 + l] =
          s_w[l * (PC * kk) + c * kk + tap];
    }
  }
  if (blockIdx.x == 0 && t < no) {  // the fold, in double like ConvWeights.refresh did on the host side of torch
    const int o = o0 + t;
    double sc = 1.0, sh = 0.0;
    if (a.gamma) {
      sc = (double)a.gamma[o] / sqrt((double)a.var[o] + (double)a.eps);
      sh = (double)a.beta[o] - (double)a.mean[o] * sc;
      if (a.bias) sh += (double)a.bias[o] * sc;
    } else if (a.bias) {
      sh = (double)a.bias[o];
    }
    a.scale[o] = (float)sc;
    a.shift[o] = (float)sh;
  }
}

}  // namespace

extern "C" {

int me_pack_conv_f32(const float* w_oihw, int32_t cout, int32_t cin, int32_t ksize, const float* bias, const float* gamma,
                     const float* beta, const float* mean, const float* var, float eps, float* ohwi, float* tiled,
                     float* rot, float* rot_tiled, float* scale, float* shift, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(w_oihw && ohwi && scale && shift, ME_E_NULLPTR, "me_pack_conv_f32: null pointer");
  ME_REQUIRE(cout > 0 && cin > 0 && ksize >= 1 && ksize <= 7, ME_E_BADARG, "me_pack_conv_f32: bad dimensions");
  ME_REQUIRE(!gamma || (beta && mean && var), ME_E_NULLPTR, "me_pack_conv_f32: BatchNorm needs gamma, beta, mean and var");
  ME_REQUIRE(!tiled || cin % 16 == 0, ME_E_BADARG, "me_pack_conv_f32: the tiled copy needs cin %% 16 == 0");
  ME_REQUIRE(!rot_tiled || (cout % 16 == 0 && rot), ME_E_BADARG, "me_pack_conv_f32: the rotated tiled copy needs cout %% 16 == 0");
  PackArgs a;
  a.w = w_oihw; a.bias = bias; a.gamma = gamma; a.beta = beta; a.mean = mean; a.var = var;
  a.ohwi = ohwi; a.tiled = tiled; a.rot = rot; a.rott = rot_tiled; a.scale = scale; a.shift = shift;
  a.cout = cout; a.cin = cin; a.k = ksize; a.eps = eps;
  const size_t lds = (size_t)PO * PC * ksize * ksize * sizeof(float);
  ME_REQUIRE(lds <= 64 * 1024, ME_E_TOOBIG, "me_pack_conv_f32: filter too large");
  const long long gy = (cout + PO - 1) / PO;
  ME_REQUIRE(gy < 65536, ME_E_TOOBIG, "me_pack_conv_f32: too many output channels");
  hipLaunchKernelGGL(pack_conv_kernel, dim3((cin + PC - 1) / PC, (unsigned)gy), dim3(256), lds, stream, a);
  return me::check_launch("pack_conv_kernel");
}

}  // extern "C"
