// optim.hip - the optimizer step of the training loops as ONE launch over every parameter tensor (gfx950).
//
// Replaces optimizer.step() of the reference's loops: Adam(lr=5e-4) of stage 3 (module3_our_dataset/train.py:161,196-197) and
// AdamW(lr=1e-4) of stage 2 (module2/train.py:122,150-151).  The stage-3 state is 40 tensors / 100 153 parameters: torch's fused
// Adam took 96 us per step on it (one workgroup per tensor chunk of 65 536 elements and pow(beta, step) on the device per
// element), this launch is bound by its launch latency.  Per element, in fp32 and in the order of torch/optim/adam.py
// _single_tensor_adam (no contraction: every product and sum is rounded on its own, like the sequence of tensor ops):
//     AdamW:  p = p * (1 - lr * wd)                    Adam with weight decay:  g = g + wd * p
//     m = m + (1 - beta1) * (g - m)                    (Tensor.lerp_, weight < 0.5)
//     v = v * beta2;  v = v + ((1 - beta2) * g) * g    (mul_, addcmul_: value * t1 * t2 from the left)
//     denom = sqrt(v) / sqrt(1 - beta2^t) + eps
//     p = p + (-(lr / (1 - beta1^t)) * m) / denom      (addcdiv_: value * t1 / t2 from the left)
// Every scalar in brackets is computed by the caller in double (Python floats in the torch class) and rounded to fp32 once - in
// particular 1 - beta2: 1.f - 0.999f is 0.00100005, not fp32(0.001).
#include <math.h>
#include "common.h"

namespace {

constexpr int ADAM_CHUNK = 1024;  // elements per workgroup: 256 threads x 4

__global__ __launch_bounds__(256) void adam_multi_kernel(me_adam_desc d) {
#pragma clang fp contract(off)
  // workgroup -> (tensor, chunk): first_chunk[] is the exclusive prefix sum of the tensors' chunk counts (<= 64 entries)
  const int bid = blockIdx.x;
  int t = 0;
  while (t + 1 < d.count && bid >= d.first_chunk[t + 1]) ++t;
  const long long base = (long long)(bid - d.first_chunk[t]) * ADAM_CHUNK;
  const long long numel = d.numel[t];
  float* __restrict__ p = d.param[t];
  const float* __restrict__ g = d.grad[t];
  float* __restrict__ m = d.exp_avg[t];
  float* __restrict__ v = d.exp_avg_sq[t];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long i = base + threadIdx.x + 256 * u;
    if (i >= numel) break;
    float pi = p[i], gi = g[i], mi = m[i], vi = v[i];
    if (d.weight_decay != 0.f) {
      if (d.decoupled) pi = pi * d.decay;
      else gi = gi + d.weight_decay * pi;
    }
    mi = mi + d.one_minus_beta1 * (gi - mi);
    vi = vi * d.beta2;
    vi = vi + (d.one_minus_beta2 * gi) * gi;
    const float denom = sqrtf(vi) / d.bias_correction2_sqrt + d.eps;
    pi = pi + (d.neg_step_size * mi) / denom;
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
  }
}

}  // namespace

extern "C" {

int me_adam_step_f32(const me_adam_desc* d, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(d != nullptr, ME_E_NULLPTR, "me_adam_step_f32: null descriptor");
  ME_REQUIRE(d->count >= 0 && d->count <= ME_ADAM_MAX_TENSORS, ME_E_BADARG, "me_adam_step_f32: %d tensors per launch (max %d)",
             d->count, ME_ADAM_MAX_TENSORS);
  ME_REQUIRE(d->beta2 >= 0.f && d->beta2 < 1.f && d->eps >= 0.f && d->weight_decay >= 0.f && d->neg_step_size <= 0.f, ME_E_BADARG,
             "me_adam_step_f32: hyper-parameters out of range");
  ME_REQUIRE(d->one_minus_beta1 > 0.f && d->one_minus_beta1 < 0.5f, ME_E_BADARG,
             "me_adam_step_f32: 1 - beta1 must lie in (0, 0.5) (Tensor.lerp_ switches to its other formula at 0.5; not implemented)");
  ME_REQUIRE(d->bias_correction2_sqrt > 0.f, ME_E_BADARG, "me_adam_step_f32: bias_correction2_sqrt must be positive (step >= 1)");
  if (d->count == 0) return 0;
  long long chunks = 0;
  for (int t = 0; t < d->count; ++t) {
    ME_REQUIRE(d->numel[t] >= 0, ME_E_BADARG, "me_adam_step_f32: negative element count");
    ME_REQUIRE(d->numel[t] == 0 || (d->param[t] && d->grad[t] && d->exp_avg[t] && d->exp_avg_sq[t]), ME_E_NULLPTR,
               "me_adam_step_f32: null pointer (tensor %d)", t);
    ME_REQUIRE(d->first_chunk[t] == chunks, ME_E_BADARG, "me_adam_step_f32: first_chunk[%d] is not the prefix sum of the chunk counts", t);
    chunks += (d->numel[t] + ADAM_CHUNK - 1) / ADAM_CHUNK;
  }
  ME_REQUIRE(chunks <= 0x7FFFFFFFll, ME_E_TOOBIG, "me_adam_step_f32: too many elements for one launch");
  if (chunks == 0) return 0;
  hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)chunks), dim3(256), 0, stream, *d);
  return me::check_launch("adam_multi_kernel");
}

int32_t me_adam_chunk(void) { return ADAM_CHUNK; }

}  // extern "C"
