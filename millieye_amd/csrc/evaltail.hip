// Evaluation tail on the device: get_batch_statistics of
// module3_our_dataset/utils/utils.py:185-236 for a whole batch of Network.forward output rows - the greedy
// true-positive assignment the reference runs as a python loop over every detection (test_fusion.py:98-100).
// Per image (one wave): detections in row order; a detection whose label occurs among the image's targets takes the
// target of maximal IoU (+1 pixel convention of bbox_iou, utils.py:248-278, first maximum on ties, *any* label - the
// label test of the reference is commented out) and is a true positive iff IoU >= thr and that target is still free;
// the scan stops once every target is taken.  float32, one rounding per operation (no FMA contraction): bit-identical
// decisions to the host code on the same inputs.
#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr int kMaxTargets = 8192;

__global__ __launch_bounds__(64) void batch_statistics_kernel(const float* __restrict__ rows, int m, int cols,
                                                              const float* __restrict__ targets, int q, float thr,
                                                              float* __restrict__ tp) {
  __shared__ unsigned taken[kMaxTargets / 32];
  const int img = blockIdx.x, lane = threadIdx.x;
  for (int i = lane; i < kMaxTargets / 32; i += 64) taken[i] = 0u;
  int n_ann = 0;
  for (int t = lane; t < q; t += 64) n_ann += (targets[(size_t)t * 6] == (float)img) ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) n_ann += __shfl_xor(n_ann, o);
  __syncthreads();
  int n_taken = 0;
  for (int r = 0; r < m; ++r) {  // wave-uniform scan in row order
    const float* row = rows + (size_t)r * cols;
    if (row[0] != (float)img) continue;
    if (lane == 0) tp[r] = 0.f;
    if (n_ann == 0 || n_taken == n_ann) continue;
    const float x1 = row[1], y1 = row[2], x2 = row[3], y2 = row[4], label = row[cols - 1];
    const float area1 = (x2 - x1 + 1.f) * (y2 - y1 + 1.f);
    float best = -1.f;
    int best_t = 0x7fffffff;
    int label_seen = 0;
    for (int t = lane; t < q; t += 64) {
      const float* g = targets + (size_t)t * 6;
      if (g[0] != (float)img) continue;
      label_seen |= (g[1] == label) ? 1 : 0;
      const float ix1 = fmaxf(x1, g[2]), iy1 = fmaxf(y1, g[3]), ix2 = fminf(x2, g[4]), iy2 = fminf(y2, g[5]);
      const float inter = fmaxf(ix2 - ix1 + 1.f, 0.f) * fmaxf(iy2 - iy1 + 1.f, 0.f);
      const float area2 = (g[4] - g[2] + 1.f) * (g[5] - g[3] + 1.f);
      const float iou = inter / (area1 + area2 - inter + 1e-16f);
      if (iou > best) {  // strictly greater: the first maximum wins inside a lane (t increases)
        best = iou;
        best_t = t;
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o);
      const int ot = __shfl_xor(best_t, o);
      label_seen |= __shfl_xor(label_seen, o);
      if (ob > best || (ob == best && ot < best_t)) {
        best = ob;
        best_t = ot;
      }
    }
    if (!label_seen) continue;
    const bool free_slot = !((taken[best_t >> 5] >> (best_t & 31)) & 1u);
    if (best >= thr && free_slot) {
      if (lane == 0) {
        tp[r] = 1.f;
        taken[best_t >> 5] |= 1u << (best_t & 31);
      }
      ++n_taken;
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int me_batch_statistics_f32(const float* rows, int32_t m, int32_t cols, const float* targets, int32_t q,
                                       int32_t n_images, float iou_threshold, float* tp, void* stream) {
  if (m == 0 || n_images == 0) return 0;
  ME_REQUIRE(rows && tp, ME_E_NULLPTR, "me_batch_statistics_f32: null pointer");
  ME_REQUIRE(m > 0 && cols >= 6 && n_images > 0 && q >= 0, ME_E_BADARG, "me_batch_statistics_f32: bad sizes");
  ME_REQUIRE(q == 0 || targets, ME_E_NULLPTR, "me_batch_statistics_f32: null targets");
  ME_REQUIRE(q <= kMaxTargets, ME_E_TOOBIG, "me_batch_statistics_f32: %d targets > %d", q, kMaxTargets);
  hipLaunchKernelGGL(batch_statistics_kernel, dim3(n_images), dim3(64), 0, (hipStream_t)stream, rows, m, cols, targets, q,
                     iou_threshold, tp);
  return me::check_launch("batch_statistics_kernel");
}
