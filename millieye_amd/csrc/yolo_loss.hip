// yolo_loss.hip - the YOLO loss of one detection scale on the device (gfx950): build_targets + the six loss terms + the
// metrics of module3_our_dataset/yolov3/models.py:181-232 and utils/utils.py:381-440.
//
// The reference (and millieye_amd/yolov3/models.py:loss_from_raw on CPU tensors) runs this as ~200 torch ops per scale - masked
// assignments, boolean-mask gathers (a nonzero + host sync each), thirteen .item() reads; on the training step of Darknet-53
// that was ~600 launches and ~2 ms of a 25 ms step for a few KB of targets.  Here: three launches per scale and ONE read-back.
//
//   yolo_targets_init      dense [N,A,G,G] target tensors: obj = 0, noobj = 1, tx / ty / tw / th / class_mask / iou_scores = 0,
//                          tcls [N,A,G,G,C] = 0
//   yolo_targets_scatter   one thread per target: grid cell, best anchor by shape IoU (first maximum), masks, regression
//                          targets, one-hot class, class_mask, IoU of the decoded prediction of that cell.  Two targets that
//                          own the same (image, anchor, cell): the LATER one wins, as in the reference's CPU index_put_
//                          (the CUDA index_put_ of the reference leaves the winner undefined); tcls keeps both labels.
//   yolo_loss_reduce       every cell: the conf terms of all cells, the x / y / w / h / cls terms of the object cells, the
//                          metric sums; per-block partial sums in double, added in block order by the last block
//                          (ticket) -> fixed order, deterministic.
//
// The dense tensors are what me_yolo_loss_bwd_f32 (train.hip) takes; result[16] (float) is read by the host once - or not at all:
// me_yolo_loss_fwd_counted_f32 + me_yolo_loss_bwd_dev_f32 keep the target count and n_obj / n_noobj in device memory, for the
// captured training step (millieye_amd/detector_graph.py).
// Arithmetic: float32 like the reference (sigmoid = 1 / (1 + exp(-x)), BCE with its log clamp at -100); sums in double.
#include <math.h>

#include "common.h"

namespace {

constexpr int YL_SUMS = 16;
// result layout
enum { R_LOSS, R_X, R_Y, R_W, R_H, R_CONF, R_CLS, R_CLS_ACC, R_RECALL50, R_RECALL75, R_PRECISION, R_CONF_OBJ, R_CONF_NOOBJ,
       R_N_OBJ, R_N_NOOBJ, R_BAD };
// partial-sum layout
enum { S_X, S_Y, S_W, S_H, S_COBJ, S_CNOOBJ, S_CLS, S_NOBJ, S_NNOOBJ, S_CLSMASK, S_R50, S_R75, S_CONF50, S_CONFOBJ, S_CONFNOOBJ,
       S_UNUSED };

struct YoloLossArgs {
  const float* raw;  // [N,G,G,A*(5+C)] NHWC, pitch floats per pixel
  long long pitch;
  const float* targets;  // [m,6] (image, class, cx, cy, w, h) in [0,1]
  const int* m_dev;      // or NULL; device word: how many of the m rows are targets (me_yolo_loss_fwd_counted_f32)
  int m, n, g, na, nc;
  float anchors[32];  // scaled anchors (w, h) in grid units, na <= 16
  float ignore_thres, obj_scale, noobj_scale;
  unsigned char* obj;
  unsigned char* noobj;
  float *tx, *ty, *tw, *th, *tcls, *tconf, *class_mask, *iou_scores;
  double* partials;   // [blocks][YL_SUMS]
  unsigned int* ticket;  // zero on entry, left zero
  int* bad;              // set when a target falls outside the grid / batch (the reference raises IndexError)
  float* result;         // [16]
};

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void yolo_targets_init_kernel(YoloLossArgs a) {
  const long long cells = (long long)a.n * a.na * a.g * a.g;
  const long long total = cells * a.nc;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    a.tcls[i] = 0.f;
    if (i < cells) {
      a.obj[i] = 0;
      a.noobj[i] = 1;
      a.tx[i] = a.ty[i] = a.tw[i] = a.th[i] = 0.f;
      a.class_mask[i] = a.iou_scores[i] = 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void yolo_targets_scatter_kernel(YoloLossArgs a) {
#pragma clang fp contract(off)
  const int t = blockIdx.x * 256 + threadIdx.x;
  int m = a.m;
  if (a.m_dev) {  // a fixed-capacity table (a captured launch): the device word says how many rows count, clamped to [0, m]
    const int live = *a.m_dev;
    m = live < 0 ? 0 : (live < m ? live : m);
  }
  if (t >= m) return;
  const int per = a.nc + 5;
  auto owner_of = [&](int k, int& b, int& best, int& gi, int& gj, float& gx, float& gy, float& gw, float& gh) {
    const float* row = a.targets + (long long)k * 6;
    b = (int)row[0];
    gx = row[2] * (float)a.g;
    gy = row[3] * (float)a.g;
    gw = row[4] * (float)a.g;
    gh = row[5] * (float)a.g;
    float best_iou = -1.f;
    best = 0;
    for (int an = 0; an < a.na; ++an) {  // bbox_wh_iou (utils.py:163-170), first maximum
      const float w1 = a.anchors[2 * an], h1 = a.anchors[2 * an + 1];
      const float inter = fminf(w1, gw) * fminf(h1, gh);
      const float uni = (w1 * h1 + 1e-16f) + gw * gh - inter;
      const float iou = inter / uni;
      if (iou > best_iou) {
        best_iou = iou;
        best = an;
      }
    }
    gi = (int)gx;  // .long(): truncation
    gj = (int)gy;
  };
  int b, best, gi, gj;
  float gx, gy, gw, gh;
  owner_of(t, b, best, gi, gj, gx, gy, gw, gh);
  if (b < 0 || b >= a.n || gi < 0 || gi >= a.g || gj < 0 || gj >= a.g) {
    *a.bad = 1;
    return;
  }
  const long long cell = (((long long)b * a.na + best) * a.g + gj) * a.g + gi;
  a.obj[cell] = 1;
  a.noobj[cell] = 0;
  // anchors whose shape fits above the threshold are neither object nor background (utils.py:422-424)
  for (int an = 0; an < a.na; ++an) {
    const float w1 = a.anchors[2 * an], h1 = a.anchors[2 * an + 1];
    const float inter = fminf(w1, gw) * fminf(h1, gh);
    const float uni = (w1 * h1 + 1e-16f) + gw * gh - inter;
    if (inter / uni > a.ignore_thres) a.noobj[(((long long)b * a.na + an) * a.g + gj) * a.g + gi] = 0;
  }
  const int label = (int)a.targets[(long long)t * 6 + 1];
  if (label >= 0 && label < a.nc) a.tcls[cell * a.nc + label] = 1.f; else *a.bad = 1;
  // a later target with the same owner cell overwrites the scalar targets (sequential index_put_ semantics)
  for (int k = t + 1; k < m; ++k) {
    int b2, best2, gi2, gj2;
    float x2, y2, w2, h2;
    owner_of(k, b2, best2, gi2, gj2, x2, y2, w2, h2);
    if (b2 == b && best2 == best && gi2 == gi && gj2 == gj) return;
  }
  a.tx[cell] = gx - floorf(gx);
  a.ty[cell] = gy - floorf(gy);
  a.tw[cell] = logf(gw / a.anchors[2 * best] + 1e-16f);
  a.th[cell] = logf(gh / a.anchors[2 * best + 1] + 1e-16f);
  const float* r = a.raw + ((long long)(b * a.g + gj) * a.g + gi) * a.pitch + best * per;
  // class_mask: argmax over the class probabilities (first maximum) == label
  int arg = 0;
  float top = sigmoidf(r[5]);
  for (int c = 1; c < a.nc; ++c) {
    const float p = sigmoidf(r[5 + c]);
    if (p > top) {
      top = p;
      arg = c;
    }
  }
  a.class_mask[cell] = arg == label ? 1.f : 0.f;
  // iou_scores: bbox_iou(pred_box, target_box, x1y1x2y2=False) (utils.py:173-200), boxes in grid units
  const float px = sigmoidf(r[0]) + (float)gi, py = sigmoidf(r[1]) + (float)gj;
  const float pw = expf(r[2]) * a.anchors[2 * best], ph = expf(r[3]) * a.anchors[2 * best + 1];
  const float b1x1 = px - pw / 2, b1x2 = px + pw / 2, b1y1 = py - ph / 2, b1y2 = py + ph / 2;
  const float b2x1 = gx - gw / 2, b2x2 = gx + gw / 2, b2y1 = gy - gh / 2, b2y2 = gy + gh / 2;
  const float ix1 = fmaxf(b1x1, b2x1), iy1 = fmaxf(b1y1, b2y1), ix2 = fminf(b1x2, b2x2), iy2 = fminf(b1y2, b2y2);
  const float inter = fmaxf(ix2 - ix1 + 1.f, 0.f) * fmaxf(iy2 - iy1 + 1.f, 0.f);
  const float a1 = (b1x2 - b1x1 + 1.f) * (b1y2 - b1y1 + 1.f), a2 = (b2x2 - b2x1 + 1.f) * (b2y2 - b2y1 + 1.f);
  a.iou_scores[cell] = inter / (a1 + a2 - inter + 1e-16f);
}

__device__ __forceinline__ float bce(float p, float t) {  // nn.BCELoss element: logs clamped at -100
  const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(logf(1.f - p), -100.f);
  return -(t * lp + (1.f - t) * lq);
}

__global__ __launch_bounds__(256) void yolo_loss_reduce_kernel(YoloLossArgs a) {
  __shared__ double s_sum[4][YL_SUMS];
  __shared__ bool s_last;
  const int per = a.nc + 5;
  const long long cells = (long long)a.n * a.na * a.g * a.g;
  double acc[YL_SUMS];
#pragma unroll
  for (int k = 0; k < YL_SUMS; ++k) acc[k] = 0.0;
  for (long long cell = (long long)blockIdx.x * 256 + threadIdx.x; cell < cells; cell += (long long)gridDim.x * 256) {
    long long t = cell;
    const int gx = (int)(t % a.g);
    t /= a.g;
    const int gy = (int)(t % a.g);
    t /= a.g;
    const int an = (int)(t % a.na);
    const int img = (int)(t / a.na);
    const float* r = a.raw + ((long long)(img * a.g + gy) * a.g + gx) * a.pitch + an * per;
    const bool is_obj = a.obj[cell] != 0, is_noobj = a.noobj[cell] != 0;
    const float pc = sigmoidf(r[4]);
    a.tconf[cell] = is_obj ? 1.f : 0.f;
    if (pc > 0.5f) acc[S_CONF50] += 1.0;
    if (is_noobj) {
      acc[S_CNOOBJ] += bce(pc, 0.f);
      acc[S_NNOOBJ] += 1.0;
      acc[S_CONFNOOBJ] += pc;
    }
    if (is_obj) {
      const float dx = sigmoidf(r[0]) - a.tx[cell], dy = sigmoidf(r[1]) - a.ty[cell];
      const float dw = r[2] - a.tw[cell], dh = r[3] - a.th[cell];
      acc[S_X] += dx * dx;
      acc[S_Y] += dy * dy;
      acc[S_W] += dw * dw;
      acc[S_H] += dh * dh;
      acc[S_COBJ] += bce(pc, 1.f);
      acc[S_NOBJ] += 1.0;
      acc[S_CONFOBJ] += pc;
      double cls = 0.0;
      for (int c = 0; c < a.nc; ++c) cls += bce(sigmoidf(r[5 + c]), a.tcls[cell * a.nc + c]);
      acc[S_CLS] += cls;
      const float cm = a.class_mask[cell];
      acc[S_CLSMASK] += cm;
      const float detected = (pc > 0.5f ? 1.f : 0.f) * cm;  // * tconf (= 1 here)
      const float iou = a.iou_scores[cell];
      if (iou > 0.5f) acc[S_R50] += detected;
      if (iou > 0.75f) acc[S_R75] += detected;
    }
  }
  // workgroup sum: wave shuffle, then the four waves in order
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < YL_SUMS; ++k) {
    double v = acc[k];
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) s_sum[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < YL_SUMS)
    a.partials[(long long)blockIdx.x * YL_SUMS + threadIdx.x] =
        ((s_sum[0][threadIdx.x] + s_sum[1][threadIdx.x]) + s_sum[2][threadIdx.x]) + s_sum[3][threadIdx.x];
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(a.ticket, 1u) == gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  // the last block adds the partials in block order and forms the reference's scalars in float32
  // (16 lanes per sum, each over the blocks lane, lane + 16, ..., then the lanes in order: one thread per sum walked up to
  //  256 dependent loads - 54 us per scale)
  __shared__ double s_lane[16][YL_SUMS];
  __shared__ double s_tot[YL_SUMS];
  {
    const int k = threadIdx.x & 15, ln = threadIdx.x >> 4;
    double v = 0.0;
    for (unsigned b = ln; b < gridDim.x; b += 16) v += a.partials[(long long)b * YL_SUMS + k];
    s_lane[ln][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < YL_SUMS) {
    double v = 0.0;
#pragma unroll
    for (int ln = 0; ln < 16; ++ln) v += s_lane[ln][threadIdx.x];
    s_tot[threadIdx.x] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float n_obj = (float)s_tot[S_NOBJ], n_noobj = (float)s_tot[S_NNOOBJ];
    const float lx = (float)s_tot[S_X] / n_obj, ly = (float)s_tot[S_Y] / n_obj;
    const float lw = (float)s_tot[S_W] / n_obj, lh = (float)s_tot[S_H] / n_obj;
    const float lco = (float)s_tot[S_COBJ] / n_obj, lcn = (float)s_tot[S_CNOOBJ] / n_noobj;
    const float lconf = a.obj_scale * lco + a.noobj_scale * lcn;
    const float lcls = (float)s_tot[S_CLS] / (n_obj * (float)a.nc);
    float* R = a.result;
    R[R_X] = lx; R[R_Y] = ly; R[R_W] = lw; R[R_H] = lh; R[R_CONF] = lconf; R[R_CLS] = lcls;
    R[R_LOSS] = ((((lx + ly) + lw) + lh) + lconf) + lcls;
    R[R_CLS_ACC] = 100.f * ((float)s_tot[S_CLSMASK] / n_obj);
    R[R_RECALL50] = (float)s_tot[S_R50] / (n_obj + 1e-16f);
    R[R_RECALL75] = (float)s_tot[S_R75] / (n_obj + 1e-16f);
    R[R_PRECISION] = (float)s_tot[S_R50] / ((float)s_tot[S_CONF50] + 1e-16f);
    R[R_CONF_OBJ] = (float)s_tot[S_CONFOBJ] / n_obj;
    R[R_CONF_NOOBJ] = (float)s_tot[S_CONFNOOBJ] / n_noobj;
    R[R_N_OBJ] = n_obj;
    R[R_N_NOOBJ] = n_noobj;
    R[R_BAD] = *a.bad ? 1.f : 0.f;
    *a.ticket = 0;
    *a.bad = 0;
  }
}

constexpr int YL_BLOCKS = 256;

}  // namespace

extern "C" {

int64_t me_yolo_loss_workspace_bytes(void) { return (int64_t)YL_BLOCKS * YL_SUMS * sizeof(double) + 256; }

static int yolo_loss_fwd(const float* raw, int64_t pitch, int32_t n, int32_t g, int32_t num_anchors, int32_t num_classes,
                         const float* scaled_anchors_host, const float* targets, int32_t m, const int32_t* m_device,
                         float ignore_thres, float obj_scale, float noobj_scale, uint8_t* obj_mask, uint8_t* noobj_mask, float* tx,
                         float* ty, float* tw, float* th, float* tcls, float* tconf, float* class_mask, float* iou_scores,
                         void* workspace, float* result, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(raw && scaled_anchors_host && obj_mask && noobj_mask && tx && ty && tw && th && tcls && tconf && class_mask &&
                 iou_scores && workspace && result && (m == 0 || targets),
             ME_E_NULLPTR, "me_yolo_loss_fwd_f32: null pointer");
  ME_REQUIRE(n > 0 && g > 0 && num_anchors > 0 && num_anchors <= 16 && num_classes > 0 && m >= 0 && pitch >= num_anchors * (num_classes + 5),
             ME_E_BADARG, "me_yolo_loss_fwd_f32: bad dimensions (at most 16 anchors per scale)");
  ME_REQUIRE(me::aligned16(workspace), ME_E_ALIGN, "me_yolo_loss_fwd_f32: workspace must be 16-byte aligned");
  YoloLossArgs a;
  a.raw = raw; a.pitch = pitch; a.targets = targets; a.m_dev = m_device; a.m = m; a.n = n; a.g = g; a.na = num_anchors; a.nc = num_classes;
  for (int i = 0; i < 2 * num_anchors; ++i) a.anchors[i] = scaled_anchors_host[i];
  a.ignore_thres = ignore_thres; a.obj_scale = obj_scale; a.noobj_scale = noobj_scale;
  a.obj = obj_mask; a.noobj = noobj_mask; a.tx = tx; a.ty = ty; a.tw = tw; a.th = th; a.tcls = tcls; a.tconf = tconf;
  a.class_mask = class_mask; a.iou_scores = iou_scores;
  a.partials = reinterpret_cast<double*>(workspace);
  // the ticket / flag words live behind the partials; they are zero between calls (the last block resets them), the very
  // first use of a workspace must hand in zeroed memory
  a.ticket = reinterpret_cast<unsigned int*>(a.partials + (size_t)YL_BLOCKS * YL_SUMS);
  a.bad = reinterpret_cast<int*>(a.ticket + 1);
  a.result = result;
  // stream-ordered reset of the two words in front of every call: a launch that ever aborted half-way (some blocks had drawn
  // their tickets) must not leave a counter behind that no later call can bring back to "last block" - the result would
  // never be written again
  ME_HIP(hipMemsetAsync(a.ticket, 0, 2 * sizeof(unsigned int), stream));
  const long long cells = (long long)n * num_anchors * g * g;
  long long ib = (cells * num_classes + 255) / 256;
  if (ib > 4096) ib = 4096;
  hipLaunchKernelGGL(yolo_targets_init_kernel, dim3((unsigned)ib), dim3(256), 0, stream, a);
  if (m > 0) hipLaunchKernelGGL(yolo_targets_scatter_kernel, dim3((m + 255) / 256), dim3(256), 0, stream, a);
  long long rb = (cells + 255) / 256;
  if (rb > YL_BLOCKS) rb = YL_BLOCKS;
  hipLaunchKernelGGL(yolo_loss_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, stream, a);
  return me::check_launch("yolo_loss_fwd");
}

int me_yolo_loss_fwd_f32(const float* raw, int64_t pitch, int32_t n, int32_t g, int32_t num_anchors, int32_t num_classes,
                         const float* scaled_anchors_host, const float* targets, int32_t m, float ignore_thres, float obj_scale,
                         float noobj_scale, uint8_t* obj_mask, uint8_t* noobj_mask, float* tx, float* ty, float* tw, float* th,
                         float* tcls, float* tconf, float* class_mask, float* iou_scores, void* workspace, float* result,
                         void* stream) {
  return yolo_loss_fwd(raw, pitch, n, g, num_anchors, num_classes, scaled_anchors_host, targets, m, nullptr, ignore_thres, obj_scale,
                       noobj_scale, obj_mask, noobj_mask, tx, ty, tw, th, tcls, tconf, class_mask, iou_scores, workspace, result,
                       stream);
}

// The same launches over a fixed-capacity target table: `capacity` rows are addressable, the device word *m_device says how many
// of them are targets.  Nothing about the call depends on the step's target count, so it can sit in a captured hipGraph
// (millieye_amd/detector_graph.py) that is replayed with another table every step.
int me_yolo_loss_fwd_counted_f32(const float* raw, int64_t pitch, int32_t n, int32_t g, int32_t num_anchors, int32_t num_classes,
                                 const float* scaled_anchors_host, const float* targets, int32_t capacity,
                                 const int32_t* m_device, float ignore_thres, float obj_scale, float noobj_scale, uint8_t* obj_mask,
                                 uint8_t* noobj_mask, float* tx, float* ty, float* tw, float* th, float* tcls, float* tconf,
                                 float* class_mask, float* iou_scores, void* workspace, float* result, void* stream) {
  ME_REQUIRE(m_device != nullptr && capacity > 0, ME_E_NULLPTR, "me_yolo_loss_fwd_counted_f32: needs a device row count and a capacity");
  return yolo_loss_fwd(raw, pitch, n, g, num_anchors, num_classes, scaled_anchors_host, targets, capacity, m_device, ignore_thres,
                       obj_scale, noobj_scale, obj_mask, noobj_mask, tx, ty, tw, th, tcls, tconf, class_mask, iou_scores, workspace,
                       result, stream);
}

}  // extern "C"
