/*
 * ORACLE (test infrastructure - never linked into the product).
 *
 * Plain-C restatement of the three torchvision 0.6.0 CPU operators the reference's hot path
 * calls (torchvision is an un-vendored dependency pinned in /root/reference/README.md:10 and
 * is NOT installed in this image):
 *
 *   roi_align      <- torchvision.ops.roi_align     module3_our_dataset/my_models.py:496
 *   ps_roi_align   <- torchvision.ops.ps_roi_align  module3_our_dataset/my_models.py:495
 *   nms / batched_nms <- torchvision.ops.boxes      module3_our_dataset/utils/utils.py:372
 *
 * PARITY UNPINNED: the reference holds no test or golden vector at this boundary and the
 * dependency cannot be executed here, so these follow the library's published algorithm
 * (torchvision/csrc/cpu/{ROIAlign,PSROIAlign,nms}_cpu.cpp @ v0.6.0),
 * float (T = float) arithmetic, same loop and summation order.  Tie order of equal scores
 * in nms (std::sort is unstable upstream) is fixed to "lower index first".
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: no FMA contraction, so results are
 * reproducible and the HIP kernels - compiled with contraction off in the index-critical
 * parts - can be compared bit for bit).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* bilinear sample with torchvision's border rules; returns value, optionally the 4 weights/positions */
static float bilinear(const float* img, int height, int width, float y, float x) {
  if (y < -1.0f || y > (float)height || x < -1.0f || x > (float)width) return 0.0f;
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; } else { y_high = y_low + 1; }
  if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; } else { x_high = x_low + 1; }
  float ly = y - y_low, lx = x - x_low;
  float hy = 1.f - ly, hx = 1.f - lx;
  float v1 = img[y_low * width + x_low], v2 = img[y_low * width + x_high];
  float v3 = img[y_high * width + x_low], v4 = img[y_high * width + x_high];
  float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
  return (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
}

static void bilinear_grad(int height, int width, float y, float x, float* w1, float* w2, float* w3, float* w4,
                          int* x_low, int* x_high, int* y_low, int* y_high) {
  if (y < -1.0f || y > (float)height || x < -1.0f || x > (float)width) {
    *w1 = *w2 = *w3 = *w4 = 0.f;
    *x_low = *x_high = *y_low = *y_high = -1;
    return;
  }
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  *y_low = (int)y;
  *x_low = (int)x;
  if (*y_low >= height - 1) { *y_high = *y_low = height - 1; y = (float)*y_low; } else { *y_high = *y_low + 1; }
  if (*x_low >= width - 1) { *x_high = *x_low = width - 1; x = (float)*x_low; } else { *x_high = *x_low + 1; }
  float ly = y - *y_low, lx = x - *x_low;
  float hy = 1.f - ly, hx = 1.f - lx;
  *w1 = hy * hx; *w2 = hy * lx; *w3 = ly * hx; *w4 = ly * lx;
}

/* input [N,C,H,W], rois [K,5] (batch, x1,y1,x2,y2), output [K,C,P,P]; aligned=False, sampling_ratio=-1 */
void tv_roi_align_forward(const float* input, int channels, int height, int width, const float* rois, int k,
                          int pooled, float spatial_scale, int sampling_ratio, int aligned, float* output) {
  for (int n = 0; n < k; n++) {
    const float* r = rois + n * 5;
    int b = (int)r[0];
    float offset = aligned ? 0.5f : 0.0f;
    float sw = r[1] * spatial_scale - offset, sh = r[2] * spatial_scale - offset;
    float ew = r[3] * spatial_scale - offset, eh = r[4] * spatial_scale - offset;
    float roi_w = ew - sw, roi_h = eh - sh;
    if (!aligned) {
      roi_w = roi_w > 1.f ? roi_w : 1.f;
      roi_h = roi_h > 1.f ? roi_h : 1.f;
    }
    float bin_h = roi_h / (float)pooled, bin_w = roi_w / (float)pooled;
    int gh = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_h / pooled);
    int gw = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_w / pooled);
    int cnt = gh * gw;
    const float count = (float)(cnt > 1 ? cnt : 1);
    for (int c = 0; c < channels; c++) {
      const float* img = input + ((size_t)b * channels + c) * height * width;
      for (int ph = 0; ph < pooled; ph++)
        for (int pw = 0; pw < pooled; pw++) {
          float acc = 0.f;
          for (int iy = 0; iy < gh; iy++) {
            const float yy = sh + ph * bin_h + ((float)(iy + .5f)) * bin_h / (float)gh;
            for (int ix = 0; ix < gw; ix++) {
              const float xx = sw + pw * bin_w + ((float)(ix + .5f)) * bin_w / (float)gw;
              acc += bilinear(img, height, width, yy, xx);
            }
          }
          acc /= count;
          output[(((size_t)n * channels + c) * pooled + ph) * pooled + pw] = acc;
        }
    }
  }
}

void tv_roi_align_backward(const float* grad_out, int batch, int channels, int height, int width, const float* rois,
                           int k, int pooled, float spatial_scale, int sampling_ratio, int aligned,
                           float* grad_in /* [batch,C,H,W], zero-filled by caller */) {
  (void)batch;
  for (int n = 0; n < k; n++) {
    const float* r = rois + n * 5;
    int b = (int)r[0];
    float offset = aligned ? 0.5f : 0.0f;
    float sw = r[1] * spatial_scale - offset, sh = r[2] * spatial_scale - offset;
    float ew = r[3] * spatial_scale - offset, eh = r[4] * spatial_scale - offset;
    float roi_w = ew - sw, roi_h = eh - sh;
    if (!aligned) {
      roi_w = roi_w > 1.f ? roi_w : 1.f;
      roi_h = roi_h > 1.f ? roi_h : 1.f;
    }
    float bin_h = roi_h / (float)pooled, bin_w = roi_w / (float)pooled;
    int gh = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_h / pooled);
    int gw = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_w / pooled);
    const float count = (float)(gh * gw);
    for (int c = 0; c < channels; c++) {
      float* gimg = grad_in + ((size_t)b * channels + c) * height * width;
      for (int ph = 0; ph < pooled; ph++)
        for (int pw = 0; pw < pooled; pw++) {
          const float g = grad_out[(((size_t)n * channels + c) * pooled + ph) * pooled + pw];
          for (int iy = 0; iy < gh; iy++) {
            const float yy = sh + ph * bin_h + ((float)(iy + .5f)) * bin_h / (float)gh;
            for (int ix = 0; ix < gw; ix++) {
              const float xx = sw + pw * bin_w + ((float)(ix + .5f)) * bin_w / (float)gw;
              float w1, w2, w3, w4;
              int xl, xh, yl, yh;
              bilinear_grad(height, width, yy, xx, &w1, &w2, &w3, &w4, &xl, &xh, &yl, &yh);
              if (xl >= 0 && xh >= 0 && yl >= 0 && yh >= 0) {
                gimg[yl * width + xl] += g * w1 / count;
                gimg[yl * width + xh] += g * w2 / count;
                gimg[yh * width + xl] += g * w3 / count;
                gimg[yh * width + xh] += g * w4 / count;
              }
            }
          }
        }
    }
  }
}

/* input [N,C,H,W] with C = c_out*P*P; output [K,c_out,P,P]; channel_mapping optional */
void tv_ps_roi_align_forward(const float* input, int channels, int height, int width, const float* rois, int k,
                             int pooled, float spatial_scale, int sampling_ratio, float* output) {
  const int c_out_n = channels / (pooled * pooled);
  for (int n = 0; n < k; n++) {
    const float* r = rois + n * 5;
    int b = (int)r[0];
    float sw = r[1] * spatial_scale - 0.5f, sh = r[2] * spatial_scale - 0.5f;
    float ew = r[3] * spatial_scale - 0.5f, eh = r[4] * spatial_scale - 0.5f;
    float roi_w = ew - sw, roi_h = eh - sh;
    float bin_h = roi_h / (float)pooled, bin_w = roi_w / (float)pooled;
    int c_in = 0;
    for (int c_out = 0; c_out < c_out_n; ++c_out)
      for (int ph = 0; ph < pooled; ++ph)
        for (int pw = 0; pw < pooled; ++pw) {
          float hstart = (float)ph * bin_h + sh;
          float wstart = (float)pw * bin_w + sw;
          int gh = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_h / pooled);
          int gw = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_w / pooled);
          const float count = (float)(gh * gw);
          const float* img = input + ((size_t)b * channels + c_in) * height * width;
          float out_sum = 0.f;
          for (int iy = 0; iy < gh; iy++) {
            const float y = hstart + ((float)(iy + .5f)) * bin_h / (float)gh;
            for (int ix = 0; ix < gw; ix++) {
              const float x = wstart + ((float)(ix + .5f)) * bin_w / (float)gw;
              out_sum += bilinear(img, height, width, y, x);
            }
          }
          out_sum /= count;
          output[(((size_t)n * c_out_n + c_out) * pooled + ph) * pooled + pw] = out_sum;
          c_in++;
        }
  }
}

void tv_ps_roi_align_backward(const float* grad_out, int channels, int height, int width, const float* rois, int k,
                              int pooled, float spatial_scale, int sampling_ratio,
                              float* grad_in /* zero-filled */) {
  const int c_out_n = channels / (pooled * pooled);
  for (int n = 0; n < k; n++) {
    const float* r = rois + n * 5;
    int b = (int)r[0];
    float sw = r[1] * spatial_scale - 0.5f, sh = r[2] * spatial_scale - 0.5f;
    float ew = r[3] * spatial_scale - 0.5f, eh = r[4] * spatial_scale - 0.5f;
    float roi_w = ew - sw, roi_h = eh - sh;
    float bin_h = roi_h / (float)pooled, bin_w = roi_w / (float)pooled;
    int c_in = 0;
    for (int c_out = 0; c_out < c_out_n; ++c_out)
      for (int ph = 0; ph < pooled; ++ph)
        for (int pw = 0; pw < pooled; ++pw) {
          float hstart = (float)ph * bin_h + sh;
          float wstart = (float)pw * bin_w + sw;
          int gh = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_h / pooled);
          int gw = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_w / pooled);
          const float count = (float)(gh * gw);
          float* gimg = grad_in + ((size_t)b * channels + c_in) * height * width;
          const float g = grad_out[(((size_t)n * c_out_n + c_out) * pooled + ph) * pooled + pw];
          for (int iy = 0; iy < gh; iy++) {
            const float y = hstart + ((float)(iy + .5f)) * bin_h / (float)gh;
            for (int ix = 0; ix < gw; ix++) {
              const float x = wstart + ((float)(ix + .5f)) * bin_w / (float)gw;
              float w1, w2, w3, w4;
              int xl, xh, yl, yh;
              bilinear_grad(height, width, y, x, &w1, &w2, &w3, &w4, &xl, &xh, &yl, &yh);
              if (xl >= 0 && xh >= 0 && yl >= 0 && yh >= 0) {
                gimg[yl * width + xl] += g * w1 / count;
                gimg[yl * width + xh] += g * w2 / count;
                gimg[yh * width + xl] += g * w3 / count;
                gimg[yh * width + xh] += g * w4 / count;
              }
            }
          }
          c_in++;
        }
  }
}

/* ---- nms --------------------------------------------------------------------------------- */
typedef struct { float s; int i; } sidx;
static int cmp_desc(const void* a, const void* b) {
  const sidx* x = (const sidx*)a; const sidx* y = (const sidx*)b;
  if (x->s > y->s) return -1;
  if (x->s < y->s) return 1;
  return (x->i > y->i) - (x->i < y->i); /* tie: lower index first (upstream: unspecified) */
}

/* dets [m,4] xyxy; returns number kept, indices (descending score) into keep[m] */
int tv_nms(const float* dets, const float* scores, int m, float thr, int64_t* keep) {
  if (m == 0) return 0;
  sidx* order = (sidx*)malloc(sizeof(sidx) * m);
  float* areas = (float*)malloc(sizeof(float) * m);
  unsigned char* sup = (unsigned char*)calloc(m, 1);
  for (int i = 0; i < m; i++) {
    order[i].s = scores[i]; order[i].i = i;
    areas[i] = (dets[4 * i + 2] - dets[4 * i]) * (dets[4 * i + 3] - dets[4 * i + 1]);
  }
  qsort(order, m, sizeof(sidx), cmp_desc);
  int nk = 0;
  for (int _i = 0; _i < m; _i++) {
    int i = order[_i].i;
    if (sup[i]) continue;
    keep[nk++] = i;
    float ix1 = dets[4 * i], iy1 = dets[4 * i + 1], ix2 = dets[4 * i + 2], iy2 = dets[4 * i + 3];
    float iarea = areas[i];
    for (int _j = _i + 1; _j < m; _j++) {
      int j = order[_j].i;
      if (sup[j]) continue;
      float xx1 = (ix1 < dets[4 * j]) ? dets[4 * j] : ix1;           /* std::max(ix1, x1[j]) */
      float yy1 = (iy1 < dets[4 * j + 1]) ? dets[4 * j + 1] : iy1;
      float xx2 = (dets[4 * j + 2] < ix2) ? dets[4 * j + 2] : ix2;   /* std::min(ix2, x2[j]) */
      float yy2 = (dets[4 * j + 3] < iy2) ? dets[4 * j + 3] : iy2;
      float dw = xx2 - xx1, dh = yy2 - yy1;
      float w = (0.f < dw) ? dw : 0.f;                               /* std::max(0, xx2-xx1) */
      float h = (0.f < dh) ? dh : 0.f;
      float inter = w * h;
      float ovr = inter / (iarea + areas[j] - inter);
      if (ovr > thr) sup[j] = 1;
    }
  }
  free(order); free(areas); free(sup);
  return nk;
}

/* boxes.max() with torch semantics (NaN propagates) */
static float max_all(const float* v, int n) {
  float m = -INFINITY;
  for (int i = 0; i < n; i++) {
    if (v[i] != v[i]) return NAN;
    if (v[i] > m) m = v[i];
  }
  return m;
}

int tv_batched_nms(const float* boxes, const float* scores, const float* idxs, int m, float thr, int64_t* keep) {
  if (m == 0) return 0;
  float maxc = max_all(boxes, 4 * m);
  float* shifted = (float*)malloc(sizeof(float) * 4 * m);
  for (int i = 0; i < m; i++) {
    float off = idxs[i] * (maxc + 1.f);
    for (int c = 0; c < 4; c++) shifted[4 * i + c] = boxes[4 * i + c] + off;
  }
  int nk = tv_nms(shifted, scores, m, thr, keep);
  free(shifted);
  return nk;
}
