// nms.hip - confidence filter + class-aware greedy NMS on the GPU (gfx950).
//
// Replaces non_max_suppression_cpp (module3_our_dataset/utils/utils.py:337-378) and the
// torchvision batched_nms / nms it calls (semantics: SURVEY.md Appendix C, restated bit for
// bit in oracle/tv_ops.c).  Integer / index work is bit-exact with that oracle: the IoU
// arithmetic below uses the same operation order with FP contraction off, comparisons are
// written as the C++ std::max / std::min the CPU kernel uses (NaN behaviour included).
//
// Round 3: up to MATN (4096) candidates per image the selection is no longer one workgroup's serial argmax loop but the
// whole chip's work (BASELINE north_star: "wavefront ballot / reduce for NMS"):
//   nms_rank    sorts an image's candidates by key (rank = number of larger keys, 256 candidates per workgroup) and writes
//               the sorted offset boxes (boxes + label * (max + 1); the maximum comes from nms_prep's atomics);
//   nms_matrix  persistent grid over every (image, 64 x 64 block) of the upper-triangular suppression bit matrix: a wave
//               per block, lane = row, 64 IoU tests per lane against the column boxes (readlane broadcast), one 64-bit
//               word per lane;
//   nms_scan    one workgroup per image walks the sorted list 64 candidates at a time: the block's 64 matrix rows are
//               staged in LDS (the next block's rows are already in flight in registers), wave 0 resolves the block with
//               scalar bit operations on the diagonal word (find-first-set over the not-yet-removed candidates, OR the
//               winner's word in) and ORs the winners' rows into the removed mask (lane = 64-candidate word); stops at
//               max_det winners.  Same visiting order and the same IoU test as before: bit-identical results.
// More candidates than MATN (conf_thresh near 0): the single-workgroup kernel below, unchanged (MILLIEYE_NMS_LEGACY=1 forces it).
//
// Pipeline per call (launches + one 12*n byte memset):
//   nms_prep    one thread per prediction row: conf >= thr filter, xywh->xyxy, class
//               max/argmax, append candidate {raw box, key, label, cls_conf} (wave-aggregated
//               atomic append; order does not matter, the key carries the row index).
//   nms_select  one 1024-thread workgroup per image: max-coordinate reduction (the batched_nms
//               offset trick), then greedy selection: repeatedly take the alive candidate with
//               the highest (score, lowest row) key - a 64-lane shuffle + LDS reduction - and
//               suppress every alive candidate whose IoU with it exceeds the threshold.  Each
//               thread owns <= 32 candidates and keeps their alive bits in one register.
//               Stops after max_det winners (the reference keeps keep[:200]).
//   nms_emit    gathers the kept rows into det[n, max_det, 7+C].
#include <math.h>
#include "common.h"

namespace {

constexpr int SEL_THREADS = 1024;
constexpr int MAX_ROWS = 32768;  // 32 candidates per thread
constexpr int LDS_CANDS = 4096;  // candidates per image kept in LDS by nms_select (96 KiB)
constexpr int MAT_CANDS = 768;   // ... up to this many are resolved through a suppression bit matrix in LDS (108 KiB)

constexpr int MATN = 1024;       // sorted candidates per image the matrix path looks at: the greedy walk stops at max_det winners,
                                 // and 200 winners almost always sit among the 1024 best-scored candidates; an image that needs
                                 // more (nms_scan sets its fallback flag) is redone by the single-workgroup kernel

struct NmsWs {
  float4* raw;               // [n][cap] raw xyxy
  float4* off;               // [n][cap] boxes + label*(max+1) (matrix path: in sorted order)
  unsigned long long* key;   // [n][cap] (sortable(score) << 32) | ~row
  float* label;              // [n][cap]
  float* clsconf;            // [n][cap]
  int* keep_slot;            // [n][cap_keep] candidate slot of every winner
  int* sslot;                // [n][cap] matrix path: candidate slot of sorted position
  unsigned long long* mat;   // [n][matn][matn / 64] suppression bits of the sorted candidates (upper triangle)
  int* cand_count;           // [n]   } one memset
  unsigned* maxbits;         // [n]   } sortable() bits of the largest coordinate among the candidates
  int* nanflag;              // [n]   } a candidate coordinate is NaN
  int* fallback;             // [n]   } nms_scan: the capped matrix did not reach max_det winners - nms_select redoes the image
  int cap, matn, wc;         // wc = matn / 64 words per matrix row
  int legacy;                // force the single-workgroup kernel
};

__host__ __device__ inline long long align_up(long long v, long long a) { return (v + a - 1) / a * a; }

inline int matn_of(int rows) {
  const int r = (rows + 63) & ~63;
  return r < MATN ? r : MATN;
}

inline long long ws_bytes(int n, int rows) {
  const long long cap = rows;
  const long long matn = matn_of(rows);
  long long b = 0;
  b += align_up((long long)n * cap * 16, 256) * 2;  // raw, off
  b += align_up((long long)n * cap * 8, 256);       // key
  b += align_up((long long)n * cap * 4, 256) * 4;   // label, clsconf, keep_slot, sslot
  b += align_up((long long)n * matn * (matn / 64) * 8, 256);  // mat
  b += align_up((long long)n * 16, 256);            // cand_count, maxbits, nanflag, fallback
  return b;
}

inline NmsWs carve(void* base, int n, int rows) {
  NmsWs w;
  char* p = reinterpret_cast<char*>(base);
  const long long cap = rows;
  w.cap = rows;
  w.raw = reinterpret_cast<float4*>(p); p += align_up((long long)n * cap * 16, 256);
  w.off = reinterpret_cast<float4*>(p); p += align_up((long long)n * cap * 16, 256);
  w.key = reinterpret_cast<unsigned long long*>(p); p += align_up((long long)n * cap * 8, 256);
  w.label = reinterpret_cast<float*>(p); p += align_up((long long)n * cap * 4, 256);
  w.clsconf = reinterpret_cast<float*>(p); p += align_up((long long)n * cap * 4, 256);
  w.keep_slot = reinterpret_cast<int*>(p); p += align_up((long long)n * cap * 4, 256);
  w.sslot = reinterpret_cast<int*>(p); p += align_up((long long)n * cap * 4, 256);
  w.matn = matn_of(rows);
  w.wc = w.matn / 64;
  w.mat = reinterpret_cast<unsigned long long*>(p); p += align_up((long long)n * w.matn * w.wc * 8, 256);
  w.cand_count = reinterpret_cast<int*>(p);
  w.maxbits = reinterpret_cast<unsigned*>(p) + n;
  w.nanflag = reinterpret_cast<int*>(p) + 2 * n;
  w.fallback = reinterpret_cast<int*>(p) + 3 * n;
  static const int legacy = [] {
    const char* e = getenv("MILLIEYE_NMS_LEGACY");
    return (e && e[0] == '1') ? 1 : 0;
  }();
  w.legacy = legacy;
  return w;
}

__device__ __forceinline__ unsigned sortable(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ unsigned long long make_key(float score, int row) {
  return ((unsigned long long)sortable(score) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)row);
}

// boxes.max() of batched_nms's offset trick, gathered while the candidates are appended: the largest non-NaN coordinate as
// sortable bits (atomicMax; 0 = "none yet" sorts below every float) and a flag for NaN coordinates (torch's max is NaN then)
__device__ __forceinline__ void note_max_coord(const NmsWs& w, int img, float x1, float y1, float x2, float y2) {
  if ((x1 != x1) | (y1 != y1) | (x2 != x2) | (y2 != y2)) atomicOr(&w.nanflag[img], 1);
  const float m = fmaxf(fmaxf(x1, y1), fmaxf(x2, y2));
  if (m == m) atomicMax(&w.maxbits[img], sortable(m));
}

__device__ __forceinline__ float unsortable(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

// ---- prep -------------------------------------------------------------------------------------
// class max / argmax of one row by a group of 16 lanes: torch.max(1) semantics = the sequential scan "replace when v > best,
// or when v is NaN and best is not" - i.e. the order: any NaN beats every number (first NaN wins), otherwise the larger
// value, ties to the lower index - which is associative, so a shuffle tree gives the same answer.
__device__ __forceinline__ bool cls_beats(float v, int i, float bv, int bi) {
  const bool vn = v != v, bn = bv != bv;
  if (vn != bn) return vn;
  if (vn) return i < bi;
  return v > bv || (v == bv && i < bi);
}

__global__ __launch_bounds__(256) void nms_prep_kernel(float* pred, int rows, int num_classes, float conf_thresh,
                                                       int writeback, NmsWs w) {
#pragma clang fp contract(off)
  __shared__ int s_rows[256];
  __shared__ int s_n, s_base, s_nan;
  __shared__ unsigned s_max;
  const int img = blockIdx.y;
  const int row = blockIdx.x * 256 + threadIdx.x;
  const int per = 5 + num_classes;
  if (threadIdx.x == 0) {
    s_n = 0;
    s_nan = 0;
    s_max = 0u;
  }
  __syncthreads();
  if (row < rows) {
    float* p = pred + ((long long)img * rows + row) * per;
    const float conf = p[4];
    if (writeback) {  // xywh2xyxy (utils.py:68-74) in place on every row: half extents as w / 2
      const float cx = p[0], cy = p[1], bw = p[2], bh = p[3];
      p[0] = cx - bw / 2; p[1] = cy - bh / 2; p[2] = cx + bw / 2; p[3] = cy + bh / 2;
    }
    if (conf >= conf_thresh) s_rows[atomicAdd(&s_n, 1)] = row;  // order is irrelevant: the key carries the row index
  }
  __syncthreads();
  // the (few) rows that pass: 16 lanes per row read its class scores in 64-byte pieces - the one-thread-per-row loop of
  // rounds 1-2 walked 80 dependent 4-byte loads per passing row while the other lanes of its wave waited (78 us at batch 32)
  const int npass = s_n;
  if (npass == 0) return;
  // ONE global atomic per block for the slots, one for the maximum: thousands of same-address atomics per image serialise in
  // the L2 (2500 candidates per image: 59 us with one atomicAdd per candidate, 78 us with the atomicMax beside it)
  if (threadIdx.x == 0) s_base = atomicAdd(&w.cand_count[img], npass);
  __syncthreads();
  const int grp = threadIdx.x >> 4, gl = threadIdx.x & 15;
  for (int q = grp; q < npass; q += 16) {
    const int r = s_rows[q];
    const float* p = pred + ((long long)img * rows + r) * per;
    float best = -INFINITY;
    int arg = 0x7fffffff;
    for (int c = gl; c < num_classes; c += 16) {
      const float v = p[5 + c];
      if (arg == 0x7fffffff || cls_beats(v, c, best, arg)) {
        best = v;
        arg = c;
      }
    }
#pragma unroll
    for (int sft = 8; sft >= 1; sft >>= 1) {
      const float ov = __shfl_xor(best, sft, 16);
      const int oi = __shfl_xor(arg, sft, 16);
      if (oi != 0x7fffffff && (arg == 0x7fffffff || cls_beats(ov, oi, best, arg))) {
        best = ov;
        arg = oi;
      }
    }
    if (gl == 0) {
      if (num_classes <= 0) {
        best = -INFINITY;
        arg = 0;
      }
      float x1, y1, x2, y2;
      if (writeback) {  // already converted above (same thread block, behind the barrier)
        x1 = p[0]; y1 = p[1]; x2 = p[2]; y2 = p[3];
      } else {
        const float cx = p[0], cy = p[1], bw = p[2], bh = p[3];
        x1 = cx - bw / 2; y1 = cy - bh / 2; x2 = cx + bw / 2; y2 = cy + bh / 2;
      }
      const long long o = (long long)img * w.cap + s_base + q;
      w.raw[o] = make_float4(x1, y1, x2, y2);
      w.key[o] = make_key(p[4], r);
      w.label[o] = (float)arg;
      w.clsconf[o] = best;
      if ((x1 != x1) | (y1 != y1) | (x2 != x2) | (y2 != y2)) atomicOr(&s_nan, 1);
      const float m = fmaxf(fmaxf(x1, y1), fmaxf(x2, y2));
      if (m == m) atomicMax(&s_max, sortable(m));
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_nan) atomicOr(&w.nanflag[img], 1);
    if (s_max) atomicMax(&w.maxbits[img], s_max);
  }
}


// ---- YOLO decode with the candidate lists as a by-product ------------------------------------------------------------------
// me_yolo_decode_cand_f32: the decode of one [yolo] scale (same arithmetic as csrc/elementwise.hip:yolo_decode_kernel, reference
// yolov3/models.py:132-179) that ALSO does nms_prep's job for its rows while they are in registers: confidence filter, class
// max / argmax, xywh -> xyxy, candidate append, maximum coordinate.  nms_prep re-read the 116 MB prediction tensor right behind the
// decode that wrote it (78 - 100 us at batch 32, the largest part of the NMS stage after round 3's select rewrite).
// One wave per row (5 + C <= 128 elements: lane k and k + 64), 64 rows per workgroup, one global atomic per workgroup.
struct YoloCand {
  me_yolo_desc y[3];  // the [yolo] scales of one launch (count of them used)
  int wg_end[3];      // exclusive prefix sums of the workgroups per scale
  int count;
  float conf_thresh;
};

// RPW rows per wave (4 * RPW per workgroup).  Every row's raw values are fetched before the first one is decoded: with the loads
// inside the row loop a wave paid one load latency plus the previous row's store acknowledgement per row (stores count in vmcnt
// on gfx9) - 1.3 - 2.7 us per row, 21 - 43 us per scale at ANY batch size.  Small batches take RPW = 1 / 4 so that the rows
// spread over the whole chip.
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v);  // (DPP network, defined with the select kernel)

__global__ __launch_bounds__(256) void zero_ints_kernel(int* p, int count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < count) p[i] = 0;
}

template <int RPW>
__global__ __launch_bounds__(256) void yolo_decode_cand_kernel(YoloCand d, NmsWs w) {
#pragma clang fp contract(off)
  __shared__ float4 s_box[64];
  __shared__ unsigned long long s_key[64];
  __shared__ float s_lab[64], s_cls[64];
  __shared__ int s_n, s_base, s_nan;
  __shared__ unsigned s_max;
  int sc = 0, bx = blockIdx.x;
  if (d.count > 1 && bx >= d.wg_end[0]) sc = (d.count > 2 && bx >= d.wg_end[1]) ? 2 : 1;
  if (sc) bx -= d.wg_end[sc - 1];
  const me_yolo_desc& y = d.y[sc];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int img = blockIdx.y;
  const int per = y.num_classes + 5;
  const int gg = y.g * y.g;
  const int rows_scale = y.num_anchors * gg;
  if (threadIdx.x == 0) {
    s_n = 0;
    s_nan = 0;
    s_max = 0u;
  }
  __syncthreads();
  const int rr0 = __builtin_amdgcn_readfirstlane(bx * (4 * RPW) + wv * RPW);  // (row arithmetic on the scalar unit)
  float t[RPW][2];
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int rr = rr0 + i;
    const int a = rr / gg, pix = rr - a * gg;
    const float* xin = y.x + ((long long)img * gg + pix) * y.x_pitch + a * per;
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) t[i][hlf] = (rr < rows_scale && lane + 64 * hlf < per) ? xin[lane + 64 * hlf] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int rr = rr0 + i;  // row of this scale
    if (rr >= rows_scale) break;  // (uniform per wave)
    const int a = rr / gg, pix = rr - a * gg;
    const int row = y.row_offset + rr;  // row of the prediction tensor
    float* out = y.out + ((long long)img * y.rows_total + row) * per;
    const float aw = y.anchors[2 * a], ah = y.anchors[2 * a + 1];
    const float gx = (float)(pix % y.g), gy = (float)(pix / y.g);
    float v[2] = {0.f, 0.f};
    // the kernel is VALU-bound (one wave per row of 85 values): ONE expf per element - of t for the w / h lanes, of -t for the
    // sigmoid lanes - instead of the three divergent paths; every element still sees the reference's operations in their order
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
      const int k = lane + 64 * hlf;
      if (k < per) {
        const float tt = t[i][hlf];
        const bool wh = hlf == 0 && (k == 2 || k == 3);
        const float e = expf(wh ? tt : -tt);
        float o;
        if (wh) {
          o = (e * (k == 2 ? aw : ah)) * y.stride;  // exp(t) * (anchor / stride), then * stride (yolov3/models.py:126,162-163,168)
        } else {
          const float sg = 1.f / (1.f + e);
          o = (hlf == 0 && k < 2) ? (sg + (k == 0 ? gx : gy)) * y.stride : sg;
        }
        out[k] = o;
        v[hlf] = o;
      }
    }
    const float conf = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v[0]), 4));
    if (!(conf >= d.conf_thresh)) continue;  // uniform
    // class max / argmax with torch.max(1) semantics (see cls_beats: any NaN beats every number, first NaN wins; else the larger
    // value, ties to the lower index) as ONE 64-bit wave maximum over the DPP network: high word = order-preserving bits of the
    // score (every NaN -> 0xFFFFFFFF; -0 -> +0, which compare equal), low word = ~index; lanes without a class hold 0
    auto class_key = [](float sc, int idx) {
      const unsigned hi = (sc != sc) ? 0xFFFFFFFFu : sortable(sc + 0.f);
      return ((unsigned long long)hi << 32) | (0xFFFFFFFFu - (unsigned)idx);
    };
    unsigned long long ck = 0ull;
    if (lane >= 5 && lane < per) ck = class_key(v[0], lane - 5);
    if (lane + 64 < per) {
      const unsigned long long c1 = class_key(v[1], lane + 59);
      ck = c1 > ck ? c1 : ck;
    }
    ck = wave_max_u64(ck);
    float best = -INFINITY;
    int arg = 0;
    if (y.num_classes > 0) {
      arg = (int)(0xFFFFFFFFu - (unsigned)(ck & 0xFFFFFFFFull));
      const int src = arg + 5;  // element index of the winner: lane src & 63 of half src >> 6
      const float b0 = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v[0]), src & 63));
      const float b1 = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v[1]), src & 63));
      best = src < 64 ? b0 : b1;
    }
    const float cx = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v[0]), 0));
    const float cy = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v[0]), 1));
    const float bw = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v[0]), 2));
    const float bh = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v[0]), 3));
    if (lane == 0) {
      const float x1 = cx - bw / 2, y1 = cy - bh / 2, x2 = cx + bw / 2, y2 = cy + bh / 2;  // xywh2xyxy (utils.py:68-74)
      const int q = atomicAdd(&s_n, 1);
      s_box[q] = make_float4(x1, y1, x2, y2);
      s_key[q] = make_key(conf, row);
      s_lab[q] = (float)arg;
      s_cls[q] = best;
      if ((x1 != x1) | (y1 != y1) | (x2 != x2) | (y2 != y2)) atomicOr(&s_nan, 1);
      const float m = fmaxf(fmaxf(x1, y1), fmaxf(x2, y2));
      if (m == m) atomicMax(&s_max, sortable(m));
    }
  }
  __syncthreads();
  const int npass = s_n;
  if (npass == 0) return;
  if (threadIdx.x == 0) {
    s_base = atomicAdd(&w.cand_count[img], npass);
    if (s_nan) atomicOr(&w.nanflag[img], 1);
    if (s_max) atomicMax(&w.maxbits[img], s_max);
  }
  __syncthreads();
  if ((int)threadIdx.x < npass) {
    const long long o = (long long)img * w.cap + s_base + threadIdx.x;
    w.raw[o] = s_box[threadIdx.x];
    w.key[o] = s_key[threadIdx.x];
    w.label[o] = s_lab[threadIdx.x];
    w.clsconf[o] = s_cls[threadIdx.x];
  }
}

// explicit boxes (box_ops.nms / batched_nms): every box is a candidate, img = 0
__global__ __launch_bounds__(256) void nms_prep_boxes_kernel(const float* boxes, const float* scores,
                                                             const float* labels, int m, NmsWs w) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= m) return;
  w.raw[row] = make_float4(boxes[4 * row], boxes[4 * row + 1], boxes[4 * row + 2], boxes[4 * row + 3]);
  w.key[row] = make_key(scores[row], row);
  w.label[row] = labels ? labels[row] : 0.f;
  w.clsconf[row] = 0.f;
  if (row == 0) w.cand_count[0] = m;
  note_max_coord(w, 0, boxes[4 * row], boxes[4 * row + 1], boxes[4 * row + 2], boxes[4 * row + 3]);
}

// ---- select -----------------------------------------------------------------------------------
// 64-lane unsigned max through the DPP network (row shifts + row broadcasts, gfx9 family): ~12 VALU ops instead of six
// ds_bpermute round trips (~120 cycles each) per 32-bit word - the argmax is on the critical path of every greedy step.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#define ME_DPP_MAX(ctrl, row_mask)                                                                      \
  {                                                                                                     \
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, row_mask, 0xf, false);    \
    v = o > v ? o : v;                                                                                  \
  }
  ME_DPP_MAX(0x111, 0xf)  // row_shr:1
  ME_DPP_MAX(0x112, 0xf)  // row_shr:2
  ME_DPP_MAX(0x114, 0xf)  // row_shr:4
  ME_DPP_MAX(0x118, 0xf)  // row_shr:8   -> lane 15 of every 16-lane row holds the row maximum
  ME_DPP_MAX(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
  ME_DPP_MAX(0x143, 0xc)  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave maximum
#undef ME_DPP_MAX
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// keys are (sortable score << 32) | ~row: maximum of the high words first, then of the low words among its holders
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
  const unsigned hi = (unsigned)(v >> 32);
  const unsigned mh = wave_max_u32(hi);
  const unsigned lo = (hi == mh) ? (unsigned)(v & 0xFFFFFFFFull) : 0u;
  const unsigned ml = wave_max_u32(lo);
  return ((unsigned long long)mh << 32) | ml;
}

// torchvision nms_cpu_kernel IoU test, literal operation order (std::max(a,b) = a < b ? b : a).
__device__ __forceinline__ bool iou_exceeds(const float4 bi, float iarea, const float4 bj, float jarea, float thr) {
#pragma clang fp contract(off)
  const float xx1 = (bi.x < bj.x) ? bj.x : bi.x;
  const float yy1 = (bi.y < bj.y) ? bj.y : bi.y;
  const float xx2 = (bj.z < bi.z) ? bj.z : bi.z;
  const float yy2 = (bj.w < bi.w) ? bj.w : bi.w;
  const float dw = xx2 - xx1, dh = yy2 - yy1;
  const float ww = (0.f < dw) ? dw : 0.f;
  const float hh = (0.f < dh) ? dh : 0.f;
  const float inter = ww * hh;
  const float ovr = inter / (iarea + jarea - inter);
  return ovr > thr;
}

__device__ __forceinline__ float box_area(const float4 b) {
#pragma clang fp contract(off)
  return (b.z - b.x) * (b.w - b.y);
}

__global__ __launch_bounds__(SEL_THREADS) void nms_select_kernel(NmsWs w, int use_offsets, float iou_thresh,
                                                                 int max_det, int* out_count) {
#pragma clang fp contract(off)
  __shared__ unsigned long long s_red[SEL_THREADS / 64];
  __shared__ float s_redf[SEL_THREADS / 64];
  __shared__ int s_redi[SEL_THREADS / 64];
  __shared__ float4 s_wbox;
  __shared__ float s_warea;
  __shared__ float s_maxc;

  const int img = blockIdx.x;
  const int t = threadIdx.x;
  const int lane = t & 63, wv = t >> 6;
  const int cnt = w.cand_count[img];
  if (cnt == 0) {
    if (t == 0) out_count[img] = 0;
    return;
  }
  if (!w.legacy && !w.fallback[img]) return;  // the matrix path (nms_rank / nms_matrix / nms_scan) has done this image
  const long long base = (long long)img * w.cap;
  // Only as many waves as the candidate count deserves take part (surplus waves exit before the first barrier;
  // s_barrier only counts live waves).  Up to 2048 candidates every thread owns <= 2 of them and the greedy loop
  // runs in "register mode" (below); beyond that the generic loop with up to 32 candidates per thread is used.
  const bool mat_mode = cnt <= MAT_CANDS;
  const bool reg_mode = !mat_mode && cnt <= 2 * SEL_THREADS;
  int T = SEL_THREADS;
  if (mat_mode) {
    T = cnt > 128 ? SEL_THREADS : ((cnt + 63) & ~63);  // the N^2 / 2 IoU tests of the bit matrix want every lane
  } else if (reg_mode) {
    T = ((cnt + 1) / 2 + 63) & ~63;
    if (T < 64) T = 64;
  }
  if (t >= T) return;
  const int nwaves = T >> 6;
  // up to LDS_CANDS candidates live in LDS (offset box + key, 24 B each): every greedy step re-reads the alive ones,
  // and from global memory that is two dependent ~1 us round trips per step - 200 steps = the whole kernel time
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  float4* s_box = reinterpret_cast<float4*>(dyn_lds);
  unsigned long long* s_key = reinterpret_cast<unsigned long long*>(dyn_lds + (size_t)LDS_CANDS * 16);
  const bool in_lds = cnt <= LDS_CANDS;

  // ---- offsets: boxes + label * (boxes.max() + 1)  (batched_nms) --------------------------
  float maxc = 0.f;
  if (use_offsets) {
    float m = -INFINITY;
    int nan = 0;
    for (int j = t; j < cnt; j += T) {
      const float4 b = w.raw[base + j];
      nan |= (b.x != b.x) | (b.y != b.y) | (b.z != b.z) | (b.w != b.w);
      m = fmaxf(m, fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w)));
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
      m = fmaxf(m, __shfl_xor(m, s, 64));
      nan |= __shfl_xor(nan, s, 64);
    }
    if (lane == 0) {
      s_redf[wv] = m;
      s_redi[wv] = nan;
    }
    __syncthreads();
    if (t == 0) {
      float mm = s_redf[0];
      int nn = s_redi[0];
      for (int k = 1; k < nwaves; ++k) {
        mm = fmaxf(mm, s_redf[k]);
        nn |= s_redi[k];
      }
      s_maxc = nn ? NAN : mm;
    }
    __syncthreads();
    maxc = s_maxc;
  }
  if (mat_mode) {
    // ---- matrix mode (the common case: a few hundred candidates).  Sort by key (rank = number of larger keys),
    // build the upper-triangular suppression bit matrix of the sorted boxes in parallel, then one wave walks it:
    // the next winner is the first clear bit of the "removed" words, and taking it ORs its matrix row in.  The greedy
    // chain of <= max_det dependent steps costs ~30 scalar-ish instructions each instead of a workgroup-wide argmax.
    unsigned long long* mkey = reinterpret_cast<unsigned long long*>(dyn_lds);                       // [MAT]
    float4* mbox = reinterpret_cast<float4*>(dyn_lds + MAT_CANDS * 8);                                // [MAT] unsorted
    float4* sbox = reinterpret_cast<float4*>(dyn_lds + MAT_CANDS * 24);                               // [MAT] sorted
    int* sslot = reinterpret_cast<int*>(dyn_lds + MAT_CANDS * 40);                                    // [MAT]
    int* skeep = reinterpret_cast<int*>(dyn_lds + MAT_CANDS * 44);                                    // [MAT]
    unsigned long long* mrow = reinterpret_cast<unsigned long long*>(dyn_lds + MAT_CANDS * 48);       // [MAT][W]
    const int W = (cnt + 63) >> 6;
    const float mp1 = maxc + 1.f;
    if (t < cnt) {
      float4 bb = w.raw[base + t];
      if (use_offsets) {
        const float o = w.label[base + t] * mp1;
        bb.x = bb.x + o; bb.y = bb.y + o; bb.z = bb.z + o; bb.w = bb.w + o;
      }
      mbox[t] = bb;
      mkey[t] = w.key[base + t];
    }
    __syncthreads();
    if (t < cnt) {
      const unsigned long long mine = mkey[t];
      int rank = 0;
      for (int i = 0; i < cnt; ++i) rank += (mkey[i] > mine) ? 1 : 0;  // keys are unique (they embed the row)
      sbox[rank] = mbox[t];
      sslot[rank] = t;
    }
    __syncthreads();
    for (int idx = t; idx < cnt * W; idx += T) {
      const int i = idx / W, wd = idx - i * W;
      unsigned long long bits = 0ull;
      if (wd >= (i >> 6)) {
        const float4 bi = sbox[i];
        const float ai = box_area(bi);
        const int j0 = wd << 6;
        for (int b = 0; b < 64; ++b) {
          const int j = j0 + b;
          if (j > i && j < cnt) {
            const float4 bj = sbox[j];
            if (iou_exceeds(bi, ai, bj, box_area(bj), iou_thresh)) bits |= 1ull << b;
          }
        }
      }
      mrow[(size_t)i * W + wd] = bits;
    }
    __syncthreads();
    if (wv != 0) return;
    // wave 0: lane l holds removed-word l (W <= 12)
    unsigned rem_lo = 0u, rem_hi = 0u;
    int kept = 0, i = 0;
    while (i < cnt && kept < max_det) {
      const int wd = i >> 6;
      const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)rem_lo, wd);
      const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)rem_hi, wd);
      unsigned long long avail = ~(((unsigned long long)hi << 32) | lo);
      avail &= ~0ull << (i & 63);
      const int last = cnt - (wd << 6);
      if (last < 64) avail &= (1ull << last) - 1ull;
      if (avail == 0ull) {
        i = (wd + 1) << 6;
        continue;
      }
      i = (wd << 6) + __builtin_ctzll(avail);
      if (lane == 0) skeep[kept] = sslot[i];
      ++kept;
      if (lane < W) {
        const unsigned long long row = mrow[(size_t)i * W + lane];
        rem_lo |= (unsigned)row;
        rem_hi |= (unsigned)(row >> 32);
      }
      ++i;
    }
    __syncthreads();  // (only wave 0 is alive) skeep written by lane 0 is read by every lane below
    for (int q = lane; q < kept; q += 64) w.keep_slot[base + q] = skeep[q];
    if (lane == 0) out_count[img] = kept;
    return;
  }
  if (reg_mode) {
    // ---- register mode: box / area / key of this thread's <= 2 candidates stay in VGPRs; one barrier per greedy
    // step: each wave publishes its best (key, box) into a parity-toggled LDS slot, after the barrier everybody
    // picks the workgroup winner from the <= 16 slots.  ~500 cycles per kept box instead of ~5000.
    __shared__ unsigned long long s_k2[2][SEL_THREADS / 64];
    __shared__ float4 s_b2[2][SEL_THREADS / 64];
    const float mp1 = maxc + 1.f;
    float4 cb[2];
    float ca[2];
    unsigned long long ck[2];
    int cj[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int j = t + c * T;
      cj[c] = j;
      ck[c] = 0ull;
      cb[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      ca[c] = 0.f;
      if (j < cnt) {
        float4 bb = w.raw[base + j];
        if (use_offsets) {
          const float o = w.label[base + j] * mp1;
          bb.x = bb.x + o; bb.y = bb.y + o; bb.z = bb.z + o; bb.w = bb.w + o;
        }
        cb[c] = bb;
        ca[c] = box_area(bb);
        ck[c] = w.key[base + j];  // never 0: the low word is 0xFFFFFFFF - row (make_key)
      }
    }
    int* s_keep = reinterpret_cast<int*>(dyn_lds);  // <= 2048 winners; register mode does not use s_box / s_key
    int kept = 0;
    int par = 0;
    float4 wbox = make_float4(0.f, 0.f, 0.f, 0.f);
    float warea = 0.f;
    bool have_winner = false;
    while (true) {
      unsigned long long best = 0ull;
      int best_c = 0;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (ck[c] == 0ull) continue;  // dead or absent
        if (have_winner && iou_exceeds(wbox, warea, cb[c], ca[c], iou_thresh)) {
          ck[c] = 0ull;
          continue;
        }
        if (ck[c] > best) {
          best = ck[c];
          best_c = c;
        }
      }
      const unsigned long long wg = wave_max_u64(best);
      if (best == wg && wg != 0ull) {  // unique owner inside the wave (keys embed the row)
        s_k2[par][wv] = wg;
        s_b2[par][wv] = cb[best_c];
      } else if (wg == 0ull && lane == 0) {
        s_k2[par][wv] = 0ull;
      }
      __syncthreads();
      unsigned long long g = s_k2[par][0];
      int gw = 0;
      for (int q = 1; q < nwaves; ++q) {
        const unsigned long long v = s_k2[par][q];
        if (v > g) {
          g = v;
          gw = q;
        }
      }
      if (g == 0ull) break;  // nothing alive (uniform)
      wbox = s_b2[par][gw];
      warea = box_area(wbox);
      have_winner = true;
      if (best == g) {  // the owner retires its candidate and records the winner
        s_keep[kept] = cj[best_c];  // LDS, flushed once at the end: a global store here would be waited for by
        ck[best_c] = 0ull;          // every following __syncthreads (vmcnt(0)) - 1 us per greedy step
      }
      ++kept;
      par ^= 1;
      if (kept >= max_det) break;
    }
    __syncthreads();
    for (int q = t; q < kept; q += T) w.keep_slot[base + q] = s_keep[q];
    if (t == 0) out_count[img] = kept;
    return;
  }
  unsigned alive = 0;
  {
    const float mp1 = maxc + 1.f;
    int i = 0;
    for (int j = t; j < cnt; j += T, ++i) {
      float4 b = w.raw[base + j];
      if (use_offsets) {
        const float o = w.label[base + j] * mp1;
        b.x = b.x + o; b.y = b.y + o; b.z = b.z + o; b.w = b.w + o;
      }
      if (in_lds) {
        s_box[j] = b;
        s_key[j] = w.key[base + j];
      } else {
        w.off[base + j] = b;
      }
      alive |= (1u << i);
    }
  }
  // each thread only ever re-reads the entries it wrote itself -> no barrier needed

  int kept = 0;
  bool have_winner = false;
  float4 wbox = make_float4(0.f, 0.f, 0.f, 0.f);
  float warea = 0.f;
  while (true) {
    unsigned long long best = 0ull;
    int best_i = -1;
    {
      unsigned bits = alive;
      while (bits) {
        const int i = __ffs(bits) - 1;
        bits &= bits - 1;
        const int j = t + i * T;
        if (have_winner) {
          const float4 b = in_lds ? s_box[j] : w.off[base + j];
          if (iou_exceeds(wbox, warea, b, box_area(b), iou_thresh)) {
            alive &= ~(1u << i);
            continue;
          }
        }
        const unsigned long long k = in_lds ? s_key[j] : w.key[base + j];
        if (k > best) {
          best = k;
          best_i = i;
        }
      }
    }
    unsigned long long g = wave_max_u64(best);
    if (lane == 0) s_red[wv] = g;
    __syncthreads();
    g = s_red[0];
    for (int k = 1; k < nwaves; ++k) g = s_red[k] > g ? s_red[k] : g;
    if (g == 0ull) break;  // nothing alive (uniform)
    if (best == g) {       // unique owner: keys embed the row index
      const int j = t + best_i * T;
      alive &= ~(1u << best_i);
      const float4 b = in_lds ? s_box[j] : w.off[base + j];
      s_wbox = b;
      s_warea = box_area(b);
      w.keep_slot[base + kept] = j;
    }
    __syncthreads();
    wbox = s_wbox;
    warea = s_warea;
    have_winner = true;
    ++kept;
    if (kept >= max_det) break;
  }
  if (t == 0) out_count[img] = kept;
}


// ---- matrix path -------------------------------------------------------------------------------
// rank: sorted position of every candidate = number of larger keys (keys are unique: they embed the row).  256 candidates
// per workgroup; the image's keys go through LDS in tiles of 2048, two keys per ds_read_b128.
constexpr int RANK_TILE = 2048;
__global__ __launch_bounds__(256) void nms_rank_kernel(NmsWs w, int use_offsets) {
#pragma clang fp contract(off)
  __shared__ __attribute__((aligned(16))) unsigned long long s_keys[RANK_TILE];
  const int img = blockIdx.y;
  const int cnt = w.cand_count[img];
  if (w.legacy || (int)blockIdx.x * 256 >= cnt) return;
  const long long base = (long long)img * w.cap;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const unsigned long long mine = j < cnt ? w.key[base + j] : ~0ull;
  int rank = 0;
  for (int t0 = 0; t0 < cnt; t0 += RANK_TILE) {
    const int len = cnt - t0 < RANK_TILE ? cnt - t0 : RANK_TILE;
    __syncthreads();
    for (int i = threadIdx.x; i < RANK_TILE; i += 256) s_keys[i] = i < len ? w.key[base + t0 + i] : 0ull;  // 0 < every key
    __syncthreads();
    const int len2 = (len + 1) & ~1;
    for (int i = 0; i < len2; i += 2) {
      const ulonglong2 k2 = *reinterpret_cast<const ulonglong2*>(&s_keys[i]);
      rank += (k2.x > mine ? 1 : 0) + (k2.y > mine ? 1 : 0);
    }
  }
  if (j >= cnt) return;  // (the matrix looks at the matn best-scored ones, the scan's continuation at the rest)
  float4 bb = w.raw[base + j];
  if (use_offsets) {
    const float maxc = w.nanflag[img] ? NAN : unsortable(w.maxbits[img]);
    const float o = w.label[base + j] * (maxc + 1.f);
    bb.x = bb.x + o; bb.y = bb.y + o; bb.z = bb.z + o; bb.w = bb.w + o;
  }
  w.off[base + rank] = bb;
  w.sslot[base + rank] = j;
}

// matrix: item = (image, block row bi, block column bj >= bi) of 64 x 64 candidates; persistent grid, one wave per item.
// Lane i owns sorted candidate bi * 64 + i and tests it against the 64 column candidates (broadcast by readlane).
// parts = 4 (small batches: fewer items than SIMDs): an item is split into four waves of 16 columns, each storing its quarter of
// the 64-bit word.
__global__ __launch_bounds__(256) void nms_matrix_kernel(NmsWs w, int n, float iou_thresh, int parts) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  // items of image g: W_g * W_g (the lower-triangular ones are skipped), W_g = blocks of 64 candidates
  int img = 0, first = 0;  // first item index of image img
  int W = 0;
  auto blocks_of = [&](int g) {
    const int c = w.cand_count[g] < w.matn ? w.cand_count[g] : w.matn;
    return w.legacy ? 0 : (c + 63) >> 6;
  };
  W = blocks_of(0);
  const int cols = 64 / parts;
  for (int qp = wave_global;; qp += nwaves) {
    const int q = qp / parts, part = qp - q * parts;
    while (img < n && q >= first + W * W) {
      first += W * W;
      ++img;
      W = img < n ? blocks_of(img) : 0;
    }
    if (img >= n) return;
    const int r = q - first;
    const int bi = r / W, bj = r - bi * W;
    if (bj < bi) continue;
    const int cnt = w.cand_count[img] < w.matn ? w.cand_count[img] : w.matn;
    const long long base = (long long)img * w.cap;
    const int row = bi * 64 + lane, col = bj * 64 + lane;
    float4 rb = make_float4(0.f, 0.f, 0.f, 0.f), cb = rb;
    if (row < cnt) rb = w.off[base + row];
    if (col < cnt) cb = w.off[base + col];
    const float ra = box_area(rb), ca = box_area(cb);
    unsigned long long bits = 0ull;
#pragma unroll 8
    for (int b = part * cols; b < (part + 1) * cols; ++b) {
      float4 bjx;
      bjx.x = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(cb.x), b));
      bjx.y = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(cb.y), b));
      bjx.z = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(cb.z), b));
      bjx.w = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(cb.w), b));
      const float ja = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(ca), b));
      const int j = bj * 64 + b;
      const bool live = j > row && j < cnt;
      if (iou_thresh >= 0.f) {
        // boxes that do not overlap have inter = 0: 0 / union is 0 (or NaN) and never exceeds a non-negative threshold; when
        // no row of the wave overlaps column b - the common case with class offsets - the division is skipped (uniform branch)
        const float xx1 = (rb.x < bjx.x) ? bjx.x : rb.x, yy1 = (rb.y < bjx.y) ? bjx.y : rb.y;
        const float xx2 = (bjx.z < rb.z) ? bjx.z : rb.z, yy2 = (bjx.w < rb.w) ? bjx.w : rb.w;
        const bool pos = live && (0.f < xx2 - xx1) && (0.f < yy2 - yy1);
        if (__builtin_amdgcn_ballot_w64(pos) == 0ull) continue;
        if (pos && iou_exceeds(rb, ra, bjx, ja, iou_thresh)) bits |= 1ull << b;
      } else if (live && iou_exceeds(rb, ra, bjx, ja, iou_thresh)) {
        bits |= 1ull << b;
      }
    }
    if (row < cnt) {
      unsigned long long* word = &w.mat[((long long)img * w.matn + row) * w.wc + bj];
      if (parts == 1) *word = bits;
      else reinterpret_cast<unsigned short*>(word)[part] = (unsigned short)(bits >> (16 * part));
    }
  }
}

// scan: one workgroup per image.  Block b's 64 matrix rows (64 x wc words) sit in LDS; the rows of block b + 1 are loaded
// into registers before wave 0 resolves block b, and stored to the other LDS buffer afterwards.
constexpr int SCAN_THREADS = 1024;
constexpr int CONT_KEEP = 1024;  // the continuation keeps the winners' boxes in LDS: max_det up to this (batched NMS: 200)

// bulk != 0: the whole bit matrix of the image (matn x wc words, <= 128 KB) is read into LDS up front - eight 16-byte loads in
// flight per thread, one memory latency - and wave 0 walks the blocks without a barrier in between; the two-buffer pipeline
// below paid a fetch latency and two 16-wave barriers per block of 64 candidates (2.5 us x 16 blocks at any batch size).
__global__ __launch_bounds__(SCAN_THREADS) void nms_scan_kernel(NmsWs w, int max_det, float iou_thresh, int* out_count,
                                                                int bulk) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) unsigned char scan_lds[];
  const int img = blockIdx.x;
  const int cnt_all = w.cand_count[img];
  if (cnt_all == 0 || w.legacy) return;  // (cnt == 0: the legacy kernel writes the zero count)
  const int cnt = cnt_all < w.matn ? cnt_all : w.matn;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  constexpr int NWV = SCAN_THREADS / 64;
  const int wc = w.wc;                                   // words per matrix row (<= 16)
  const int nblk = (cnt + 63) >> 6;
  unsigned long long* buf[2] = {reinterpret_cast<unsigned long long*>(scan_lds),
                                reinterpret_cast<unsigned long long*>(scan_lds) + 64 * wc};
  int* s_keep = reinterpret_cast<int*>(reinterpret_cast<unsigned long long*>(scan_lds) +
                                       (bulk ? w.matn : 128) * wc);  // [keep_cap] sorted positions
  __shared__ int s_stop, s_kept;
  __shared__ unsigned long long s_mask[NWV], s_diag[64];
  const unsigned long long* mat = w.mat + (long long)img * w.matn * wc;
  const long long base = (long long)img * w.cap;
  // a block = 64 rows x wc words; thread t moves 16-byte piece t (64 * wc / 2 <= 512 pieces)
  const int pieces = 64 * wc / 2;
  ulonglong2 stage = make_ulonglong2(0ull, 0ull);
  auto fetch = [&](int blk) {
    stage = make_ulonglong2(0ull, 0ull);
    if (t < pieces) {
      const int row = blk * 64 + (t * 2) / wc;
      if (row < cnt) stage = *reinterpret_cast<const ulonglong2*>(mat + (long long)blk * 64 * wc + t * 2);
    }
  };
  auto put = [&](int which) {
    if (t < pieces) *reinterpret_cast<ulonglong2*>(buf[which] + t * 2) = stage;
  };
  if (t == 0) {
    s_stop = 0;
    s_kept = 0;
  }
  if (!bulk) {
    fetch(0);
    put(0);
  }
  __syncthreads();
  unsigned long long rem = 0ull;  // wave 0: lane l = removed bits of candidates 64 l .. 64 l + 63
  int kept = 0;
  // the serial part of a block, shared by both phases: walk the not-yet-removed candidates of the 64-bit word in order, OR the
  // diagonal word of every winner in (lane i holds row i's word in dlo / dhi); returns the winners' mask
  auto resolve = [&](unsigned long long curbits, unsigned long long valid, unsigned dlo, unsigned dhi, int blk) {
    unsigned long long avail = ~curbits & valid;
    unsigned long long keptmask = 0ull;
    const int before = kept;
    while (avail != 0ull && kept < max_det) {  // (scalar unit only: ctz, two readlanes, a few 64-bit bit operations per winner)
      const int i = __builtin_ctzll(avail);
      keptmask |= 1ull << i;
      ++kept;
      const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)dlo, i);
      const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)dhi, i);
      curbits |= ((unsigned long long)hi << 32) | lo;
      avail = ~curbits & valid & (i == 63 ? 0ull : (~0ull << (i + 1)));
    }
    if ((keptmask >> lane) & 1ull)  // the winners' sorted positions, in order, by all lanes at once
      s_keep[before + __builtin_popcountll(keptmask & ((1ull << lane) - 1ull))] = (blk << 6) + lane;
    return keptmask;
  };
  // wave 0, block blk with its 64 matrix rows at B
  auto wave0_block = [&](const unsigned long long* B, int blk) {
    // diagonal word of this lane's row (lower-triangular / stale words are never read: word index >= block index)
    const unsigned long long diag = B[lane * wc + blk];
    const unsigned rlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)rem, blk);
    const unsigned rhi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(rem >> 32), blk);
    const int last = cnt - (blk << 6);
    const unsigned long long valid = last >= 64 ? ~0ull : ((1ull << last) - 1ull);
    const unsigned long long keptmask =
        resolve(((unsigned long long)rhi << 32) | rlo, valid, (unsigned)diag, (unsigned)(diag >> 32), blk);
    // the winners' rows into the removed mask: lane = (group g of four, word); four winners per LDS read, the groups merged by
    // two butterfly steps; lane l < wc ends up with word l like before
    const int word = lane & 15, g = lane >> 4;
    unsigned long long km = keptmask, acc = 0ull;
    while (km != 0ull) {
      int pick[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        pick[u] = km != 0ull ? __builtin_ctzll(km) : -1;
        km &= km - 1ull;  // (0 stays 0)
      }
      const int mine = g == 0 ? pick[0] : (g == 1 ? pick[1] : (g == 2 ? pick[2] : pick[3]));
      if (mine >= 0 && word < wc) acc |= B[mine * wc + word];
    }
    acc |= __shfl_xor(acc, 16, 64);
    acc |= __shfl_xor(acc, 32, 64);
    rem |= acc;
  };
  if (bulk) {
    unsigned long long* all = reinterpret_cast<unsigned long long*>(scan_lds);
    const int pieces_all = nblk * 64 * wc / 2;  // rows >= cnt of the last block were never written: zeros
    for (int p0 = 0; p0 < pieces_all; p0 += 8 * SCAN_THREADS) {
      ulonglong2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int pc = p0 + u * SCAN_THREADS + t;
        v[u] = make_ulonglong2(0ull, 0ull);
        if (pc < pieces_all && (pc * 2) / wc < cnt) v[u] = *reinterpret_cast<const ulonglong2*>(mat + (long long)pc * 2);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int pc = p0 + u * SCAN_THREADS + t;
        if (pc < pieces_all) *reinterpret_cast<ulonglong2*>(all + (long long)pc * 2) = v[u];
      }
    }
    __syncthreads();
    if (wv == 0) {
      for (int blk = 0; blk < nblk && kept < max_det; ++blk) wave0_block(all + (long long)blk * 64 * wc, blk);
      if (lane == 0) s_kept = kept;
    }
  } else
  for (int blk = 0; blk < nblk; ++blk) {
    const int cur = blk & 1;
    const bool more = blk + 1 < nblk;
    if (more) fetch(blk + 1);  // in flight while wave 0 works
    if (wv == 0) {
      wave0_block(buf[cur], blk);
      if (lane == 0) {
        s_kept = kept;
        if (kept >= max_det) s_stop = 1;
      }
    }
    __syncthreads();             // wave 0 is done with buf[cur]; s_stop visible
    if (s_stop) break;
    if (more) put(cur ^ 1);
    __syncthreads();
  }
  __syncthreads();
  int total = s_kept;
  if (total < max_det && cnt_all > cnt) {
    // ---- continuation behind the matrix: the walk ran off the matn best-scored candidates without max_det winners.  The rest of
    // the sorted list is processed 64 candidates at a time with the IoU tests done on demand by this workgroup: (A) each block
    // against the winners so far (their boxes sit in LDS; wave v takes winners v, v + 16, ...; lane = candidate; ballot),
    // (B) the block's own 64 x 64 tests (wave v: rows v, v + 16, ...), (C) the same serial resolve.  Linear in the number of
    // candidates - (winners + 64) tests each - where the matrix is quadratic.
    if (max_det > CONT_KEEP) {  // (plain nms on thousands of boxes keeps everything: the single-workgroup kernel redoes it)
      if (t == 0) w.fallback[img] = 1;
      return;
    }
    float4* kbox = reinterpret_cast<float4*>(scan_lds);   // the matrix buffers are free now: [CONT_KEEP] boxes
    __syncthreads();
    for (int q = t; q < total; q += SCAN_THREADS) kbox[q] = w.off[base + s_keep[q]];
    __syncthreads();
    kept = total;  // (every wave tracks the count; wave 0's resolve is replayed by all through s_kept)
    for (int blk = w.matn >> 6; (blk << 6) < cnt_all && kept < max_det; ++blk) {
      const int j = (blk << 6) + lane;
      float4 cb = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < cnt_all) cb = w.off[base + j];
      const float ca = box_area(cb);
      bool sup = false;
      for (int k = wv; k < kept; k += NWV) {
        const float4 kb = kbox[k];
        sup = sup || iou_exceeds(kb, box_area(kb), cb, ca, iou_thresh);
      }
      const unsigned long long m = __builtin_amdgcn_ballot_w64(sup);
      if (lane == 0) s_mask[wv] = m;
      for (int i = wv; i < 64; i += NWV) {   // diagonal rows: row i against the columns j > i of the block
        float4 rb;
        rb.x = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(cb.x), i));
        rb.y = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(cb.y), i));
        rb.z = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(cb.z), i));
        rb.w = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(cb.w), i));
        const float ra = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(ca), i));
        const bool hit = lane > i && j < cnt_all && iou_exceeds(rb, ra, cb, ca, iou_thresh);
        const unsigned long long word = __builtin_amdgcn_ballot_w64(hit);
        if (lane == 0) s_diag[i] = word;
      }
      __syncthreads();
      if (wv == 0) {
        unsigned long long removed = 0ull;
        for (int v = 0; v < NWV; ++v) removed |= s_mask[v];
        const unsigned long long diag = s_diag[lane];
        const int last = cnt_all - (blk << 6);
        const unsigned long long valid = last >= 64 ? ~0ull : ((1ull << last) - 1ull);
        const int before = kept;
        const unsigned long long keptmask = resolve(removed, valid, (unsigned)diag, (unsigned)(diag >> 32), blk);
        if ((keptmask >> lane) & 1ull)  // winners append their boxes in order
          kbox[before + __builtin_popcountll(keptmask & ((1ull << lane) - 1ull))] = cb;
        if (lane == 0) s_kept = kept;
      }
      __syncthreads();
      kept = s_kept;
    }
    total = kept;
  }
  __syncthreads();
  for (int q = t; q < total; q += SCAN_THREADS) w.keep_slot[base + q] = w.sslot[base + s_keep[q]];
  if (t == 0) out_count[img] = total;
}

constexpr size_t kSelectLds = 112 * 1024;  // max(LDS_CANDS * 24, matrix mode: keys + boxes + sorted copies + 768 x 12 x 8 B bits)

inline hipError_t select_lds_attr() {  // > 64 KiB of dynamic LDS needs the opt-in, once per process
  static hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void*>(nms_select_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSelectLds);
  return rc;
}

// ---- emit -------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void nms_emit_kernel(const float* pred, int rows, int num_classes, int max_det,
                                                       NmsWs w, const int* count, float* det) {
  const int img = blockIdx.y, k = blockIdx.x;
  if (k >= count[img]) return;
  const long long base = (long long)img * w.cap;
  const int slot = w.keep_slot[base + k];
  const unsigned long long key = w.key[base + slot];
  const int row = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
  const int per = 5 + num_classes, width = 7 + num_classes;
  const float* p = pred + ((long long)img * rows + row) * per;
  float* d = det + ((long long)img * max_det + k) * width;
  const float4 b = w.raw[base + slot];
  for (int c = threadIdx.x; c < width; c += 128) {
    float v;
    if (c == 0) v = b.x;
    else if (c == 1) v = b.y;
    else if (c == 2) v = b.z;
    else if (c == 3) v = b.w;
    else if (c == 4) v = p[4];
    else if (c == 5) v = w.clsconf[base + slot];
    else if (c == 6) v = w.label[base + slot];
    else v = p[5 + (c - 7)];
    d[c] = v;
  }
}

__global__ __launch_bounds__(256) void nms_emit_indices_kernel(NmsWs w, const int* count, long long* keep) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= count[0]) return;
  const int slot = w.keep_slot[k];
  const unsigned long long key = w.key[slot];
  keep[k] = (long long)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
}

// the three launches of the matrix path (every kernel returns at once for images it does not own)
inline int launch_matrix_path(const NmsWs& w, int n, int use_offsets, float iou_thresh, int max_det, int* out_count,
                              hipStream_t stream) {
  if (w.legacy) return 0;
  hipLaunchKernelGGL(nms_rank_kernel, dim3((w.cap + 255) / 256, n), dim3(256), 0, stream, w, use_offsets);
  int rc = me::check_launch("nms_rank_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(nms_matrix_kernel, dim3(1024), dim3(256), 0, stream, w, n, iou_thresh, n * 136 < 2048 ? 4 : 1);
  rc = me::check_launch("nms_matrix_kernel");
  if (rc) return rc;
  const int keep_cap = max_det < w.cap ? max_det : w.cap;
  static const int bulk_env = getenv("MILLIEYE_NMS_BULK") ? atoi(getenv("MILLIEYE_NMS_BULK")) : 1;
  const int bulk = bulk_env && (size_t)w.matn * w.wc * 8 + (size_t)keep_cap * 4 <= 150 * 1024;  // the whole matrix in LDS
  size_t lds = (size_t)(bulk ? w.matn : 128) * w.wc * 8;                     // all / two blocks of matrix rows ...
  if (max_det <= CONT_KEEP && lds < (size_t)CONT_KEEP * 16) lds = (size_t)CONT_KEEP * 16;  // ... or the continuation's winner boxes
  lds += (size_t)keep_cap * 4;
  static bool attr = false;
  if (!attr) {
    ME_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nms_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               150 * 1024));
    attr = true;
  }
  ME_REQUIRE(lds <= 150 * 1024, ME_E_TOOBIG, "me_nms: max_det %d too large for the scan kernel's LDS", max_det);
  hipLaunchKernelGGL(nms_scan_kernel, dim3(n), dim3(SCAN_THREADS), lds, stream, w, max_det, iou_thresh, out_count, bulk);
  return me::check_launch("nms_scan_kernel");
}

}  // namespace

extern "C" {

int64_t me_nms_workspace_bytes(int32_t n, int32_t rows) {
  if (n <= 0 || rows <= 0) return 0;
  return ws_bytes(n, rows);
}

static int nms_batched(const me_nms_desc* d, void* stream_, int prepped);

int me_nms_batched_f32(const me_nms_desc* d, void* stream) { return nms_batched(d, stream, 0); }

/* the candidate lists are already in the workspace (me_yolo_decode_cand_f32 of every scale): select + emit only */
int me_nms_batched_prepped_f32(const me_nms_desc* d, void* stream) { return nms_batched(d, stream, 1); }

int me_yolo_decode_cand_multi_f32(const me_yolo_desc* const* ys, int32_t count, float conf_thresh, void* nms_workspace,
                                  int32_t first, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(ys && nms_workspace, ME_E_NULLPTR, "me_yolo_decode_cand_f32: null pointer");
  ME_REQUIRE(count >= 1 && count <= 3, ME_E_BADARG, "me_yolo_decode_cand_f32: 1 - 3 scales per launch (got %d)", count);
  ME_REQUIRE((reinterpret_cast<uintptr_t>(nms_workspace) & 255u) == 0, ME_E_ALIGN, "me_yolo_decode_cand_f32: workspace alignment");
  long long rows_all = 0;
  for (int i = 0; i < count; ++i) {
    const me_yolo_desc* y = ys[i];
    ME_REQUIRE(y && y->x && y->out, ME_E_NULLPTR, "me_yolo_decode_cand_f32: null pointer");
    ME_REQUIRE(y->n > 0 && y->g > 0 && y->num_anchors > 0 && y->num_anchors <= 8 && y->num_classes >= 0 &&
                   y->num_classes + 5 <= 128, ME_E_BADARG, "me_yolo_decode_cand_f32: bad dimensions (5 + classes <= 128)");
    ME_REQUIRE(y->x_pitch >= y->num_anchors * (y->num_classes + 5), ME_E_BADARG, "me_yolo_decode_cand_f32: x_pitch too small");
    ME_REQUIRE(y->row_offset >= 0 && y->row_offset + y->num_anchors * y->g * y->g <= y->rows_total && y->rows_total <= MAX_ROWS,
               ME_E_BADARG, "me_yolo_decode_cand_f32: rows out of range");
    ME_REQUIRE(y->n <= 65535 && y->stride > 0.f, ME_E_BADARG, "me_yolo_decode_cand_f32: bad batch / stride");
    ME_REQUIRE(y->n == ys[0]->n && y->rows_total == ys[0]->rows_total && y->out == ys[0]->out, ME_E_BADARG,
               "me_yolo_decode_cand_f32: the scales of one launch share the batch and the prediction tensor");
    rows_all += (long long)y->num_anchors * y->g * y->g;
  }
  const int n = ys[0]->n;
  NmsWs w = carve(nms_workspace, n, ys[0]->rows_total);
  if (first) {
    // a kernel, not hipMemsetAsync: this launch sequence is captured into hipGraphs (engine._run_graph), and on this ROCm (7.2) a
    // captured hipMemsetAsync of 64 bytes or more clears the words on the FIRST replay only - later replays write a pointer-like
    // pattern instead (tools/memset_graph_probe.py, profiles/r06_memset_graph_probe.txt; 8 bytes are fine).  The counts then
    // started at garbage and the lists overran: a memory fault on the second replay (round 6).
    hipLaunchKernelGGL(zero_ints_kernel, dim3((4 * n + 255) / 256), dim3(256), 0, stream, w.cand_count, 4 * n);
    const int rc0 = me::check_launch("zero_ints_kernel");
    if (rc0) return rc0;
  }
  // rows per wave: 16 when that still leaves >= 2048 workgroups, else 4, else 1 (batch 1: 10 647 rows over 2 662 workgroups)
  const long long wg16 = rows_all * n / 64;
  const int rpw = wg16 >= 2048 ? 16 : (wg16 >= 512 ? 4 : 1);
  YoloCand d;
  d.count = count;
  d.conf_thresh = conf_thresh;
  int wgs = 0;
  for (int i = 0; i < 3; ++i) {
    d.y[i] = *ys[i < count ? i : 0];
    if (i < count) wgs += (ys[i]->num_anchors * ys[i]->g * ys[i]->g + 4 * rpw - 1) / (4 * rpw);
    d.wg_end[i] = wgs;
  }
  const dim3 grid(wgs, n), block(256);
  if (rpw == 16) hipLaunchKernelGGL(yolo_decode_cand_kernel<16>, grid, block, 0, stream, d, w);
  else if (rpw == 4) hipLaunchKernelGGL(yolo_decode_cand_kernel<4>, grid, block, 0, stream, d, w);
  else hipLaunchKernelGGL(yolo_decode_cand_kernel<1>, grid, block, 0, stream, d, w);
  return me::check_launch("yolo_decode_cand_kernel");
}

int me_yolo_decode_cand_f32(const me_yolo_desc* y, float conf_thresh, void* nms_workspace, int32_t first, void* stream) {
  return me_yolo_decode_cand_multi_f32(&y, 1, conf_thresh, nms_workspace, first, stream);
}

static int nms_batched(const me_nms_desc* d, void* stream_, int prepped) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(d && d->pred && d->det && d->count && d->workspace, ME_E_NULLPTR, "me_nms_batched_f32: null pointer");
  ME_REQUIRE(d->n > 0 && d->rows > 0 && d->num_classes >= 0 && d->max_det > 0, ME_E_BADARG,
             "me_nms_batched_f32: bad dimensions");
  ME_REQUIRE(d->rows <= MAX_ROWS, ME_E_TOOBIG, "me_nms_batched_f32: rows %d > capacity %d", d->rows, MAX_ROWS);
  ME_REQUIRE(d->n <= 65535, ME_E_TOOBIG, "me_nms_batched_f32: batch too large");
  ME_REQUIRE((reinterpret_cast<uintptr_t>(d->workspace) & 255u) == 0, ME_E_ALIGN,
             "me_nms_batched_f32: workspace not 256-byte aligned");
  NmsWs w = carve(d->workspace, d->n, d->rows);
  int rc = 0;
  if (!prepped) {
    // counts, max-coordinate bits, NaN / fallback flags (a kernel: see me_yolo_decode_cand_multi_f32 on captured memsets)
    hipLaunchKernelGGL(zero_ints_kernel, dim3((4 * d->n + 255) / 256), dim3(256), 0, stream, w.cand_count, 4 * d->n);
    hipLaunchKernelGGL(nms_prep_kernel, dim3((d->rows + 255) / 256, d->n), dim3(256), 0, stream, d->pred, d->rows,
                       d->num_classes, d->conf_thresh, d->writeback_xyxy, w);
    rc = me::check_launch("nms_prep_kernel");
    if (rc) return rc;
  } else {
    ME_REQUIRE(!d->writeback_xyxy, ME_E_BADARG, "me_nms_batched_prepped_f32: the fused decode does not rewrite the rows as xyxy");
  }
  const int max_det = d->max_det < d->rows ? d->max_det : d->rows;
  rc = launch_matrix_path(w, d->n, 1, d->iou_thresh, max_det, d->count, stream);
  if (rc) return rc;
  ME_HIP(select_lds_attr());
  hipLaunchKernelGGL(nms_select_kernel, dim3(d->n), dim3(SEL_THREADS), kSelectLds, stream, w, 1, d->iou_thresh, max_det,
                     d->count);
  rc = me::check_launch("nms_select_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(nms_emit_kernel, dim3(max_det, d->n), dim3(128), 0, stream, d->pred, d->rows, d->num_classes,
                     d->max_det, w, d->count, d->det);
  return me::check_launch("nms_emit_kernel");
}

int me_nms_boxes_f32(const float* boxes, const float* scores, const float* labels, int32_t m, float iou_thresh,
                     int64_t* keep, int32_t* keep_count, void* workspace, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  ME_REQUIRE(keep_count, ME_E_NULLPTR, "me_nms_boxes_f32: null keep_count");
  if (m == 0) {  // batched_nms on an empty set returns an empty index tensor
    ME_HIP(hipMemsetAsync(keep_count, 0, sizeof(int32_t), stream));
    return 0;
  }
  ME_REQUIRE(boxes && scores && keep && workspace, ME_E_NULLPTR, "me_nms_boxes_f32: null pointer");
  ME_REQUIRE(m > 0 && m <= MAX_ROWS, ME_E_TOOBIG, "me_nms_boxes_f32: m %d outside (0, %d]", m, MAX_ROWS);
  ME_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, ME_E_ALIGN,
             "me_nms_boxes_f32: workspace not 256-byte aligned");
  NmsWs w = carve(workspace, 1, m);
  hipLaunchKernelGGL(zero_ints_kernel, dim3(1), dim3(256), 0, stream, w.cand_count, 4);
  hipLaunchKernelGGL(nms_prep_boxes_kernel, dim3((m + 255) / 256), dim3(256), 0, stream, boxes, scores, labels, m, w);
  int rc = me::check_launch("nms_prep_boxes_kernel");
  if (rc) return rc;
  rc = launch_matrix_path(w, 1, labels ? 1 : 0, iou_thresh, m, keep_count, stream);
  if (rc) return rc;
  ME_HIP(select_lds_attr());
  hipLaunchKernelGGL(nms_select_kernel, dim3(1), dim3(SEL_THREADS), kSelectLds, stream, w, labels ? 1 : 0, iou_thresh, m,
                     keep_count);
  rc = me::check_launch("nms_select_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(nms_emit_indices_kernel, dim3((m + 255) / 256), dim3(256), 0, stream, w, keep_count,
                     reinterpret_cast<long long*>(keep));
  return me::check_launch("nms_emit_indices_kernel");
}

}  // extern "C"
