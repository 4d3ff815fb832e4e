// Evaluation path of the Detect head: box decode and non-maximum suppression on the device.
//
//   mmi_detect_decode  replaces models/yolo_test.py:57-66 (sigmoid, grid/anchor decode, view, cat over levels)
//   mmi_nms            replaces utils/general.py:486-580 non_max_suppression (candidate filter, obj*cls confidence,
//                      xywh->xyxy, best-class or multi-label rows, class filter, class-offset batched NMS, max_det), whose
//                      NMS proper is torchvision.ops.nms (not vendored by the reference, no version pinned): greedy,
//                      boxes visited by decreasing score, a box is dropped when IoU > iou_thres with a kept one,
//                      IoU = inter / (area_a + area_b - inter) without epsilon.
//
// NMS without a sort: one workgroup per image repeatedly takes the arg-max of the still-alive scores (ties: lowest
// (row, class) key, so the result does not depend on the order in which candidates were compacted) and suppresses in
// parallel; at most max_det rounds.  Latency-bound integer/compare work, no MFMA.
//
// Deviation: the reference truncates to the 30 000 best candidates before NMS (max_nms, general.py:555-557); here every
// candidate takes part.  Results differ only for an image with more than 30 000 candidates above conf_thres.
#include "common.h"

namespace {

__global__ void detect_decode_kernel(const float* __restrict__ x, float* __restrict__ z, int B, int na, int ny, int nx,
                                     int no, int64_t R, int64_t row_off, float stride, const float* __restrict__ anchor_px) {
  const int64_t P = (int64_t)ny * nx, total = (int64_t)B * na * P * no;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % no);
    int64_t r = i / no;                      // (b, a, p)
    const int64_t p = r % P;
    r /= P;
    const int a = (int)(r % na);
    const int64_t b = r / na;
    const float s = 1.0f / (1.0f + expf(-x[i]));
    float v = s;
    if (c == 0) v = (s * 2.0f - 0.5f + (float)(p % nx)) * stride;
    else if (c == 1) v = (s * 2.0f - 0.5f + (float)(p / nx)) * stride;
    else if (c == 2 || c == 3) {
      const float t = s * 2.0f;
      v = t * t * anchor_px[a * 2 + (c - 2)];
    }
    z[(b * R + row_off + (int64_t)a * P + p) * no + c] = v;
  }
}

struct NmsBuf {
  float* box;    // [B][cap][4] xyxy, not offset
  float* score;  // [B][cap]   (< 0: suppressed)
  int* cls;      // [B][cap]
  int* key;      // [B][cap]   row * nc + class
  int* count;    // [B]
};

__global__ void nms_candidates_kernel(const float* __restrict__ pred, int B, int64_t R, int nc, float conf_thres,
                                      int multi_label, const uint8_t* __restrict__ class_allow, int64_t cap, NmsBuf w) {
  const int64_t total = (int64_t)B * R;
  const int no = nc + 5;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float* p = pred + i * no;
    const float obj = p[4];
    if (!(obj > conf_thres)) continue;
    const int64_t b = i / R, row = i - b * R;
    const float hw = p[2] / 2.0f, hh = p[3] / 2.0f;          // xywh2xyxy (general.py:392-400)
    const float x1 = p[0] - hw, y1 = p[1] - hh, x2 = p[0] + hw, y2 = p[1] + hh;
    auto emit = [&](int j, float conf) {
      const int slot = atomicAdd(w.count + b, 1);
      if (slot >= cap) return;                                 // cannot happen: cap is the worst case
      const int64_t o = b * cap + slot;
      w.box[o * 4 + 0] = x1; w.box[o * 4 + 1] = y1; w.box[o * 4 + 2] = x2; w.box[o * 4 + 3] = y2;
      w.score[o] = conf;
      w.cls[o] = j;
      w.key[o] = (int)(row * nc + j);
    };
    if (multi_label) {
      for (int j = 0; j < nc; ++j) {
        const float conf = p[5 + j] * obj;
        if (conf > conf_thres && (class_allow == nullptr || class_allow[j])) emit(j, conf);
      }
    } else {
      int bj = 0;
      float best = p[5] * obj;
      for (int j = 1; j < nc; ++j) {
        const float conf = p[5 + j] * obj;
        if (conf > best) best = conf, bj = j;                  // first maximum, as torch.max
      }
      if (best > conf_thres && (class_allow == nullptr || class_allow[bj])) emit(bj, best);
    }
  }
}

__global__ __launch_bounds__(1024) void nms_greedy_kernel(NmsBuf w, int64_t cap, float iou_thres, float class_offset,
                                                          int max_det, float* __restrict__ out, int* __restrict__ nout) {
  __shared__ float s_score[16];
  __shared__ int s_key[16], s_idx[16];
  __shared__ float s_sel[4];
  __shared__ int s_stop;
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n = min((int64_t)w.count[b], cap);
  float* box = w.box + (int64_t)b * cap * 4;
  float* score = w.score + (int64_t)b * cap;
  const int* cls = w.cls + (int64_t)b * cap;
  const int* key = w.key + (int64_t)b * cap;
  int kept = 0;
  for (; kept < max_det; ++kept) {
    float bs = -1.0f;
    int bk = 0x7fffffff, bi = -1;
    for (int j = t; j < n; j += 1024) {
      const float s = score[j];
      if (s > bs || (s == bs && s >= 0.0f && key[j] < bk)) bs = s, bk = key[j], bi = j;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float os = __shfl_xor(bs, off);
      const int ok = __shfl_xor(bk, off), oi = __shfl_xor(bi, off);
      if (os > bs || (os == bs && os >= 0.0f && ok < bk)) bs = os, bk = ok, bi = oi;
    }
    if (lane == 0) s_score[wave] = bs, s_key[wave] = bk, s_idx[wave] = bi;
    __syncthreads();
    if (t == 0) {
      for (int v = 1; v < 16; ++v)
        if (s_score[v] > bs || (s_score[v] == bs && bs >= 0.0f && s_key[v] < bk)) bs = s_score[v], bk = s_key[v], bi = s_idx[v];
      s_stop = bi < 0 || bs < 0.0f;
      if (!s_stop) {
        const float c = (float)cls[bi] * class_offset;
        float* o = out + ((int64_t)b * max_det + kept) * 6;
        for (int e = 0; e < 4; ++e) {
          o[e] = box[bi * 4 + e];
          s_sel[e] = box[bi * 4 + e] + c;                     // boxes + c (general.py:560-562), fp32 as the reference
        }
        o[4] = bs;
        o[5] = (float)cls[bi];
        score[bi] = -1.0f;
      }
    }
    __syncthreads();
    if (s_stop) break;
    const float ax1 = s_sel[0], ay1 = s_sel[1], ax2 = s_sel[2], ay2 = s_sel[3];
    const float aarea = (ax2 - ax1) * (ay2 - ay1);
    for (int j = t; j < n; j += 1024) {
      if (score[j] < 0.0f) continue;
      const float c = (float)cls[j] * class_offset;
      const float bx1 = box[j * 4] + c, by1 = box[j * 4 + 1] + c, bx2 = box[j * 4 + 2] + c, by2 = box[j * 4 + 3] + c;
      const float iw = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.0f), ih = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.0f);
      const float inter = iw * ih;
      const float iou = inter / (aarea + (bx2 - bx1) * (by2 - by1) - inter);
      if (iou > iou_thres) score[j] = -1.0f;
    }
    __syncthreads();
  }
  if (t == 0) nout[b] = kept;
}

inline int nms_blocks(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}
inline int64_t nms_cap(int64_t R, int nc, int multi_label) { return R * (multi_label ? nc : 1); }

}  // namespace

extern "C" int mmi_detect_decode(const float* x, float* z, int B, int na, int ny, int nx, int no, int64_t total_rows,
                                 int64_t row_offset, float stride, const float* anchor_px, void* stream) {
  MMI_CHECK_ARG(x && z && anchor_px && B > 0 && na > 0 && ny > 0 && nx > 0 && no > 5, "mmi_detect_decode: bad arguments");
  MMI_CHECK_ARG(row_offset >= 0 && row_offset + (int64_t)na * ny * nx <= total_rows, "mmi_detect_decode: level does not fit the output");
  hipLaunchKernelGGL(detect_decode_kernel, dim3(nms_blocks((int64_t)B * na * ny * nx * no)), dim3(256), 0, (hipStream_t)stream,
                     x, z, B, na, ny, nx, no, total_rows, row_offset, stride, anchor_px);
  MMI_CHECK_LAUNCH("mmi_detect_decode");
  return MMI_OK;
}

extern "C" size_t mmi_nms_workspace(int B, int64_t R, int nc, int multi_label) {
  if (B <= 0 || R <= 0 || nc <= 0) return 0;
  const int64_t cap = nms_cap(R, nc, multi_label);
  return (size_t)B * cap * (4 + 1 + 1 + 1) * 4 + (size_t)B * 4 + 64;
}

extern "C" int mmi_nms(const float* pred, int B, int64_t R, int nc, float conf_thres, float iou_thres,
                       const uint8_t* class_allow, int agnostic, int multi_label, int max_det, float max_wh, void* workspace, size_t workspace_bytes,
                       float* out, int* nout, void* stream) {
  MMI_CHECK_ARG(pred && out && nout && workspace && B > 0 && R > 0 && nc > 0 && max_det > 0, "mmi_nms: bad arguments");
  MMI_CHECK_ARG((int64_t)R * nc < (1LL << 31), "mmi_nms: too many (row, class) pairs");
  if (workspace_bytes < mmi_nms_workspace(B, R, nc, multi_label)) {
    mmi_set_error("mmi_nms: workspace too small (%zu < %zu)", workspace_bytes, mmi_nms_workspace(B, R, nc, multi_label));
    return MMI_ERR_WORKSPACE;
  }
  multi_label = (multi_label && nc > 1) ? 1 : 0;                // general.py:505
  const int64_t cap = nms_cap(R, nc, multi_label);
  hipStream_t s = (hipStream_t)stream;
  NmsBuf w;
  char* base = (char*)workspace;
  w.count = (int*)base;
  base += ((size_t)B * 4 + 63) / 64 * 64;
  w.box = (float*)base;
  base += (size_t)B * cap * 16;
  w.score = (float*)base;
  base += (size_t)B * cap * 4;
  w.cls = (int*)base;
  base += (size_t)B * cap * 4;
  w.key = (int*)base;
  if (int e = mmi_fill_bytes(w.count, 0, (size_t)B * 4, s)) return e;
  hipLaunchKernelGGL(nms_candidates_kernel, dim3(nms_blocks((int64_t)B * R)), dim3(256), 0, s, pred, B, R, nc, conf_thres,
                     multi_label, class_allow, cap, w);
  MMI_CHECK_LAUNCH("mmi_nms(candidates)");
  hipLaunchKernelGGL(nms_greedy_kernel, dim3(B), dim3(1024), 0, s, w, cap, iou_thres, agnostic ? 0.0f : max_wh, max_det, out,
                     nout);
  MMI_CHECK_LAUNCH("mmi_nms(greedy)");
  return MMI_OK;
}
