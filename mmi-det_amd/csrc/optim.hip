// Fused multi-tensor SGD(nesterov) + ModelEMA update: ONE launch for every parameter / buffer tensor of the model, with
// the hyper-parameters (per-group lr and weight decay, momentum, EMA decay) read from device memory so the launch can sit
// inside a captured hipGraph while the schedule keeps moving.  Pure HBM streaming: p, g, momentum buffer, EMA each
// touched once per step (~7 GB at yolov5l) instead of ~10 passes of per-tensor ATen kernels.
//
// Replaces optimizer.step() + ema.update(model) of the reference's step (train.py:799-804; torch.optim.SGD nesterov with
// the three parameter groups of train.py:572-589; utils/torch_utils.py:269-299 ModelEMA: d = decay*(1-exp(-n/2000)),
// applied to every floating state_dict entry, buffers included).  SURVEY.md §8f-1.
#include "common.h"

namespace {

struct OptRec {  // must match the numpy dtype in mmidet_hip/optim.py
  float* p;          // parameter (or buffer, for EMA-only records)
  const float* g;    // gradient (SGD records)
  float* buf;        // momentum buffer (SGD records)
  float* ema;        // EMA shadow (0 if none)
  int64_t n;
  int32_t group;     // 0..2 -> hyper lr/wd slot
  int32_t flags;     // 1 = SGD, 2 = EMA, 4 = every pointer 16-byte aligned
};
struct OptChunk {
  int32_t rec, chunk;
};

// hyper: [0..2] lr, [3..5] weight decay, [6] momentum, [7] EMA decay d, [8] != 0 on the very first step
__global__ __launch_bounds__(256) void sgd_ema_kernel(const OptRec* __restrict__ recs, const OptChunk* __restrict__ chunks,
                                                      const float* __restrict__ hyper) {
  const OptChunk ck = chunks[blockIdx.x];
  const OptRec r = recs[ck.rec];
  const int64_t beg = (int64_t)ck.chunk * MMI_OPT_CHUNK;
  const int64_t end = min(beg + (int64_t)MMI_OPT_CHUNK, r.n);
  const bool sgd = r.flags & 1, has_ema = (r.flags & 2) && r.ema != nullptr;
  const float lr = hyper[r.group], wd = hyper[3 + r.group], mom = hyper[6], d = hyper[7];
  const bool first = hyper[8] != 0.f;
  const float omd = 1.0f - d;
  if ((r.flags & 4) && ((end - beg) & 3) == 0) {
    for (int64_t i = beg + (int64_t)threadIdx.x * 4; i < end; i += 256 * 4) {
      f32x4 p = *reinterpret_cast<const f32x4*>(r.p + i);
      if (sgd) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(r.g + i);
        f32x4 dp = g + wd * p;
        f32x4 b = dp;
        if (!first) b = mom * *reinterpret_cast<const f32x4*>(r.buf + i) + dp;
        *reinterpret_cast<f32x4*>(r.buf + i) = b;
        dp = dp + mom * b;  // nesterov
        p = p - lr * dp;
        *reinterpret_cast<f32x4*>(r.p + i) = p;
      }
      if (has_ema) {
        const f32x4 e = *reinterpret_cast<const f32x4*>(r.ema + i);
        *reinterpret_cast<f32x4*>(r.ema + i) = d * e + omd * p;
      }
    }
  } else {
    for (int64_t i = beg + threadIdx.x; i < end; i += 256) {
      float p = r.p[i];
      if (sgd) {
        float dp = r.g[i] + wd * p;
        const float b = first ? dp : mom * r.buf[i] + dp;
        r.buf[i] = b;
        dp = dp + mom * b;
        p = p - lr * dp;
        r.p[i] = p;
      }
      if (has_ema) r.ema[i] = d * r.ema[i] + omd * p;
    }
  }
}

}  // namespace

extern "C" int mmi_sgd_ema_step(const void* recs_dev, const void* chunks_dev, int nchunks, const float* hyper_dev,
                                void* stream) {
  MMI_CHECK_ARG(recs_dev && chunks_dev && hyper_dev && nchunks > 0, "mmi_sgd_ema_step: bad arguments");
  static_assert(sizeof(OptRec) == 48 && sizeof(OptChunk) == 8, "record layout is part of the ABI");
  hipLaunchKernelGGL(sgd_ema_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const OptRec*)recs_dev,
                     (const OptChunk*)chunks_dev, hyper_dev);
  MMI_CHECK_LAUNCH("mmi_sgd_ema_step");
  return MMI_OK;
}
