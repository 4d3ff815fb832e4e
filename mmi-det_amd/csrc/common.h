// Shared helpers for libmmidet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <initializer_list>

#include "../../include/mmidet_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

void mmi_set_error(const char* fmt, ...);

#define MMI_CHECK_ARG(cond, ...)      \
  do {                                \
    if (!(cond)) {                    \
      mmi_set_error(__VA_ARGS__);     \
      return MMI_ERR_ARG;             \
    }                                 \
  } while (0)

#define MMI_CHECK_LAUNCH(name)                                             \
  do {                                                                     \
    hipError_t e_ = hipGetLastError();                                     \
    if (e_ != hipSuccess) {                                                \
      mmi_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return MMI_ERR_LAUNCH;                                               \
    }                                                                      \
  } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- library-internal cross-file helpers (not part of the C ABI) ----------------------------------------------------
// bn.hip: fp64 fold of partials[part][2][C] into out0[C] (slot 0) and out1[C] (slot 1)
int mmi_pair_colsum(float* partials, int nparts, int C, float* out0, float* out1, void* stream);  // consumes partials
int mmi_i64_increment(int64_t* counter, void* stream);   // bn.hip: *counter += 1
// api.hip: fill `bytes` bytes (a multiple of 4, 4-byte aligned) with the byte `value` -- a KERNEL, not hipMemsetAsync: inside a
// captured step a memset NODE was observed out of order with its neighbouring kernel nodes when the whole step is one stream
// (ROCm 7.2; profiles/r03_graph_memset_nodes.txt), a kernel node is ordered like every other launch
int mmi_fill_bytes(void* ptr, int value, size_t bytes, hipStream_t stream);
// cem.hip: direct VALU convolutions for the 3<->24-channel CEM layers, reached through the public conv entry points
bool mmi_smallconv_supported(const mmi_conv_desc* d);
bool mmi_smallconv_dgrad_supported(const mmi_conv_desc* d);
int mmi_smallconv_blocks(const mmi_conv_desc* d);
int mmi_smallconv_fwd(const float* x, const float* w, const float* bias, float* y, float* stat_part, const mmi_conv_desc* d,
                      hipStream_t s);
int mmi_smallconv_dgrad(const float* dy, const float* w, float* dx, const mmi_conv_desc* d, hipStream_t s);
size_t mmi_smallconv_wgrad_workspace(const mmi_conv_desc* d);
int mmi_smallconv_wgrad(const float* dy, const float* x, float* dw, void* workspace, const mmi_conv_desc* d, hipStream_t s);

// XCD-aware bijective remap of a linear workgroup id: blocks b and b+8 share an XCD (observed round-robin), so give
// each XCD a contiguous chunk of the tile sequence; neighbouring tiles (same activation rows, all output-channel
// tiles) then hit the same 4 MiB L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}

// ---- last-arriver fold of per-row-block column statistics (BatchNorm forward statistics out of the conv epilogue,
// dgamma/dbeta out of the BN backward reduce pass) --------------------------------------------------------------------
// Every workgroup of the producing kernel writes one row of part[row][2][C] (two sums per channel, its BNC columns starting
// at c0) with agent-scope stores and then calls stat_arrive().  Rows are grouped G at a time: the LAST workgroup to arrive
// in a group folds the group's rows into l1[group][2][C]; the last GROUP to finish folds l1 and returns true with the
// column totals (fp64) in threads t < BNC.  Nobody waits for anybody; the fold order is fixed (row order), so the result
// does not depend on who arrives last: run-to-run bit-identical, no float atomics, and no separate "finalize" launch.
// cnt[ngroups * nct + nct] arrival counters: zero before the launch, zero again after it (the electing thread resets).
struct StatFold {
  float* part;  // [nrows][2][C]
  float* l1;    // [ngroups][2][C]
  int* cnt;
  int nrows, C, nct, G;
};
// Publishing hand-off used by every last-arriver fold (MI355X_MICROARCH.md, "inter-workgroup visibility", first row of the
// sc1 hand-off table): payload stored with sc1 (st_agent) -> EVERY storing wave waits for its own stores (this call) ->
// workgroup barrier -> one lane's agent-scope atomic add; the workgroup whose add came last reads with sc1 loads (ld_agent)
// behind a workgroup barrier.  A workgroup-scope release fence is NOT enough here: on gfx950 it lowers to lgkmcnt(0) only
// and leaves the global stores in flight (round-2 advisor finding).  Inline asm so that no compiler pass can drop the wait.
__device__ __forceinline__ void mmi_drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// fold rows [r0, r1) of src[row][2][C], columns [c0, c0 + BNC): totals in threads t < BNC (red: LDS, 2 * 256 doubles)
template <int BNC>
__device__ __forceinline__ void stat_fold_rows(const float* src, int r0, int r1, int C, int c0, double* red, double& s1, double& s2) {
  constexpr int RL = 256 / BNC;
  const int t = threadIdx.x, col = t % BNC, rl = t / BNC;
  double a = 0.0, b = 0.0;
  if (c0 + col < C) {
    constexpr int U = 4;  // the loads of one round are independent: a round costs one memory latency, not U
    const float* base = src + c0 + col;
    for (int r = r0 + rl; r < r1; r += RL * U) {
      float va[U], vb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int rr = r + RL * u;
        const bool ok = rr < r1;
        const int64_t o = (int64_t)(ok ? rr : r) * 2 * C;
        va[u] = ld_agent(base + o);
        vb[u] = ld_agent(base + o + C);
        if (!ok) va[u] = vb[u] = 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        a += (double)va[u];
        b += (double)vb[u];
      }
    }
  }
  __syncthreads();  // red may still be read by the previous fold
  red[(0 * RL + rl) * BNC + col] = a;
  red[(1 * RL + rl) * BNC + col] = b;
  __syncthreads();
  s1 = s2 = 0.0;
  if (t < BNC) {
#pragma unroll
    for (int i = 0; i < RL; ++i) {
      s1 += red[(0 * RL + i) * BNC + t];
      s2 += red[(1 * RL + i) * BNC + t];
    }
  }
}

// all 256 threads of the workgroup, after its row `row` (columns of tile `ct`) has been stored with st_agent()
template <int BNC>
__device__ __forceinline__ bool stat_arrive(const StatFold& f, int row, int ct, int c0, double* red, int* flag, double& s1, double& s2) {
  const int t = threadIdx.x;
  const int ngroups = (f.nrows + f.G - 1) / f.G;
  const int g = row / f.G, r0 = g * f.G, r1 = min(r0 + f.G, f.nrows);
  mmi_drain_stores();  // every storing wave: its sc1 (write-through) stores have left the CU before the barrier below
  __syncthreads();
  if (t == 0) {
    int* c = f.cnt + g * f.nct + ct;
    const int last = __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == r1 - r0 - 1;
    if (last) __hip_atomic_store(c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = last;
  }
  __syncthreads();
  if (!*flag) return false;  // uniform
  stat_fold_rows<BNC>(f.part, r0, r1, f.C, c0, red, s1, s2);
  if (t < BNC && c0 + t < f.C) {
    st_agent(f.l1 + ((int64_t)g * 2 + 0) * f.C + c0 + t, (float)s1);
    st_agent(f.l1 + ((int64_t)g * 2 + 1) * f.C + c0 + t, (float)s2);
  }
  mmi_drain_stores();  // every storing wave: its sc1 (write-through) stores have left the CU before the barrier below
  __syncthreads();
  if (t == 0) {
    int* c = f.cnt + ngroups * f.nct + ct;
    const int last = __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1;
    if (last) __hip_atomic_store(c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = last;
  }
  __syncthreads();
  if (!*flag) return false;
  stat_fold_rows<BNC>(f.l1, 0, ngroups, f.C, c0, red, s1, s2);
  return true;
}
constexpr int MMI_STAT_GROUP = 32;            // rows per group (raised for very long lists so that ngroups <= 1024)
constexpr int MMI_STAT_MAX_COUNTERS = 16384;  // ints reserved for the arrival counters in a workspace
static inline int stat_group_size(int nrows) {
  int g = MMI_STAT_GROUP;
  while ((nrows + g - 1) / g > 1024) g *= 2;
  return g;
}

// ---- counter-based dropout: keep iff hash(seed, index) >= p * 2^32 (tokens.hip kernels and the Linear epilogues) ----
__device__ __forceinline__ uint32_t mix32(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}
__device__ __forceinline__ float drop_scale(uint64_t seed, uint64_t idx, uint32_t thresh, float inv_keep) {
  return mix32(seed, idx) >= thresh ? inv_keep : 0.f;
}
static inline uint32_t drop_thresh(float p) {
  if (p <= 0.f) return 0u;
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}
// GELU (erf form, nn.GELU default) and its derivative
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float v) {
  const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752f));
  const float pdf = 0.39894228040143268f * expf(-0.5f * v * v);
  return cdf + v * pdf;
}

__device__ __forceinline__ float silu_f(float z) { return z / (1.0f + __expf(-z)); }
// MMI_FAST_SILU (a build-time A/B): the hardware exp2 / reciprocal (v_exp_f32, v_rcp_f32: one ulp each)
// instead of expf and an IEEE division -- ~25 fewer VALU instructions per element in the BatchNorm passes.
#ifndef MMI_FAST_SILU
#define MMI_FAST_SILU 0
#endif
__device__ __forceinline__ float act_fwd(float z, int act) {
  if (act == MMI_ACT_SILU) return MMI_FAST_SILU ? z * __frcp_rn(1.0f + __expf(-z)) : z / (1.0f + expf(-z));
  if (act == MMI_ACT_LEAKY) return z > 0.f ? z : 0.1f * z;
  return z;
}
__device__ __forceinline__ float act_grad(float z, int act) {
  if (act == MMI_ACT_SILU) {
    const float s = MMI_FAST_SILU ? __frcp_rn(1.0f + __expf(-z)) : 1.0f / (1.0f + expf(-z));
    return s * (1.0f + z * (1.0f - s));
  }
  if (act == MMI_ACT_LEAKY) return z > 0.f ? 1.0f : 0.1f;
  return 1.0f;
}
