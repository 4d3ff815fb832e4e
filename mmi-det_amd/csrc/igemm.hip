// fp32-MFMA implicit-GEMM convolution for gfx950: forward, data-gradient, weight-gradient.
//
// Replaces (SURVEY.md §8a rows 4,5,6,9,11,16) the ATen conv2d / linear forward+backward reached from
// models/common.py:114,764,772,333,337,1167-1170,1254,1257 and models/yolo_test.py:44 of the reference.
//
// Design (MI355X-first, not a cuDNN-style port):
//  * activations NHWC, weights OHWI  -> the GEMM K axis (tap, channel) is contiguous in both operands, every global
//    access is a 16-byte lane access over 128-byte row segments;
//  * v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD); 256-thread workgroups = 4 waves in a 2x2 grid, each wave
//    owning (BM/2)x(BN/2) of the BMxBN tile as 32x32 accumulator tiles;
//  * A/B K-slabs of 32 staged through LDS, double buffered, one barrier per K-step; next slab's global loads are issued
//    before the MFMA block so HBM/L2 latency hides behind 4096 cycles of matrix work;
//  * LDS rows padded 32->36 floats so the ds_read_b128 operand fetches are bank-conflict-free; one b128 per lane
//    feeds FOUR k-steps (the K order inside a slab is permuted identically for A and B, which a GEMM sum allows);
//  * XCD-aware tile order: all output-channel tiles of a pixel tile run on one XCD (shared 4 MiB L2);
//  * epilogue fuses bias and the BatchNorm batch-statistics partial sums (no extra pass over y for mean/var).
#include "common.h"

namespace {

#ifndef MMI_BK
#define MMI_BK 32
#endif
constexpr int BK = MMI_BK;            // K-slab depth (32; 64 is an experiment: half the barriers per MFMA, two workgroups per CU)
constexpr int KT = BK / 4;            // loader threads per tile row (one float4 each)
constexpr int RPP = 256 / KT;         // tile rows covered by one pass of the 256 loader threads
#ifndef MMI_IGEMM_STAGES
#define MMI_IGEMM_STAGES 1  // LDS stages of the fwd/dgrad kernel: 1 = single buffer + register prefetch (3 waves/SIMD,
                            // measured +3 % over the double-buffered 2-waves/SIMD form); 2 = double buffer
#endif
constexpr int LDS_PAD = BK + 4;  // floats per [row][k] LDS row
// which MFMA group (0..3) of the current slab issues prefetch load number i of the next slab
#ifndef MMI_LOAD_SPREAD
#define MMI_LOAD_SPREAD(i, g) ((i) % 3 == (g))
#endif

// Invalid lanes of the branch-free tile loaders read this instead of being masked afterwards: no select on the loaded
// value, so hipcc does not have to wait for the load where it is issued (it would: `ok ? v : 0` forces vmcnt(0)).
// (The pointer travels as a kernel argument so that it stays in the global address space: selecting against the
// symbol itself degrades every tile load to a flat_load.)
__device__ f32x4 g_zero4 = {0.f, 0.f, 0.f, 0.f};
#define ZERO_SRC p.zero

const float* zero_src() {
  static const float* ptr[64] = {nullptr};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  if (ptr[dev] == nullptr) {
    void* q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_zero4)) == hipSuccess) ptr[dev] = (const float*)q;
  }
  return ptr[dev];
}

struct IgemmP {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  float* stat_part;
  const float* zero;  // 16 zero bytes in global memory (source of masked lanes)
  int M, Ncol, Kc, KH, KW, P, Q, Hs, Ws, lda, ldc, stride, pad, Ktot, ldb, mtiles, ntiles;
  int par;  // dgrad of a stride-2 conv: blockIdx.y = output-pixel parity class, which only sees its own taps
  const float* res;  // inference epilogue (Model.fuse()): y = act(acc + bias) + res[row * ldr + col]; null = no residual
  int ldr, act;
  float* sk_slots;  // stream-K: 2 partial-tile slots of BM*BN floats per workgroup
  int* sk_count;    // stream-K: per-tile arrival counters (zero before and after every launch)
  // token-side Linear epilogues (mmi_linear_epilogue): MMI_EPI_*
  int epi, ldaux, ldaux_out;
  const float* aux;
  float* aux_out;
  uint64_t seed;
  const uint64_t* seed_dev;
  uint32_t drop_thresh;
  float inv_keep;
  // uniform-tap loaders: byte extents of the A tensor (incl. the margin in front of it) and of the weights; of the output
  // (0: too large for 31-bit offsets, the epilogue keeps its pointer stores)
  uint32_t a_bytes, b_bytes, c_bytes;
  // BatchNorm statistics finished inside this launch (mmi_conv_bn_fwd): the workgroups that arrive last fold the partial
  // rows (stat_arrive, common.h) and write mean / 1/sqrt(var + eps), the running statistics and num_batches_tracked, so no
  // "finalize" launch follows the convolution.  bn_mi == null: the partial rows are all there is (mmi_conv_fwd).
  StatFold bn_fold;
  float* bn_mi;
  float* bn_rmean;
  float* bn_rvar;
  int64_t* bn_nbt;
  int bn_nnbt;
  float bn_eps, bn_momentum;
  double bn_inv_rows, bn_unbias;  // 1 / rows, rows / (rows - 1)
};

// Tap enumeration of the K axis: k = tap * Kc + c, tap = ti * ntw + tj, (kh, kw) = (kh0 + khs*ti, kw0 + kws*tj).
// Generic conv: all KH x KW taps.  Stride-2 dgrad, pixel parity (pa, qa): rows with p even only meet kh = 1, rows with
// p odd meet kh in {0, 2} (same for columns), so each class runs 1, 2, 2 or 4 taps instead of 9 (exact FLOPs, no
// multiply-by-zero work).
struct Taps {
  int kh0, khs, kw0, kws, ntw, Ktot;
};

// Division-free cursor over the K axis: k = tap * Kc + c with tap = ti * ntw + tj; advance() moves k by one slab (BK).
struct KCur {
  int c, tap, ti, tj;
  __device__ __forceinline__ void init(int k, int Kc, int ntw) {
    tap = k / Kc;
    c = k - tap * Kc;
    ti = tap / ntw;
    tj = tap - ti * ntw;
  }
  __device__ __forceinline__ void advance(int Kc, int ntw) {
    c += BK;
    while (c >= Kc) {  // at most once when Kc >= BK (every layer but the 12-channel Focus input)
      c -= Kc;
      ++tap;
      if (++tj == ntw) {
        tj = 0;
        ++ti;
      }
    }
  }
};

// one row of the A tile as seen by a loader thread
struct RowInfo {
  int64_t base;  // source-image pixel base (img * Hs * Ws); -1 -> row out of range
  int ph, qw;    // fwd: p*stride-pad ; dgrad: p+pad
};

template <bool DGRAD>
__device__ __forceinline__ bool src_pixel(const IgemmP& p, const RowInfo& r, int kh, int kw, int64_t& pix) {
  int ih, iw;
  if (!DGRAD) {
    ih = r.ph + kh;
    iw = r.qw + kw;
  } else {
    int th = r.ph - kh, tw = r.qw - kw;
    if (th < 0 || tw < 0) return false;
    if (p.stride == 2) {
      if ((th | tw) & 1) return false;
      ih = th >> 1;
      iw = tw >> 1;
    } else {
      ih = th;
      iw = tw;
    }
  }
  if (r.base < 0 || ih < 0 || iw < 0 || ih >= p.Hs || iw >= p.Ws) return false;
  pix = r.base + (int64_t)ih * p.Ws + iw;
  return true;
}

template <bool DGRAD, bool VEC>
__device__ __forceinline__ f32x4 load_a(const IgemmP& p, const Taps& tp, const RowInfo& r, int k) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (VEC) {
    if (k < tp.Ktot) {
      const int tap = k / p.Kc, c = k - tap * p.Kc;
      const int ti = tap / tp.ntw, tj = tap - ti * tp.ntw;
      const int kh = tp.kh0 + tp.khs * ti, kw = tp.kw0 + tp.kws * tj;
      int64_t pix;
      if (src_pixel<DGRAD>(p, r, kh, kw, pix)) v = *reinterpret_cast<const f32x4*>(p.A + pix * p.lda + c);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ke = k + e;
      if (ke < tp.Ktot) {
        const int tap = ke / p.Kc, c = ke - tap * p.Kc;
        const int ti = tap / tp.ntw, tj = tap - ti * tp.ntw;
        const int kh = tp.kh0 + tp.khs * ti, kw = tp.kw0 + tp.kws * tj;
        int64_t pix;
        if (src_pixel<DGRAD>(p, r, kh, kw, pix)) v[e] = p.A[pix * p.lda + c];
      }
    }
  }
  return v;
}

// fwd weights: B[n][k], k contiguous
template <bool VEC>
__device__ __forceinline__ f32x4 load_b_nk(const IgemmP& p, int n, int k) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (n < p.Ncol) {
    const float* src = p.B + (int64_t)n * p.ldb + k;
    if (VEC) {
      if (k < p.Ktot) v = *reinterpret_cast<const f32x4*>(src);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k + e < p.Ktot) v[e] = src[e];
    }
  }
  return v;
}

// dgrad weights: B[k=(tap,co)][n=ci] = W[co][tap][ci], n contiguous
template <bool VEC>
__device__ __forceinline__ f32x4 load_b_kn(const IgemmP& p, const Taps& tp, int k, int n) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (k < tp.Ktot) {
    const int t = k / p.Kc, co = k - t * p.Kc;
    const int ti = t / tp.ntw, tj = t - ti * tp.ntw;
    const int tap = (tp.kh0 + tp.khs * ti) * p.KW + tp.kw0 + tp.kws * tj;
    const float* src = p.B + (int64_t)co * p.ldb + (int64_t)tap * p.Ncol + n;
    if (VEC) {
      if (n < p.Ncol) v = *reinterpret_cast<const f32x4*>(src);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < p.Ncol) v[e] = src[e];
    }
  }
  return v;
}

// Stream-K schedule (SK): the grid is exactly the number of resident workgroup slots (CUs x occupancy) and every
// workgroup runs the same number (+-1) of K-slab iterations of the tile-major iteration space [tiles x nk), so a grid of
// 800 equal tiles no longer costs 4 "layers" of 256 on a chip that holds 768 (measured: 0.335 ms vs 0.262 ms for 768).
// A workgroup's range is a tail of one tile, whole tiles, and a head of another; partial accumulators go to a workspace
// slot, a per-tile arrival counter elects the last contributor, which sums the parts in K order (deterministic) and runs
// the normal epilogue.  Nobody waits on anybody, so residency is a performance assumption, not a correctness one.
struct SkRange {
  int q, r;  // every workgroup owns q iterations, the first r own one more
  __device__ __forceinline__ int start(int b) const { return b * q + (b < r ? b : r); }
  __device__ __forceinline__ int owner(int x) const {
    const int edge = r * (q + 1);
    return x < edge ? x / (q + 1) : r + (x - edge) / q;
  }
};

// PREC = 0: exact fp32 products (v_mfma_f32_32x32x2_f32).  PREC = 1 (opt-in, forward-layout operands only): every fp32
// operand is split into two bf16 terms when it is staged into LDS, x = hi + lo with |x - hi - lo| <= 2^-17 |x|, and a product
// is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 3 instructions of 8 passes per 16 k instead
// of 8 instructions of 16 passes, relative error of a product <= 2^-16 (the dropped lo*lo term is 2^-18).
// PREC = 2: three bf16 terms per operand (x = t0 + t1 + t2, residual <= 2^-25 |x|) and the six products of total order <= 2
// (t0*t0, t0*t1, t1*t0, t0*t2, t2*t0, t1*t1): dropped terms <= 2^-24 per product, at 6 x 8 passes per 16 k.
// PREC = 3: the same three terms -- which represent a 24-bit significand exactly -- and all nine products, each exact in
// fp32: every fp32 product is formed exactly, as by the fp32 MFMA; only the order of the fp32 accumulation differs.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
// eight consecutive k (rows p, p+1, ... p+7 of a k-major bf16 image) of this lane's column via two ds_read_b64_tr_b16
__device__ __forceinline__ bf16x8 tr_read8(const char* p, int row_bytes) {
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p));
  const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p + 4 * row_bytes));
  const s16x8 v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// x -> NP bf16 terms, each the rounded residual of the previous ones
template <int NP>
__device__ __forceinline__ void split_bf16(const f32x4& v, bf16x4 (&t)[NP]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float r = v[i];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      t[k][i] = (__bf16)r;
      r -= (float)t[k][i];
    }
  }
}

// EPI: the token-side Linear epilogues (p.epi) are compiled in; a separate instantiation, because their registers
// (64-bit hash, erf) would otherwise cost the convolution kernels occupancy.
// UNI (uniform-tap loaders): when the channel count is a multiple of the slab depth, every thread of the workgroup is in
// the same filter tap during a slab, so the tap's pixel displacement and the channel offset are one SCALAR; a thread's part
// of an address (its rows, its k lane) is computed once per tile, tap validity is one bit per (row, tap), and the loads are
// buffer loads (SGPR resource + per-lane offset + scalar offset) whose out-of-range lanes return zero.  That leaves about
// 10 VALU instructions per K slab instead of 90-160: tools/mfma_mix.hip shows that VALU instructions issued next to an MFMA
// stream cost MFMA throughput at three waves per SIMD (1 per MFMA: 87 % of peak, 2: 80 %, LDS reads: nothing), which is
// exactly where the cursor-based loaders (1.4-2.9 VALU per MFMA, a third of them 64-bit) had left these kernels.
#ifndef MMI_UNI_OCC
#define MMI_UNI_OCC 3
#endif
// W41 (narrow outputs: Focus' input gradient has N = 12): the four waves are stacked along M, each owning 32 rows x the whole
// tile width, and skip the 32-column blocks beyond the last output column -- in the 2 x 2 layout half of the waves would own
// nothing but padding and leave their SIMDs' matrix pipes idle.
template <int BM, int BN, bool DGRAD, bool VEC, bool SK, int PREC = 0, bool EPI = false, bool UNI = false, bool W41 = false>
__global__ __launch_bounds__(256, (MMI_IGEMM_STAGES == 1 && BK == 32 && (PREC < 2 || PREC >= 4)) ? (UNI ? MMI_UNI_OCC : 3) : 2) void igemm_kernel(IgemmP p) {
  static_assert(!W41 || (DGRAD && !SK && PREC == 0 && !EPI && BM == 128), "the stacked wave layout exists for the plain fp32 dgrad tiles");
  static_assert(PREC == 0 || (VEC && BK == 32), "the split-bf16 forms exist for the vector loaders only");
  static_assert(!UNI || VEC, "uniform-tap loaders are a form of the vector loaders");
  // PREC = 4 (bf16 STORAGE): the activation operand A and the output C live in HBM as bf16 (the weights stay
  // fp32 master copies, rounded when a tile is staged), one bf16 MFMA product per element pair, fp32 accumulation, fp32
  // BatchNorm statistics taken from the accumulators.  Same tile machinery as the split forms with a single plane.
  constexpr bool BF = PREC == 4;
  static_assert(!BF || !UNI, "bf16 storage uses the cursor loaders");
  // PREC = 5 ("bf16x1"): fp32 operands in HBM, each rounded to ONE bf16 term when staged, one bf16 MFMA product: the arithmetic of
  // the bf16-storage mode for the GEMMs whose operands stay fp32 (the token-side Linear layers, Focus, Detect)
  constexpr bool ONE = BF || PREC == 5;
  constexpr int NP = PREC == 0 || ONE ? 1 : (PREC == 3 ? 3 : PREC + 1);      // bf16 planes per operand
  constexpr int OL = ONE ? 0 : (PREC == 3 ? 2 * (NP - 1) : NP - 1);          // highest total order of the products kept
  // floats per [row][k] LDS record: fp32 32 + 4 pad; split forms NP x 64 B of bf16 + 16 B pad (20, 36 or 52 floats: each makes
  // the ds_read_b128 of 8 consecutive rows hit 8 different 16-byte bank groups)
  constexpr int RSF = ONE ? 20 : (PREC >= 2 ? 52 : LDS_PAD);
  constexpr int WM = W41 ? BM / 4 : BM / 2, WN = W41 ? BN : BN / 2, TM = WM / 32, TN = WN / 32;
  constexpr int RA = BM / RPP;                      // A rows per loader thread
  constexpr int A_ELEMS = BM * RSF;
  // split-bf16 dgrad: the weight tile stays k-major ([k][n], as it comes from OHWI memory) in two bf16 planes whose rows are
  // padded by 64 B (conflict-free ds_read_b64_tr_b16: the MFMA B operand is fetched with the hardware transpose read)
  constexpr int B_RSB = BN * 2 + 64;                                  // bytes per k row of one plane
  constexpr int B_ELEMS = DGRAD ? (PREC >= 1 ? NP * BK * B_RSB / 4 : BK * BN) : BN * RSF;
  constexpr int STAGE = A_ELEMS + B_ELEMS;
  constexpr int RB = BN / RPP;                      // fwd: B rows per loader thread
  constexpr int VPR = BN / 4, RPI = 256 / VPR, KB_IT = BK / RPI;  // dgrad B loader geometry
  __shared__ __align__(16) float smem[MMI_IGEMM_STAGES * STAGE];
  __shared__ int rowmap[BM];  // parity mode: tile row -> output pixel
  __shared__ int sk_last;
  __shared__ int bn_flag;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = W41 ? wave : wave >> 1, wn = W41 ? 0 : wave & 1;
  Taps tp{0, 1, 0, 1, p.KW, p.Ktot};
  int pa = 0, qa = 0, Pc = p.P, Qc = p.Q, Mc = p.M, ntile_tot = p.mtiles * p.ntiles;
  if (!SK && DGRAD && p.par) {  // uniform per workgroup
    pa = blockIdx.y >> 1;
    qa = blockIdx.y & 1;
    Pc = (p.P - pa + 1) >> 1;
    Qc = (p.Q - qa + 1) >> 1;
    Mc = (p.M / (p.P * p.Q)) * Pc * Qc;
    tp = Taps{pa ? 0 : 1, pa ? 2 : 0, qa ? 0 : 1, qa ? 2 : 0, qa ? 2 : 1, (pa ? 2 : 1) * (qa ? 2 : 1) * p.Kc};
    ntile_tot = ((Mc + BM - 1) / BM) * p.ntiles;
    if ((int)blockIdx.x >= ntile_tot) return;
  }
  const bool par = !SK && DGRAD && p.par;
  const int nk = (tp.Ktot + BK - 1) / BK;
  const int ntaps = tp.Ktot / p.Kc;
  const int kq = (t % KT) * 4;  // this thread's k offset inside a slab ([row][k] tiles)
  const int lrow = t / KT;      // 0..RPP-1
  const int l31 = lane & 31, lh = lane >> 5;
  __amdgpu_buffer_rsrc_t srd_a, srd_b;
  if constexpr (UNI) {
    const int64_t margin = ((int64_t)p.KH * p.Ws + p.KW) * p.lda;  // floats in front of A that row offsets may reach into
    srd_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A - margin), 0, (int)p.a_bytes, 0x00020000);
    srd_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)p.b_bytes, 0x00020000);
  }

  // iteration range of this workgroup: data-parallel = the nk slabs of one tile; stream-K = an even share of everything
  SkRange sk{0, 0};
  int bid = 0, it = 0, it_end = nk;
  if (SK) {
    bid = xcd_remap(blockIdx.x, gridDim.x);
    const int total = ntile_tot * nk;
    sk.q = total / (int)gridDim.x;
    sk.r = total - sk.q * (int)gridDim.x;
    it = sk.start(bid);
    it_end = it + sk.q + (bid < sk.r ? 1 : 0);
  }
  const int it_begin = it;

  while (it < it_end) {
    int tile, ks0, ks1;
    if (SK) {
      tile = it / nk;
      ks0 = it - tile * nk;
      ks1 = min(nk, ks0 + (it_end - it));
    } else {
      tile = xcd_remap(blockIdx.x, ntile_tot);
      ks0 = 0;
      ks1 = nk;
    }
    const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    // a 1x1 stride-1 layer (C3's cv1/cv2/cv3, every Linear): source pixel = output pixel, no taps to test, no divisions
    const bool lin1 = UNI && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0;  // uniform
    RowInfo rows[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int m = m0 + lrow + RPP * i;
      int orow = -1;
      if (lin1) {
        rows[i].base = m < Mc ? 0 : -1;
        rows[i].ph = rows[i].qw = 0;
      } else if (m < Mc) {
        const int pq = Pc * Qc;
        const int img = m / pq, rem = m - img * pq;
        int pp = rem / Qc, qq = rem - pp * Qc;
        if (par) {
          pp = 2 * pp + pa;
          qq = 2 * qq + qa;
          orow = (img * p.P + pp) * p.Q + qq;
        }
        rows[i].base = (int64_t)img * p.Hs * p.Ws;
        rows[i].ph = DGRAD ? pp + p.pad : pp * p.stride - p.pad;
        rows[i].qw = DGRAD ? qq + p.pad : qq * p.stride - p.pad;
      } else {
        rows[i].base = -1;
        rows[i].ph = rows[i].qw = 0;
      }
      if (par && (t % KT) == 0) rowmap[lrow + RPP * i] = orow;  // visible after the K loop's barriers
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[RA];
    bf16x4 rab[BF ? RA : 1];  // bf16 storage: the A operand arrives as 4 bf16 per load
    constexpr int NB = DGRAD ? KB_IT : RB;
    f32x4 rb[NB];

    // ---- UNI: per-thread address parts and tap-validity bits of this tile (see the kernel's header comment) ----
    // Source position of row r under tap (ti, tj):  ih = ihb[r] + sgn * dh * ti,  iw = iwb[r] + sgn * dw * tj  with
    // forward: sgn = +1, (dh, dw) = (khs, kws), ihb = p*stride - pad + kh0;   dgrad (stride 1, or one parity class of a
    // stride-2 layer): sgn = -1, ihb = (p + pad - kh0) >> sh, (dh, dw) = (khs, kws) >> sh.  Offsets are taken from the lowest
    // position any tap reaches, shifted by a margin of KH rows + KW pixels so that they are never negative.
    constexpr uint32_t OOB = 0x80000000u;  // >= num_records (host checks that every tensor is below 2 GiB)
    uint32_t aoff[UNI ? RA : 1], amask[UNI ? RA : 1], boff[UNI ? NB : 1];
    const int u_sh = (DGRAD && p.stride == 2) ? 1 : 0;
    const int u_dh = tp.khs >> u_sh, u_dw = tp.kws >> u_sh;
    const int u_nth = ntaps / tp.ntw;
    if constexpr (UNI) {
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        aoff[i] = OOB;
        amask[i] = 0xFFFFFFFFu;
        if (lin1) {
          if (rows[i].base >= 0) {
            aoff[i] = (uint32_t)((((int64_t)(m0 + lrow + RPP * i) + p.Ws + 1) * p.lda + kq) * 4);  // margin = KH*Ws + KW pixels
            amask[i] = 0u;
          }
        } else if (rows[i].base >= 0) {
          const int ihb = DGRAD ? ((rows[i].ph - tp.kh0) >> u_sh) : rows[i].ph + tp.kh0;
          const int iwb = DGRAD ? ((rows[i].qw - tp.kw0) >> u_sh) : rows[i].qw + tp.kw0;
          const int ihlo = DGRAD ? ihb - u_dh * (u_nth - 1) : ihb, iwlo = DGRAD ? iwb - u_dw * (tp.ntw - 1) : iwb;
          const int64_t pix = rows[i].base + (int64_t)(ihlo + p.KH) * p.Ws + iwlo + p.KW;
          aoff[i] = (uint32_t)((pix * p.lda + kq) * 4);
          // separable: a tap is out if its row is out or its column is out
          uint32_t bw = 0, bad = 0;
          for (int tj = 0; tj < tp.ntw; ++tj)
            bw |= ((unsigned)(DGRAD ? iwb - u_dw * tj : iwb + u_dw * tj) >= (unsigned)p.Ws ? 1u : 0u) << tj;
          const uint32_t roww = (1u << tp.ntw) - 1u;
          for (int ti = 0; ti < u_nth; ++ti)
            bad |= ((unsigned)(DGRAD ? ihb - u_dh * ti : ihb + u_dh * ti) >= (unsigned)p.Hs ? roww : bw) << (ti * tp.ntw);
          amask[i] = bad;
        }
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        if (!DGRAD) {
          const int n = n0 + lrow + RPP * i;
          boff[i] = n < p.Ncol ? (uint32_t)(((int64_t)n * p.ldb + kq) * 4) : OOB;
        } else {
          const int n = n0 + (t % VPR) * 4;
          boff[i] = n < p.Ncol ? (uint32_t)(((int64_t)(t / VPR + RPI * i) * p.ldb + n) * 4) : OOB;
        }
      }
    }
    // the slab's scalars: channel offset inside the tap, tap coordinates (clamped to the last slab: see advance())
    int u_c0 = 0, u_ti = 0, u_tj = 0;

    // Vector path: division-free K cursors (advanced by one slab per step) and branch-free loads (an invalid lane reads
    // the zero source), so the loads of the NEXT slab can be issued piecewise between the MFMA groups of the current one
    // and their address arithmetic runs in the MFMA shadow.
    KCur ca, cb[NB];
    int k0cur = ks0 * BK;
    if constexpr (UNI) {
      const int tap0 = k0cur / p.Kc;
      u_c0 = k0cur - tap0 * p.Kc;
      u_ti = tap0 / tp.ntw;
      u_tj = tap0 - u_ti * tp.ntw;
    }
    if (VEC && !UNI) {
      ca.init(k0cur + kq, p.Kc, tp.ntw);
      if (DGRAD) {
#pragma unroll
        for (int i = 0; i < NB; ++i) cb[i].init(k0cur + t / VPR + RPI * i, p.Kc, tp.ntw);
      }
    }
    auto load_a_row = [&](int i) {
      if (!VEC) {
        ra[i] = load_a<DGRAD, VEC>(p, tp, rows[i], k0cur + kq);
        return;
      }
      if constexpr (UNI) {
        const int tap = u_ti * tp.ntw + u_tj;
        const int dpix = DGRAD ? (u_nth - 1 - u_ti) * u_dh * p.Ws + (tp.ntw - 1 - u_tj) * u_dw : u_ti * u_dh * p.Ws + u_tj * u_dw;
        const uint32_t soff = (uint32_t)(dpix * p.lda + u_c0) * 4u;
        const uint32_t inv = (uint32_t)__builtin_amdgcn_sbfe(amask[i], tap, 1);   // -1 where this tap leaves the image
        ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_a, aoff[i] | (inv & OOB), soff, 0));
        return;
      }
      int64_t pix = 0;
      const bool ok = (ca.tap < ntaps) & src_pixel<DGRAD>(p, rows[i], tp.kh0 + tp.khs * ca.ti, tp.kw0 + tp.kws * ca.tj, pix);
      if constexpr (BF)
        rab[i] = *reinterpret_cast<const bf16x4*>(ok ? reinterpret_cast<const char*>(p.A) + (pix * p.lda + ca.c) * 2
                                                     : reinterpret_cast<const char*>(ZERO_SRC));
      else
        ra[i] = *reinterpret_cast<const f32x4*>(ok ? p.A + pix * p.lda + ca.c : ZERO_SRC);
    };
    auto load_b_row = [&](int i) {
      if (!DGRAD) {
        if (!VEC) {
          rb[i] = load_b_nk<VEC>(p, n0 + lrow + RPP * i, k0cur + kq);
          return;
        }
        if constexpr (UNI) {
          const int tapw = (tp.kh0 + tp.khs * u_ti) * p.KW + tp.kw0 + tp.kws * u_tj;   // (forward: all taps, so tapw = tap)
          rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_b, boff[i], (uint32_t)(tapw * p.Kc + u_c0) * 4u, 0));
          return;
        }
        const int n = n0 + lrow + RPP * i, k = k0cur + kq;
        const bool ok = (n < p.Ncol) & (k < tp.Ktot);
        rb[i] = *reinterpret_cast<const f32x4*>(ok ? p.B + (int64_t)n * p.ldb + k : ZERO_SRC);
      } else {
        if (!VEC) {
          rb[i] = load_b_kn<VEC>(p, tp, k0cur + t / VPR + RPI * i, n0 + (t % VPR) * 4);
          return;
        }
        if constexpr (UNI) {
          const int tapw = (tp.kh0 + tp.khs * u_ti) * p.KW + tp.kw0 + tp.kws * u_tj;
          rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_b, boff[i], (uint32_t)(u_c0 * p.ldb + tapw * p.Ncol) * 4u, 0));
          return;
        }
        const int n = n0 + (t % VPR) * 4;
        const bool ok = (cb[i].tap < ntaps) & (n < p.Ncol);
        const int tapw = (tp.kh0 + tp.khs * cb[i].ti) * p.KW + tp.kw0 + tp.kws * cb[i].tj;
        rb[i] = *reinterpret_cast<const f32x4*>(ok ? p.B + (int64_t)cb[i].c * p.ldb + (int64_t)tapw * p.Ncol + n : ZERO_SRC);
      }
    };
    auto advance = [&]() {  // move every cursor to the next slab
      k0cur += BK;
      if constexpr (UNI) {
        // Past the last slab the cursor stays where it is: the surplus prefetch of the last iteration then re-reads the last
        // slab (a scalar offset beyond the tap table would leave the buffer's range check, which covers the lane offset).
        if (k0cur < tp.Ktot) {
          u_c0 += BK;
          if (u_c0 >= p.Kc) {
            u_c0 = 0;
            if (++u_tj == tp.ntw) {
              u_tj = 0;
              ++u_ti;
            }
          }
        }
      }
      if (VEC && !UNI) {
        ca.advance(p.Kc, tp.ntw);
        if (DGRAD) {
#pragma unroll
          for (int i = 0; i < NB; ++i) cb[i].advance(p.Kc, tp.ntw);
        }
      }
    };
    auto gload = [&]() {
#pragma unroll
      for (int i = 0; i < RA; ++i) load_a_row(i);
#pragma unroll
      for (int i = 0; i < NB; ++i) load_b_row(i);
    };
    auto lstore = [&](int stage) {
      float* As = smem + stage * STAGE;
      float* Bs = As + A_ELEMS;
      if constexpr (PREC >= 1) {
        // row record: NP planes of 32 bf16 (64 B each) | 16 B pad
#pragma unroll
        for (int i = 0; i < RA; ++i) {
          bf16x4 tm[NP];
          if constexpr (BF) tm[0] = rab[i];
          else split_bf16<NP>(ra[i], tm);
          __bf16* row = reinterpret_cast<__bf16*>(As + (lrow + RPP * i) * RSF);
#pragma unroll
          for (int k = 0; k < NP; ++k) *reinterpret_cast<bf16x4*>(row + 32 * k + kq) = tm[k];
        }
        if constexpr (!DGRAD) {
#pragma unroll
          for (int i = 0; i < RB; ++i) {
            bf16x4 tm[NP];
            split_bf16<NP>(rb[i], tm);
            __bf16* row = reinterpret_cast<__bf16*>(Bs + (lrow + RPP * i) * RSF);
#pragma unroll
            for (int k = 0; k < NP; ++k) *reinterpret_cast<bf16x4*>(row + 32 * k + kq) = tm[k];
          }
        } else {
          char* base = reinterpret_cast<char*>(Bs);
#pragma unroll
          for (int i = 0; i < KB_IT; ++i) {
            bf16x4 tm[NP];
            split_bf16<NP>(rb[i], tm);
            char* dst = base + (t / VPR + RPI * i) * B_RSB + (t % VPR) * 8;
#pragma unroll
            for (int k = 0; k < NP; ++k) *reinterpret_cast<bf16x4*>(dst + k * BK * B_RSB) = tm[k];
          }
        }
        return;
      }
#pragma unroll
      for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4*>(As + (lrow + RPP * i) * LDS_PAD + kq) = ra[i];
      if (!DGRAD) {
#pragma unroll
        for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4*>(Bs + (lrow + RPP * i) * LDS_PAD + kq) = rb[i];
      } else {
#pragma unroll
        for (int i = 0; i < KB_IT; ++i) *reinterpret_cast<f32x4*>(Bs + (t / VPR + RPI * i) * BN + (t % VPR) * 4) = rb[i];
      }
    };

    gload();
    lstore(0);
    __syncthreads();

    for (int ks = ks0; ks < ks1; ++ks) {
      // The next slab is fetched unconditionally (straight-line code, counted waits): past the end of K every lane is
      // masked to the zero source, so the last iteration only stages zeros (or, in a stream-K segment that stops short
      // of the tile's end, an unused slab) into the idle buffer.
      advance();
      const float* As = smem + (MMI_IGEMM_STAGES == 2 ? ((ks - ks0) & 1) : 0) * STAGE;
      const float* Bs = As + A_ELEMS;
      if constexpr (PREC >= 1) {
#pragma unroll
        for (int kb = 0; kb < BK / 16; ++kb) {
          // the next slab's global loads: two thirds ahead of the first 16-k block, the rest ahead of the second
#pragma unroll
          for (int i = 0; i < RA; ++i)
            if (kb == 0 ? (i % 3 != 2) : (i % 3 == 2)) load_a_row(i);
#pragma unroll
          for (int i = 0; i < NB; ++i)
            if (kb == 0 ? ((RA + i) % 3 != 2) : ((RA + i) % 3 == 2)) load_b_row(i);
          bf16x8 af[NP][TM], bf[NP][TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const float* row = As + (wm * WM + i * 32 + l31) * RSF + kb * 8 + lh * 4;   // float index = byte offset / 4
#pragma unroll
            for (int k = 0; k < NP; ++k) af[k][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(row + 16 * k));
          }
          if constexpr (!DGRAD) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              const float* row = Bs + (wn * WN + j * 32 + l31) * RSF + kb * 8 + lh * 4;
#pragma unroll
              for (int k = 0; k < NP; ++k) bf[k][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(row + 16 * k));
            }
          } else {
            // transposed read: per 16-lane group a block of 4 k-rows x 16 columns; lane 4q+p supplies row q, columns 4p..4p+3
            // and receives column (lane % 16) of the four rows; two reads = the 8 consecutive k of this lane's column
            const int q = (lane & 15) >> 2, pp = lane & 3, m0 = ((lane >> 4) & 1) * 16;
            const char* base = reinterpret_cast<const char*>(Bs) + (kb * 16 + lh * 8 + q) * B_RSB + (wn * WN + m0 + 4 * pp) * 2;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int k = 0; k < NP; ++k) bf[k][j] = tr_read8(base + j * 64 + k * BK * B_RSB, B_RSB);
          }
          // products of total order <= NP-1, smallest terms first
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int o = OL; o >= 0; --o)
#pragma unroll
                for (int ka = (o > NP - 1 ? o - (NP - 1) : 0); ka <= (o < NP - 1 ? o : NP - 1); ++ka)
                  acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ka][i], bf[o - ka][j], acc[i][j], 0, 0, 0);
        }
      } else {
#pragma unroll
      for (int g = 0; g < BK / 8; ++g) {
        // a third of the next slab's global loads ahead of each of the first three MFMA groups (the fourth group's
        // 1024 MFMA cycles then cover the tail of the load latency before the LDS stores below)
#pragma unroll
        for (int i = 0; i < RA; ++i)
          if (MMI_LOAD_SPREAD(i, g)) load_a_row(i);
#pragma unroll
        for (int i = 0; i < NB; ++i)
          if (MMI_LOAD_SPREAD(RA + i, g)) load_b_row(i);
        // a wave whose 32-column blocks all lie beyond the last output column (Focus' input gradient: N = 12 in a 64-wide
        // tile) has nothing to multiply: it still loads and synchronises, but leaves the matrix pipe to the others
        if constexpr (DGRAD && BN == 64 && !SK && !W41) {      // (only where it occurs: elsewhere the branch costs registers)
          if (n0 + wn * WN >= p.Ncol) continue;
        }
        f32x4 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          a[i] = *reinterpret_cast<const f32x4*>(As + (wm * WM + i * 32 + l31) * LDS_PAD + g * 8 + lh * 4);
        if (!DGRAD) {
#pragma unroll
          for (int j = 0; j < TN; ++j)
            b[j] = *reinterpret_cast<const f32x4*>(Bs + (wn * WN + j * 32 + l31) * LDS_PAD + g * 8 + lh * 4);
        } else {
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) b[j][e] = Bs[(g * 8 + lh * 4 + e) * BN + wn * WN + j * 32 + l31];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              if constexpr (W41) {
                if (j > 0 && n0 + j * 32 >= p.Ncol) continue;      // (uniform: a column block of pure padding)
              }
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
            }
      }
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the LDS stores (and their vmcnt waits) behind every MFMA of the slab
      if (MMI_IGEMM_STAGES == 1) __syncthreads();  // single LDS stage: everyone is done reading before it is overwritten
      lstore(MMI_IGEMM_STAGES == 2 ? ((ks - ks0 + 1) & 1) : 0);
      __syncthreads();
    }
    it += ks1 - ks0;

    if (SK && ks1 - ks0 < nk) {
      // ---- partial tile: publish, count arrivals, the last contributor folds every part in K order ----
      // Partials travel with device-scope (sc1) stores and loads: they are coherent across the eight XCD L2s by
      // themselves, so no agent-scope fence is needed (one would write back and invalidate the whole L2 per segment,
      // which costs far more than the schedule saves).
      constexpr int SLOT = BM * BN;
      float* mine = p.sk_slots + (int64_t)(2 * bid + (it - (ks1 - ks0) != it_begin ? 1 : 0)) * SLOT;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            __hip_atomic_store(mine + ((i * TN + j) * 16 + r) * 256 + t, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      mmi_drain_stores();  // every storing wave: its sc1 (write-through) stores have left the CU before the barrier below
      __syncthreads();
      const int lo = tile * nk;
      const int b_first = sk.owner(lo), b_last = sk.owner(lo + nk - 1);
      if (t == 0) {
        const int old = __hip_atomic_fetch_add(p.sk_count + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == b_last - b_first;
        if (last) __hip_atomic_store(p.sk_count + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero for the next launch
        sk_last = last;
      }
      __syncthreads();
      if (!sk_last) continue;  // uniform
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      for (int bb = b_first; bb <= b_last; ++bb) {
        const float* part = p.sk_slots + (int64_t)(2 * bb + (sk.start(bb) < lo ? 1 : 0)) * SLOT;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              acc[i][j][r] += __hip_atomic_load(part + ((i * TN + j) * 16 + r) * 256 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }

    // ---- epilogue: C/D layout of 32x32 tiles: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
    const uint64_t epi_seed = EPI ? p.seed + (p.seed_dev != nullptr ? p.seed_dev[0] : 0ull) : 0ull;
    // Interior tiles of the plain training epilogue (bias / BN statistics only): every store is a buffer store whose row
    // displacement is a scalar, so an element costs its statistics (add, fma) and nothing else -- the epilogue runs beside
    // other workgroups' MFMA streams, where VALU instructions are not free (1x1 layers: 4 K slabs per tile).
    const bool fast_store = UNI && !EPI && !par && p.c_bytes != 0 && p.act == MMI_ACT_NONE && p.res == nullptr &&
                            m0 + BM <= Mc && n0 + BN <= p.Ncol;  // uniform
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * WN + j * 32 + l31;
      const bool cok = col < p.Ncol;
      const float bv = (p.bias != nullptr && cok) ? p.bias[col] : 0.f;
      float s1 = 0.f, s2 = 0.f;
      if (UNI && fast_store) {
        const __amdgpu_buffer_rsrc_t srd_c = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (int)p.c_bytes, 0x00020000);
        const uint32_t voff = (uint32_t)(((m0 + wm * WM + 4 * lh) * p.ldc + col) * 4);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[i][j][r] + bv;
            s1 += v;
            s2 = __builtin_fmaf(v, v, s2);  // (explicit, so that both epilogue forms round alike)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), srd_c, voff,
                                                  (uint32_t)((i * 32 + (r & 3) + 8 * (r >> 2)) * p.ldc) * 4u, 0);
          }
        }
      } else
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lr = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int row = m0 + lr;
          float v = acc[i][j][r] + bv;
          s1 += v;
          s2 = __builtin_fmaf(v, v, s2);
          if (!DGRAD && p.act != MMI_ACT_NONE) v = act_fwd(v, p.act);  // uniform; training never sets it (BN follows)
          if (!DGRAD && p.res != nullptr && cok && row < Mc) v += p.res[(int64_t)row * p.ldr + col];
          if (EPI && p.epi != MMI_EPI_NONE && cok && row < Mc) {  // uniform switch; 1x1 only, so `row` is the output row
            const int64_t ao = (int64_t)row * p.ldaux + col;
            if (p.epi == MMI_EPI_DROPOUT_RESIDUAL) {
              if (p.drop_thresh) v *= drop_scale(epi_seed, (uint64_t)((int64_t)row * p.Ncol + col), p.drop_thresh, p.inv_keep);
              v += p.aux[ao];
            } else if (p.epi == MMI_EPI_GELU) {
              p.aux_out[(int64_t)row * p.ldaux_out + col] = v;
              v = gelu_f(v);
            } else if (p.epi == MMI_EPI_GELU_GRAD) {
              v *= gelu_grad_f(p.aux[ao]);
            } else if (p.epi == MMI_EPI_ACCUMULATE) {
              v += BF ? (float)reinterpret_cast<const __bf16*>(p.aux)[ao] : p.aux[ao];
            }
          }
          if (cok && row < Mc) {
            if constexpr (BF) reinterpret_cast<__bf16*>(p.C)[(int64_t)(par ? rowmap[lr] : row) * p.ldc + col] = (__bf16)v;
            else p.C[(int64_t)(par ? rowmap[lr] : row) * p.ldc + col] = v;
          }
        }
      }
      if (p.stat_part != nullptr) {  // uniform branch; rows >= M hold exact zeros (zero A rows, no bias with BN)
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        float* red = smem;  // [2 stats][2 wm][BN]; safe: the K loop ended with a barrier
        if (lh == 0) {
          red[(0 * 2 + wm) * BN + wn * WN + j * 32 + l31] = s1;
          red[(1 * 2 + wm) * BN + wn * WN + j * 32 + l31] = s2;
        }
      }
    }
    if (p.stat_part != nullptr) {
      __syncthreads();
      for (int idx = t; idx < 2 * BN; idx += 256) {
        const int s = idx / BN, c = idx - s * BN;
        const int col = n0 + c;
        if (col < p.Ncol) st_agent(p.stat_part + ((int64_t)mt * 2 + s) * p.Ncol + col, smem[(s * 2 + 0) * BN + c] + smem[(s * 2 + 1) * BN + c]);
      }
      if constexpr (!DGRAD) {
        if (p.bn_mi != nullptr) {  // uniform
          __syncthreads();         // smem[0, 4*BN) has been consumed; the fold reuses it
          double s1, s2;
          if (stat_arrive<BN>(p.bn_fold, mt, nt, n0, reinterpret_cast<double*>(smem), &bn_flag, s1, s2)) {
            const int col = n0 + t;
            if (t < BN && col < p.Ncol) {
              // as mmi_bn_finalize, except that the reciprocals come from the host and the square root is taken in fp32
              // (fp64 division / sqrt are long software sequences whose registers this kernel cannot spare)
              const double mean = s1 * p.bn_inv_rows;
              double var = s2 * p.bn_inv_rows - mean * mean;  // biased (normalisation) variance
              if (var < 0.0) var = 0.0;
              p.bn_mi[col] = (float)mean;
              p.bn_mi[p.Ncol + col] = 1.0f / sqrtf((float)(var + (double)p.bn_eps));
              if (p.bn_rmean != nullptr) {
                p.bn_rmean[col] = (float)((1.0 - p.bn_momentum) * (double)p.bn_rmean[col] + p.bn_momentum * mean);
                p.bn_rvar[col] = (float)((1.0 - p.bn_momentum) * (double)p.bn_rvar[col] + p.bn_momentum * (var * p.bn_unbias));
              }
            }
            if (nt == 0 && t < p.bn_nnbt) p.bn_nbt[t] += 1;
          }
        }
      }
      if (SK) __syncthreads();  // the next segment's prologue overwrites smem
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// weight gradient: C[co][(tap,ci)] = sum_pix dy[pix][co] * x[gather(pix,tap)][ci]; both operands are K(pixel)-strided,
// tiles live in LDS as [k][m] / [k][n] and the MFMA operands are conflict-free ds_read_b32.
struct WgradP {
  const float* DY;
  const float* X;
  float* OUT;   // dw, or slab base when splits > 1
  float* OUTB;  // bias gradient (column sums of dy) of split 0, or null; split z writes OUTB + z * slab_stride
  const float* zero;
  int Mpix, Cout, Cin, KH, KW, Ho, Wo, H, W, stride, pad, ldx, ldy, Ntot, chunk, mtiles, ntiles, splits;
  int64_t slab_stride;
  uint32_t x_bytes;  // TAB loaders: byte extent of x including the margin in front of it
  // split-K fold inside the launch: per-tile arrival counters (zero before and after); the workgroup that completes a
  // tile's last split sums the splits' partial tiles in split order (deterministic) into DW (and DB): no reduce launch
  int* cnt;
  int cnt_per_tile;  // counters of one tile: sum over the tree's levels of ceil(nodes / 4)
  float* DW;
  float* DB;
  // TAB loaders: the per-pixel {source offset, invalid-tap mask} table of the layer's geometry, precomputed once
  // (mmi_conv_wgrad_table_build: it depends on shapes and strides only, not on data); null = built in the kernel, slab by slab
  const uint2* tab;
};

// LDS stages of the wgrad kernel: single-buffered (3+ workgroups per CU, +3..10 % measured on the 3x3 layers) except for
// the 64x64 tile of the tall-skinny 1x1 layers, whose short MFMA phase cannot hide a second barrier per slab.
#ifndef MMI_WGRAD_OCC
#define MMI_WGRAD_OCC 3
#endif
#define MMI_WGRAD_STAGES ((BM == 64 && BN == 64) ? 2 : 1)
// MMI_WGRAD_LDS_B32 = 1: one ds_read_b32 with a 16-bit immediate per MFMA fragment instead of the compiler's ds_read2_b32 pairs +
// a v_add_u32 per pair: 29 -> 5 VALU instructions per K slab, +2..8 % stand-alone on every shape (3x3 128->128: 112.1 -> 114.1
// TFLOP/s) -- and 1.0 ms SLOWER inside the step (122.75 vs 121.7 ms, three interleaved pairs, profiles/r02_ab_wgrad_lds_b32.txt):
// twice the LDS instructions, and in the step wgrad shares every CU's LDS pipe with the lane's dgrad.  The step decides: off.
#ifndef MMI_WGRAD_LDS_B32
#define MMI_WGRAD_LDS_B32 0
#endif

// TAB (pixel-table loaders, the wgrad counterpart of the uniform-tap loaders above).  Here K runs over output pixels, so
// what every thread of a row has in common is the pixel: per slab ONE wave (taking turns) writes a 32-entry LDS table
// {byte offset of the pixel's top-left source position, bit mask of the taps that leave the image (all ones past the
// split's end)}; a loader thread adds its own constant tap/channel displacement, tests its own tap bit (2 VALU) and issues
// a buffer load whose masked lanes return zero.  The dy rows need nothing per slab: constant lane offsets against a buffer
// resource that is re-based (scalar arithmetic) to the slab's first pixel and ends at the split's last one.
template <int BM, int BN, bool VEC, int PREC = 0, bool TAB = false>
__global__ __launch_bounds__(256, (BK == 32 && PREC == 0) ? MMI_WGRAD_OCC : ((BK == 32 && (PREC < 2 || PREC >= 4)) ? 3 : 2)) void wgrad_kernel(WgradP p) {
  static_assert(PREC == 0 || (VEC && BK == 32), "the split-bf16 forms exist for the vector loaders only");
  static_assert(!TAB || (VEC && BK == 32), "pixel-table loaders are a form of the vector loaders");
  constexpr bool BF = PREC == 4;   // bf16 storage: dy and x are bf16 in HBM, dw stays fp32 (see igemm_kernel)
  static_assert(!BF || !TAB, "bf16 storage uses the cursor loaders");
  constexpr bool ONE = BF || PREC == 5;   // PREC = 5: fp32 operands, one bf16 term each (see igemm_kernel)
  constexpr int NP = PREC == 0 || ONE ? 1 : (PREC == 3 ? 3 : PREC + 1);
  constexpr int OL = ONE ? 0 : (PREC == 3 ? 2 * (NP - 1) : NP - 1);
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  constexpr int VA = BM / 4, RPA = 256 / VA, ITA = BK / RPA;
  constexpr int VB = BN / 4, RPB = 256 / VB, ITB = BK / RPB;
  // split-bf16 (PREC = 1): both tiles stay k-major in two bf16 planes with rows padded by 64 B; the MFMA operands (8
  // consecutive pixels of one channel) come out of ds_read_b64_tr_b16
  constexpr int A_RSB = BM * 2 + 64, B_RSB = BN * 2 + 64;
  constexpr int A_ELEMS = PREC >= 1 ? NP * BK * A_RSB / 4 : BK * BM, B_ELEMS = PREC >= 1 ? NP * BK * B_RSB / 4 : BK * BN;
  constexpr int STAGE = A_ELEMS + B_ELEMS;
  __shared__ __align__(16) float smem[MMI_WGRAD_STAGES * STAGE];
  __shared__ uint2 ptab[TAB ? 2 : 1][TAB ? BK : 1];  // TAB: {source offset, invalid-tap mask} per pixel row, two slabs
  __shared__ int fold_flag;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, so the (tile, split) pairs are renumbered to put
  // all tiles of one pixel chunk on one XCD back to back: they read the same dy / x rows, which then come out of that
  // XCD's L2 instead of crossing the fabric once per tile (PMC: 885 MB fetched per 3x3 128->128 launch before, 105 MB
  // algorithmic).
  const int ntile_tot = p.mtiles * p.ntiles;
  const int wg = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int split = wg / ntile_tot, tile = wg - split * ntile_tot;
  const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const int kbeg = split * p.chunk;
  const int kend = min(kbeg + p.chunk, p.Mpix);

  // A loader: float4 along co
  const int am = m0 + (t % VA) * 4, akr = t / VA;
  // B loader: float4 along (tap,ci): fixed per thread
  const int bn = n0 + (t % VB) * 4, bkr = t / VB;
  int b_kh[4], b_kw[4], b_ci[4];
  bool b_ok[4];
#pragma unroll
  for (int e = 0; e < (VEC ? 1 : 4); ++e) {
    const int n = bn + e;
    b_ok[e] = n < p.Ntot;
    const int tap = b_ok[e] ? n / p.Cin : 0;
    b_ci[e] = n - tap * p.Cin;
    b_kh[e] = tap / p.KW;
    b_kw[e] = tap - b_kh[e] * p.KW;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[ITA], rb[ITB];
  bf16x4 rab[BF ? ITA : 1], rbb[BF ? ITB : 1];
  const int howo = p.Ho * p.Wo;
  const bool want_bias = (p.OUTB != nullptr) && (nt == 0);  // uniform: the first N-tile of each (M-tile, split)
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

  // division-free pixel cursors for the B (activation) rows: pixel -> (img, oh, ow), advanced by BK per slab
  int k0cur = kbeg;
  int cimg[ITB], coh[ITB], cow[ITB];
#pragma unroll
  for (int i = 0; i < ITB; ++i) {
    const int pix = kbeg + bkr + RPB * i;
    cimg[i] = pix / howo;
    const int rem = pix - cimg[i] * howo;
    coh[i] = rem / p.Wo;
    cow[i] = rem - coh[i] * p.Wo;
  }
  // ---- TAB state ----
  constexpr uint32_t OOB = 0x80000000u;
  uint32_t a_voff[TAB ? ITA : 1], b_tapoff = 0;
  int b_tapbit = 0, tab_sel = 0;
  int timg = 0, toh = 0, tow = 0, tpix = 0;  // this wave's table cursor: pixel kbeg + (wave + 4 j) * BK + lane
  __amdgpu_buffer_rsrc_t srd_x;
  if constexpr (TAB) {
#pragma unroll
    for (int i = 0; i < ITA; ++i) a_voff[i] = am < p.Cout ? (uint32_t)(((akr + RPA * i) * p.ldy + am) * 4) : OOB;
    b_tapbit = b_kh[0] * p.KW + b_kw[0];
    b_tapoff = b_ok[0] ? (uint32_t)(((b_kh[0] * p.W + b_kw[0]) * p.ldx + b_ci[0]) * 4) : OOB;
    const int64_t margin = ((int64_t)p.KH * p.W + p.KW) * p.ldx;
    srd_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.X - margin), 0, (int)p.x_bytes, 0x00020000);
    tpix = kbeg + wave * BK + (lane & (BK - 1));
    timg = tpix / howo;
    const int rem = tpix - timg * howo;
    toh = rem / p.Wo;
    tow = rem - toh * p.Wo;
  }
  // wave (j & 3) writes the table of slab j (lanes 0..BK-1), then moves its cursor four slabs on
  // 1x1 stride-1 layers: x rows are as linear in the pixel index as the dy rows, so they take the same re-based resource
  // and no table at all
  const bool lin1w = TAB && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0;  // uniform
  // With a precomputed table the wave whose turn it is just copies the slab's 32 entries: one 8-byte load per lane, issued a
  // slab early (fetch_table) and stored when the slab's MFMAs are done (build_table) -- 2 instructions instead of ~100 VALU.
  const bool gtab = TAB && p.tab != nullptr;  // uniform
  uint2 tnext = {0u, 0xFFFFFFFFu};
  auto fetch_table = [&](int j) {
    if constexpr (TAB) {
      if (gtab && !lin1w && wave == (j & 3) && lane < BK) tnext = p.tab[(int64_t)kbeg + (int64_t)j * BK + lane];
    }
  };
  auto build_table = [&](int j) {
    if constexpr (TAB) {
      if (gtab) {
        if (!lin1w && wave == (j & 3) && lane < BK) ptab[j & 1][lane] = tnext;
        return;
      }
      if (!lin1w && wave == (j & 3) && lane < BK) {
        uint2 e = {0u, 0xFFFFFFFFu};
        if (tpix < kend) {
          const int ih0 = toh * p.stride - p.pad, iw0 = tow * p.stride - p.pad;
          e.x = (uint32_t)(((((int64_t)timg * p.H + ih0 + p.KH) * p.W + iw0 + p.KW) * p.ldx) * 4);
          uint32_t bw = 0, m = 0;
          for (int kw = 0; kw < p.KW; ++kw) bw |= ((unsigned)(iw0 + kw) >= (unsigned)p.W ? 1u : 0u) << kw;
          const uint32_t roww = (1u << p.KW) - 1u;
          for (int kh = 0; kh < p.KH; ++kh) m |= (((unsigned)(ih0 + kh) >= (unsigned)p.H) ? roww : bw) << (kh * p.KW);
          e.y = m;
        }
        ptab[j & 1][lane] = e;
        tpix += 4 * BK;
        if (howo == 1) {
          timg += 4 * BK;
        } else {
          tow += 4 * BK;
          while (tow >= p.Wo) {
            tow -= p.Wo;
            if (++toh == p.Ho) {
              toh = 0;
              ++timg;
            }
          }
        }
      }
    }
  };
  auto advance = [&]() {
    k0cur += BK;
    if constexpr (TAB) {
      tab_sel ^= 1;
      return;
    }
#pragma unroll
    for (int i = 0; i < ITB; ++i) {
      if (howo == 1) {  // Linear layers: every row is its own 1x1 "image"
        cimg[i] += BK;
        continue;
      }
      cow[i] += BK;
      while (cow[i] >= p.Wo) {
        cow[i] -= p.Wo;
        if (++coh[i] == p.Ho) {
          coh[i] = 0;
          ++cimg[i];
        }
      }
    }
  };
  auto load_a_row = [&](int i) {
    if constexpr (TAB) {
      // the resource starts at the slab's first dy row and ends with the split: rows past the end are out of range -> 0
      const int64_t left = (int64_t)(kend - k0cur) * p.ldy * 4;
      const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(p.DY + (int64_t)k0cur * p.ldy), 0, left > 0 ? (left < 0x7FFFFFFF ? (int)left : 0x7FFFFFFF) : 0, 0x00020000);
      ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_a, a_voff[i], 0, 0));
      return;
    }
    const int pix = k0cur + akr + RPA * i;
    if (VEC) {  // branch-free: an invalid lane reads the base address and is zeroed
      const bool ok = (pix < kend) & (am < p.Cout);
      if constexpr (BF)
        rab[i] = *reinterpret_cast<const bf16x4*>(ok ? reinterpret_cast<const char*>(p.DY) + ((int64_t)pix * p.ldy + am) * 2
                                                     : reinterpret_cast<const char*>(ZERO_SRC));
      else
        ra[i] = *reinterpret_cast<const f32x4*>(ok ? p.DY + (int64_t)pix * p.ldy + am : ZERO_SRC);
    } else {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (pix < kend) {
        const float* src = p.DY + (int64_t)pix * p.ldy + am;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (am + e < p.Cout) v[e] = src[e];
      }
      ra[i] = v;
    }
  };
  auto load_b_row = [&](int i) {
    if constexpr (TAB) {
      if (lin1w) {
        const int64_t left = (int64_t)(kend - k0cur) * p.ldx * 4;
        const __amdgpu_buffer_rsrc_t srd_xs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.X + (int64_t)k0cur * p.ldx), 0, left > 0 ? (left < 0x7FFFFFFF ? (int)left : 0x7FFFFFFF) : 0, 0x00020000);
        const uint32_t voff = b_ok[0] ? (uint32_t)(((bkr + RPB * i) * p.ldx + b_ci[0]) * 4) : OOB;
        rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_xs, voff, 0, 0));
        return;
      }
      const uint2 e = ptab[tab_sel][bkr + RPB * i];
      const uint32_t inv = (uint32_t)__builtin_amdgcn_sbfe((int)e.y, b_tapbit, 1);  // -1: this thread's tap leaves the image
      rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_x, (e.x + b_tapoff) | (inv & OOB), 0, 0));
      return;
    }
    const int pix = k0cur + bkr + RPB * i;
    const int ih0 = coh[i] * p.stride - p.pad, iw0 = cow[i] * p.stride - p.pad;
    if (VEC) {
      const int ih = ih0 + b_kh[0], iw = iw0 + b_kw[0];
      const bool ok = (pix < kend) & b_ok[0] & (ih >= 0) & (iw >= 0) & (ih < p.H) & (iw < p.W);
      if constexpr (BF)
        rbb[i] = *reinterpret_cast<const bf16x4*>(
            ok ? reinterpret_cast<const char*>(p.X) + ((((int64_t)cimg[i] * p.H + ih) * p.W + iw) * p.ldx + b_ci[0]) * 2
               : reinterpret_cast<const char*>(ZERO_SRC));
      else
        rb[i] = *reinterpret_cast<const f32x4*>(
            ok ? p.X + (((int64_t)cimg[i] * p.H + ih) * p.W + iw) * p.ldx + b_ci[0] : ZERO_SRC);
    } else {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (pix < kend) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ih = ih0 + b_kh[e], iw = iw0 + b_kw[e];
          if (b_ok[e] && ih >= 0 && iw >= 0 && ih < p.H && iw < p.W)
            v[e] = p.X[(((int64_t)cimg[i] * p.H + ih) * p.W + iw) * p.ldx + b_ci[e]];
        }
      }
      rb[i] = v;
    }
  };
  auto gload = [&]() {
#pragma unroll
    for (int i = 0; i < ITA; ++i) load_a_row(i);
#pragma unroll
    for (int i = 0; i < ITB; ++i) load_b_row(i);
  };
  auto lstore = [&](int stage) {
    float* As = smem + stage * STAGE;
    float* Bs = As + A_ELEMS;
    if constexpr (PREC >= 1) {
      char* ab = reinterpret_cast<char*>(As);
      char* bb = reinterpret_cast<char*>(Bs);
#pragma unroll
      for (int i = 0; i < ITA; ++i) {
        bf16x4 tm[NP];
        if constexpr (BF) tm[0] = rab[i];
        else split_bf16<NP>(ra[i], tm);
        char* dst = ab + (akr + RPA * i) * A_RSB + (t % VA) * 8;
#pragma unroll
        for (int k = 0; k < NP; ++k) *reinterpret_cast<bf16x4*>(dst + k * BK * A_RSB) = tm[k];
        if (want_bias) {
          if constexpr (BF) {
#pragma unroll
            for (int e = 0; e < 4; ++e) bsum[e] += (float)rab[i][e];
          } else {
            bsum += ra[i];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < ITB; ++i) {
        bf16x4 tm[NP];
        if constexpr (BF) tm[0] = rbb[i];
        else split_bf16<NP>(rb[i], tm);
        char* dst = bb + (bkr + RPB * i) * B_RSB + (t % VB) * 8;
#pragma unroll
        for (int k = 0; k < NP; ++k) *reinterpret_cast<bf16x4*>(dst + k * BK * B_RSB) = tm[k];
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < ITA; ++i) {
      *reinterpret_cast<f32x4*>(As + (akr + RPA * i) * BM + (t % VA) * 4) = ra[i];
      if (want_bias) bsum += ra[i];  // the dy tile passes through here exactly once: its column sums are the bias gradient
    }
#pragma unroll
    for (int i = 0; i < ITB; ++i) *reinterpret_cast<f32x4*>(Bs + (bkr + RPB * i) * BN + (t % VB) * 4) = rb[i];
  };

  const int nk = (kend - kbeg + BK - 1) / BK;
  const int l31 = lane & 31, lh = lane >> 5;
  if constexpr (TAB) {
    fetch_table(0);
    build_table(0);
    fetch_table(1);
    build_table(1);
    __syncthreads();
  }
  if (nk > 0) {
    gload();
    lstore(0);
  }
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    advance();  // unconditional prefetch of the next slab (lanes past the split's end read the zero source)
    fetch_table(ks + 2);
    const float* As = smem + (MMI_WGRAD_STAGES == 2 ? (ks & 1) : 0) * STAGE;
    const float* Bs = As + A_ELEMS;
    if constexpr (PREC >= 1) {
      const int q = (lane & 15) >> 2, pp = lane & 3, m0 = ((lane >> 4) & 1) * 16;
#pragma unroll
      for (int kb = 0; kb < BK / 16; ++kb) {
#pragma unroll
        for (int i = 0; i < ITA; ++i)
          if (kb == 0 ? (i % 3 != 2) : (i % 3 == 2)) load_a_row(i);
#pragma unroll
        for (int i = 0; i < ITB; ++i)
          if (kb == 0 ? ((ITA + i) % 3 != 2) : ((ITA + i) % 3 == 2)) load_b_row(i);
        const char* ab = reinterpret_cast<const char*>(As) + (kb * 16 + lh * 8 + q) * A_RSB + (wm * WM + m0 + 4 * pp) * 2;
        const char* bb = reinterpret_cast<const char*>(Bs) + (kb * 16 + lh * 8 + q) * B_RSB + (wn * WN + m0 + 4 * pp) * 2;
        bf16x8 af[NP][TM], bf[NP][TN];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
#pragma unroll
          for (int i = 0; i < TM; ++i) af[k][i] = tr_read8(ab + i * 64 + k * BK * A_RSB, A_RSB);
#pragma unroll
          for (int j = 0; j < TN; ++j) bf[k][j] = tr_read8(bb + j * 64 + k * BK * B_RSB, B_RSB);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int o = OL; o >= 0; --o)
#pragma unroll
              for (int ka = (o > NP - 1 ? o - (NP - 1) : 0); ka <= (o < NP - 1 ? o : NP - 1); ++ka)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ka][i], bf[o - ka][j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {  // groups of four k-steps: all fragment reads up front, then 4*TM*TN MFMAs
      // a third of the next slab's loads ahead of each of the first three groups
#pragma unroll
      for (int i = 0; i < ITA; ++i)
        if (MMI_LOAD_SPREAD(i, g)) load_a_row(i);
#pragma unroll
      for (int i = 0; i < ITB; ++i)
        if (MMI_LOAD_SPREAD(ITA + i, g)) load_b_row(i);
      float a[4][TM], b[4][TN];
#if MMI_WGRAD_LDS_B32
      // One ds_read_b32 per fragment, each with its own 16-bit immediate offset from ONE per-thread base: left to itself the
      // compiler pairs the fragments into ds_read2_b32, whose 8-bit offsets do not reach from one k-step to the next (1 KB), and
      // pays a v_add_u32 per pair -- 28 VALU instructions per K slab next to the MFMA stream (tools/mfma_mix.hip: LDS reads
      // cost the matrix pipe nothing, VALU instructions do).  `volatile` is what keeps the reads apart.
      typedef __attribute__((address_space(3))) const volatile float* lds_vfp;   // (stays an LDS access: ds_read_b32)
      lds_vfp ap = (lds_vfp)(As + lh * BM + wm * WM + l31);
      lds_vfp bp = (lds_vfp)(Bs + lh * BN + wn * WN + l31);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[e][i] = ap[2 * (4 * g + e) * BM + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[e][j] = bp[2 * (4 * g + e) * BN + j * 32];
      }
#else
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[e][i] = As[(2 * (4 * g + e) + lh) * BM + wm * WM + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[e][j] = Bs[(2 * (4 * g + e) + lh) * BN + wn * WN + j * 32 + l31];
      }
#endif
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e][i], b[e][j], acc[i][j], 0, 0, 0);
    }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MMI_WGRAD_STAGES == 1) __syncthreads();
    lstore(MMI_WGRAD_STAGES == 2 ? ((ks + 1) & 1) : 0);
    // read during the next iteration (its loads are those of slab ks + 2); shares a buffer with slab ks.  (Built at the top of
    // the iteration instead, in the shadow of the MFMAs, it costs 3-9 %: VALU next to the MFMA stream again.)
    build_table(ks + 2);
    __syncthreads();
  }

  const bool fold = p.cnt != nullptr;  // uniform
  if (want_bias) {  // fold the RPA row-lanes of each channel quad through LDS (free after the loop's last barrier)
    float* red = smem;  // [RPA][BM]
    *reinterpret_cast<f32x4*>(red + akr * BM + (t % VA) * 4) = bsum;
    __syncthreads();
    if (t < BM && m0 + t < p.Cout) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < RPA; ++i) s += red[i * BM + t];
      float* dst = p.OUTB + (int64_t)split * p.slab_stride + m0 + t;
      if (fold) st_agent(dst, s);
      else *dst = s;
    }
  }
  float* out = p.OUT + (int64_t)split * p.slab_stride;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * WN + j * 32 + l31;
    if (col < p.Ntot) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row < p.Cout) {
            if (fold) st_agent(out + (int64_t)row * p.Ntot + col, acc[i][j][r]);
            else out[(int64_t)row * p.Ntot + col] = acc[i][j][r];
          }
        }
    }
  }
  if (!fold) return;
  // ---- split-K fold inside the launch, as a tree of fan-in 4 ------------------------------------------------------------
  // Level 0 holds the splits' slabs.  At every level the members of a group of four consecutive nodes arrive on the group's
  // counter; the last one sums the group (in node order: deterministic) into the slab of the group's first member -- which
  // becomes the node of the next level -- and goes on to arrive there; the group that is alone at its level writes dW (and
  // dbias) instead.  A fold therefore never reads more than four slabs, whatever the split count (one workgroup walking a
  // long list serially was 2x slower than the separate reduce launch: profiles/r02_wgrad_fold_microbench.txt), and the
  // folds of different groups run on different workgroups.  Counters: p.cnt + tile * p.cnt_per_tile, level after level.
  int node = split, nodes = p.splits, stride = 1;          // stride: slab distance between neighbouring nodes of this level
  int* cnt = p.cnt + (int64_t)tile * p.cnt_per_tile;
  while (true) {
    const int group = node >> 2, gfirst = group << 2, gsize = min(4, nodes - gfirst);
    mmi_drain_stores();  // every storing wave: its sc1 (write-through) stores have left the CU before the barrier below
    __syncthreads();
    if (t == 0) {
      const int last = __hip_atomic_fetch_add(cnt + group, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1;
      if (last) __hip_atomic_store(cnt + group, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      fold_flag = last;
    }
    __syncthreads();
    if (!fold_flag) return;
    const bool root = nodes <= 4;
    const int64_t z0 = (int64_t)gfirst * stride;             // slab of the group's first member (and of its sum)
    // one 32x32 sub-tile at a time (16 values per lane); the group's loads in flight together, summed in node order
#pragma unroll 1
    for (int ij = 0; ij < TM * TN; ++ij) {
      const int i = ij / TN, j = ij - i * TN;
      const int col = n0 + wn * WN + j * 32 + l31;
      const int rbase = m0 + wm * WM + i * 32 + 4 * lh;
      if (col >= p.Ntot) continue;
      const int64_t e0 = (int64_t)rbase * p.Ntot + col;
      float u[4][16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float* part = p.OUT + (z0 + (int64_t)min(q, gsize - 1) * stride) * p.slab_stride + e0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          u[q][r] = rbase + dr < p.Cout ? ld_agent(part + (int64_t)dr * p.Ntot) : 0.f;
        }
      }
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = u[0][r];
#pragma unroll
      for (int q = 1; q < 4; ++q)
        if (q < gsize) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] += u[q][r];
        }
      float* dst = root ? p.DW + e0 : p.OUT + z0 * p.slab_stride + e0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        if (rbase + dr < p.Cout) {
          if (root) dst[(int64_t)dr * p.Ntot] = v[r];
          else st_agent(dst + (int64_t)dr * p.Ntot, v[r]);
        }
      }
    }
    if (want_bias && t < BM && m0 + t < p.Cout) {
      float sb = 0.f;
      for (int q = 0; q < gsize; ++q) sb += ld_agent(p.OUTB + (z0 + (int64_t)q * stride) * p.slab_stride + m0 + t);
      if (root) p.DB[m0 + t] = sb;
      else st_agent(p.OUTB + z0 * p.slab_stride + m0 + t, sb);
    }
    if (root) return;
    cnt += (nodes + 3) >> 2;          // next level's counters follow this level's
    node = group;
    nodes = (nodes + 3) >> 2;
    stride <<= 2;
  }
}

// The pixel table of a layer geometry, entry p = output pixel p: exactly what wgrad_kernel's in-kernel builder produces
// (source byte offset of the pixel's top-left tap incl. the margin; bit t set = tap t leaves the image), followed by
// invalid entries for the slabs a split may prefetch past the last pixel.
__global__ void wgrad_table_kernel(uint2* __restrict__ tab, int Mpix, int total, int Ho, int Wo, int H, int W, int KH, int KW,
                                   int stride, int pad, int ldx) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= total) return;
  uint2 e = {0u, 0xFFFFFFFFu};
  if (p < Mpix) {
    const int howo = Ho * Wo, img = p / howo, rem = p - img * howo, oh = rem / Wo, ow = rem - oh * Wo;
    const int ih0 = oh * stride - pad, iw0 = ow * stride - pad;
    e.x = (uint32_t)(((((int64_t)img * H + ih0 + KH) * W + iw0 + KW) * ldx) * 4);
    uint32_t bw = 0, m = 0;
    for (int kw = 0; kw < KW; ++kw) bw |= ((unsigned)(iw0 + kw) >= (unsigned)W ? 1u : 0u) << kw;
    const uint32_t roww = (1u << KW) - 1u;
    for (int kh = 0; kh < KH; ++kh) m |= (((unsigned)(ih0 + kh) >= (unsigned)H) ? roww : bw) << (kh * KW);
    e.y = m;
  }
  tab[p] = e;
}

// out = sum over splits of slabs[z]: 16-byte lanes, 4 independent loads in flight per thread (HBM-bound)
// (elements [0, n1) go to out, the bias tail [n1, n) to out2)
__global__ void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, float* __restrict__ out2,
                                   int64_t n1, int64_t n, int64_t count, int splits) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= count) return;  // count = n (with the bias tail) or n1 (without); n is the slab stride
  if (i + 4 <= n1 && (n & 3) == 0) {
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int z = 0;
    for (; z + 4 <= splits; z += 4) {
      s0 += *reinterpret_cast<const f32x4*>(slabs + (int64_t)z * n + i);
      s1 += *reinterpret_cast<const f32x4*>(slabs + (int64_t)(z + 1) * n + i);
      s2 += *reinterpret_cast<const f32x4*>(slabs + (int64_t)(z + 2) * n + i);
      s3 += *reinterpret_cast<const f32x4*>(slabs + (int64_t)(z + 3) * n + i);
    }
    for (; z < splits; ++z) s0 += *reinterpret_cast<const f32x4*>(slabs + (int64_t)z * n + i);
    *reinterpret_cast<f32x4*>(out + i) = (s0 + s1) + (s2 + s3);
  } else {
    for (int64_t j = i; j < count && j < i + 4; ++j) {
      float s = 0.f;
      for (int z = 0; z < splits; ++z) s += slabs[(int64_t)z * n + j];
      if (j < n1) out[j] = s;
      else out2[j - n1] = s;
    }
  }
}

struct FwdPlan {
  int bm, bn, mtiles, ntiles;
  int sk_grid;  // > 0: stream-K schedule over this many workgroups (needs the workspace), 0: one workgroup per tile
};
// mmi_set_uniform_loaders / MMIDET_UNIFORM_LOADERS=0 (A/B switch): 1 = use the uniform-tap loaders where they apply
int g_uniform_loaders = getenv("MMIDET_UNIFORM_LOADERS") ? atoi(getenv("MMIDET_UNIFORM_LOADERS")) : 1;
int g_gemm_prec = 0;  // mmi_set_gemm_precision: 0 = exact fp32 MFMA, 1 = split-bf16 products for forward-layout GEMMs
int g_tile_bm = 0, g_tile_bn = 0;  // mmi_set_tile_override (tuning): force one tile variant, one workgroup per tile
int g_wgrad_force[3] = {0, 0, 0};  // mmi_set_wgrad_override (tuning): bm, bn, splits (0 = automatic)
FwdPlan plan_tiles(int64_t M, int Ncol) {
  FwdPlan f;
  if (g_tile_bm > 0) {
    f.bm = g_tile_bm;
    f.bn = g_tile_bn;
    f.mtiles = cdiv(M, f.bm);
    f.ntiles = cdiv(Ncol, f.bn);
    f.sk_grid = 0;
    return f;
  }
  f.bn = Ncol > 64 ? 128 : 64;
  f.bm = 128;
  // small problems: shrink the tile until there are >= 2 workgroups per CU (256 CUs)
  if (f.bn == 128 && (int64_t)cdiv(M, 128) * cdiv(Ncol, 128) < 512) f.bn = 64;
  if ((int64_t)cdiv(M, 128) * cdiv(Ncol, f.bn) < 512) f.bm = 64, f.bn = 64;
  f.mtiles = cdiv(M, f.bm);
  f.ntiles = cdiv(Ncol, f.bn);
  f.sk_grid = 0;
  return f;
}

int g_sk_slots = 0;  // mmi_set_streamk_slots: 0 = chip-sized, > 0 = this many workgroups, < 0 = schedule off
constexpr int SK_MAX_TILES = 65536;                                  // arrival counters at the head of the workspace
constexpr size_t SK_COUNTER_BYTES = (size_t)SK_MAX_TILES * sizeof(int);

int device_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      n = prop.multiProcessorCount;
    else
      n = 256;  // MI355X; also what a GPU-less host plans with
    (void)hipGetLastError();
  }
  return n;
}

// Resident workgroups per CU of a stream-K kernel variant (registers and LDS decide; 3 by the launch bound).
template <bool DGRAD>
int sk_occupancy(int bn) {
  static int cache[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
  int& c = cache[g_gemm_prec][bn == 128 ? 1 : 0];
  if (c == 0) {
    int n = 0;
    hipError_t e;
#define OCC(P_) (bn == 128 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, igemm_kernel<128, 128, DGRAD, true, true, P_>, 256, 0) \
                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, igemm_kernel<128, 64, DGRAD, true, true, P_>, 256, 0))
    // (fp32: the uniform-tap variant is what nearly every stream-K shape runs; the few others fit its grid as well)
    if (g_gemm_prec == 0 && g_uniform_loaders)
      e = bn == 128 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, igemm_kernel<128, 128, DGRAD, true, true, 0, false, true>, 256, 0)
                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, igemm_kernel<128, 64, DGRAD, true, true, 0, false, true>, 256, 0);
    else
      e = g_gemm_prec == 0 ? OCC(0) : (g_gemm_prec == 1 ? OCC(1) : (g_gemm_prec == 2 ? OCC(2) : OCC(3)));
#undef OCC
    c = (e == hipSuccess && n > 0) ? n : (g_gemm_prec >= 2 ? 2 : 3);
    (void)hipGetLastError();
  }
  return c;
}

// Schedule choice for the vector kernels.  A kernel's time grows in steps of one workgroup per CU (769 tiles of 128x128
// cost as much as 1024), so:
//  * long K (>= 32 slabs per resident workgroup): stream-K over 128-wide tiles unless one workgroup per tile already
//    fills the chip evenly (>= 93 % of the last "layer" of 256);
//  * short K (1x1 convs, token projections; tools/sweep_tiles.py): one workgroup per tile, plan_tiles' shrink rule.
template <bool DGRAD>
FwdPlan plan_igemm(int64_t M, int Ncol, int Ktot, bool vec, bool allow_sk) {
  FwdPlan f = plan_tiles(M, Ncol);
  if (!vec && f.bn == 128) f.bn = 64, f.ntiles = cdiv(Ncol, 64);
  static const bool off = getenv("MMIDET_NO_STREAMK") != nullptr;
  const int nk = cdiv(Ktot, BK);
  if (!vec || !allow_sk || g_tile_bm > 0 || g_gemm_prec == 5) return f;   // (mode 5: short launches, one workgroup per tile)
  FwdPlan g;
  g.bm = 128;
  g.bn = Ncol > 64 ? 128 : 64;
  g.mtiles = cdiv(M, 128);
  g.ntiles = cdiv(Ncol, g.bn);
  g.sk_grid = 0;
  const int64_t tiles = (int64_t)g.mtiles * g.ntiles;
  const int cus = device_cus();
  const int slots = g_sk_slots > 0 ? g_sk_slots : cus * sk_occupancy<DGRAD>(g.bn);
  const double layers = (double)tiles / cus, dp_eff = layers / ceil(layers);
  const bool sk_ok = !off && g_sk_slots >= 0 && nk >= (g_sk_slots > 0 ? 2 : 16) && tiles <= SK_MAX_TILES &&
                     tiles * nk >= (int64_t)(g_sk_slots > 0 ? 1 : 32) * slots && tiles * nk < (1LL << 31);
  if (sk_ok) {
    if (g_sk_slots == 0 && tiles >= slots && dp_eff >= 0.93) return g;  // already even
    g.sk_grid = slots;
    return g;
  }
  // short K on many rows (the 1x1 convs): with the uniform-tap loaders the 64x64 tile no longer pays more address
  // arithmetic per MFMA than the wide ones, and its finer grid wins 10-20 % stand-alone (tools/sweep_tiles.py)
  // (in the step: 126.7 -> 125.8 ms; MMIDET_SHORTK_TILE=0 restores plan_tiles' choice)
  static const int shortk_tile = getenv("MMIDET_SHORTK_TILE") ? atoi(getenv("MMIDET_SHORTK_TILE")) : 64;
  if (shortk_tile == 64 && g_gemm_prec == 0 && g_uniform_loaders && nk <= 64 && Ncol <= 1024 && M >= 8192) {
    f.bm = f.bn = 64;
    f.mtiles = cdiv(M, 64);
    f.ntiles = cdiv(Ncol, 64);
    return f;
  }
  // short K, or too little work per workgroup for stream-K: one workgroup per tile with the tile shrunk until the grid has
  // two workgroups per CU (plan_tiles).  Finer 64x64 tiles win 10-16 % on the 1x1 layers stand-alone
  // (profiles/r01_tile_sweep_short_k.txt) but nothing inside the step, where the other lane fills the tail; always taking the
  // 128-wide tile loses 4 %.
  return f;
}

// Workspace of a forward / dgrad launch (zero-filled when first handed over, self-cleaning afterwards):
//   [0, SK_COUNTER_BYTES)            stream-K arrival counters
//   [.., + BN_COUNTER_BYTES)         arrival counters of the in-launch BatchNorm statistics fold (forward)
//   [WS_HEADER_BYTES, ..)            no zero-fill needed: level-1 statistics partials (forward), then the stream-K slots
// The header is the same for every shape and direction, because launches on one stream share one buffer.
constexpr size_t BN_COUNTER_BYTES = (size_t)MMI_STAT_MAX_COUNTERS * sizeof(int);
constexpr size_t WS_HEADER_BYTES = SK_COUNTER_BYTES + BN_COUNTER_BYTES;
size_t bn_l1_bytes(const FwdPlan& f, int Ncol) {
  const int G = stat_group_size(f.mtiles);
  return (((size_t)cdiv(f.mtiles, G) * 2 * Ncol * sizeof(float)) + 15) & ~(size_t)15;
}
bool bn_fold_fits(const FwdPlan& f) {
  const int ngroups = cdiv(f.mtiles, stat_group_size(f.mtiles));
  return (int64_t)ngroups * f.ntiles + f.ntiles <= MMI_STAT_MAX_COUNTERS;
}
size_t sk_slot_bytes(const FwdPlan& f) { return f.sk_grid > 0 ? (size_t)f.sk_grid * 2 * f.bm * f.bn * sizeof(float) : 0; }
size_t sk_workspace_bytes(const FwdPlan& f) { return f.sk_grid > 0 ? WS_HEADER_BYTES + sk_slot_bytes(f) : 0; }
size_t fwd_workspace_bytes(const FwdPlan& f, int Ncol) { return WS_HEADER_BYTES + bn_l1_bytes(f, Ncol) + sk_slot_bytes(f); }

template <bool DGRAD, bool EPI = false>
int launch_igemm(const IgemmP& p0, const FwdPlan& f, bool vec, void* workspace, size_t workspace_bytes, hipStream_t s,
                 size_t slot_offset = WS_HEADER_BYTES) {
  const char* who = DGRAD ? "mmi_conv_dgrad" : "mmi_conv_fwd";
  IgemmP p = p0;
  p.zero = zero_src();
  if (p.zero == nullptr) {
    mmi_set_error("igemm: cannot resolve the zero-source symbol");
    return MMI_ERR_LAUNCH;
  }
  p.mtiles = f.mtiles;
  p.ntiles = f.ntiles;
  // uniform-tap loaders (igemm_kernel<..., UNI>): whole slabs inside one tap, tap table in 32 bits, 31-bit byte offsets
  bool uni = false;
  if (vec && g_uniform_loaders && g_gemm_prec == 0 && p.Kc % BK == 0 && p.KH * p.KW <= 32 && !(DGRAD && p.stride == 2 && !p.par)) {
    const int64_t margin = ((int64_t)p.KH * p.Ws + p.KW) * p.lda;
    const int64_t npix = (int64_t)(p.M / ((int64_t)p.P * p.Q)) * p.Hs * p.Ws;
    const int64_t a_bytes = (margin + (npix - 1) * p.lda + p.Kc) * 4;
    const int64_t b_bytes = DGRAD ? (int64_t)p.Kc * p.ldb * 4 : (int64_t)p.Ncol * p.ldb * 4;
    if (a_bytes < (1LL << 31) && b_bytes < (1LL << 31)) {
      uni = true;
      p.a_bytes = (uint32_t)a_bytes;
      p.b_bytes = (uint32_t)b_bytes;
      const int64_t c_bytes = ((int64_t)(p.M - 1) * p.ldc + p.Ncol) * 4;
      p.c_bytes = c_bytes < (1LL << 31) ? (uint32_t)c_bytes : 0u;
    }
  }
  if (f.sk_grid > 0) {
    if (workspace == nullptr || workspace_bytes < slot_offset + sk_slot_bytes(f) || ((uintptr_t)workspace & 15)) {
      mmi_set_error("%s: this shape runs the stream-K schedule and needs a 16-byte aligned workspace of %zu bytes (got %zu)",
                    who, slot_offset + sk_slot_bytes(f), workspace_bytes);
      return MMI_ERR_WORKSPACE;
    }
    p.sk_count = (int*)workspace;
    p.sk_slots = (float*)((char*)workspace + slot_offset);
    const dim3 grid(f.sk_grid), block(256);
    if (g_gemm_prec == 1) {
      if (f.bn == 128) hipLaunchKernelGGL((igemm_kernel<128, 128, DGRAD, true, true, 1, EPI>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((igemm_kernel<128, 64, DGRAD, true, true, 1, EPI>), grid, block, 0, s, p);
      MMI_CHECK_LAUNCH(who);
      return MMI_OK;
    }
    if (g_gemm_prec == 2) {
      if (f.bn == 128) hipLaunchKernelGGL((igemm_kernel<128, 128, DGRAD, true, true, 2, EPI>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((igemm_kernel<128, 64, DGRAD, true, true, 2, EPI>), grid, block, 0, s, p);
      MMI_CHECK_LAUNCH(who);
      return MMI_OK;
    }
    if (g_gemm_prec == 3) {
      if (f.bn == 128) hipLaunchKernelGGL((igemm_kernel<128, 128, DGRAD, true, true, 3, EPI>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((igemm_kernel<128, 64, DGRAD, true, true, 3, EPI>), grid, block, 0, s, p);
      MMI_CHECK_LAUNCH(who);
      return MMI_OK;
    }
    if (uni) {
      if (f.bn == 128) hipLaunchKernelGGL((igemm_kernel<128, 128, DGRAD, true, true, 0, EPI, true>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((igemm_kernel<128, 64, DGRAD, true, true, 0, EPI, true>), grid, block, 0, s, p);
    } else if (f.bn == 128) hipLaunchKernelGGL((igemm_kernel<128, 128, DGRAD, true, true, 0, EPI>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((igemm_kernel<128, 64, DGRAD, true, true, 0, EPI>), grid, block, 0, s, p);
    MMI_CHECK_LAUNCH(who);
    return MMI_OK;
  }
  const dim3 grid(f.mtiles * f.ntiles, p.par ? 4 : 1), block(256);
#define LAUNCH(BM_, BN_, VEC_)                                                                      \
  hipLaunchKernelGGL((igemm_kernel<BM_, BN_, DGRAD, VEC_, false, 0, (VEC_) && EPI>), grid, block, 0, s, p)
  if (vec && g_gemm_prec >= 1) {
#define LAUNCH_B3(BM_, BN_)                                                                                          \
  do {                                                                                                               \
    if (g_gemm_prec == 1) hipLaunchKernelGGL((igemm_kernel<BM_, BN_, DGRAD, true, false, 1, EPI>), grid, block, 0, s, p); \
    else if (g_gemm_prec == 2) hipLaunchKernelGGL((igemm_kernel<BM_, BN_, DGRAD, true, false, 2, EPI>), grid, block, 0, s, p); \
    else if (g_gemm_prec == 5) hipLaunchKernelGGL((igemm_kernel<BM_, BN_, DGRAD, true, false, 5, EPI>), grid, block, 0, s, p); \
    else hipLaunchKernelGGL((igemm_kernel<BM_, BN_, DGRAD, true, false, 3, EPI>), grid, block, 0, s, p);                  \
  } while (0)
    if (f.bm == 128 && f.bn == 128) LAUNCH_B3(128, 128);
    else if (f.bm == 128 && f.bn == 64) LAUNCH_B3(128, 64);
    else LAUNCH_B3(64, 64);
#undef LAUNCH_B3
    MMI_CHECK_LAUNCH(who);
    return MMI_OK;
  }
  if (!vec) {
    if (EPI) {
      mmi_set_error("%s: the fused Linear epilogues need channel counts and row strides that are multiples of 4", who);
      return MMI_ERR_ARG;
    }
    if (f.bm == 128 && f.bn == 64) LAUNCH(128, 64, false);
    else LAUNCH(64, 64, false);
  } else if (uni) {
#define LAUNCH_UNI(BM_, BN_) hipLaunchKernelGGL((igemm_kernel<BM_, BN_, DGRAD, true, false, 0, EPI, true>), grid, block, 0, s, p)
    if constexpr (DGRAD && !EPI) {
      static const bool w41_off = getenv("MMIDET_DGRAD_W41") != nullptr && atoi(getenv("MMIDET_DGRAD_W41")) == 0;   // (A/B switch)
      if (f.bm == 128 && f.bn == 64 && p.Ncol <= 32 && !w41_off) {
        hipLaunchKernelGGL((igemm_kernel<128, 64, true, true, false, 0, false, true, true>), grid, block, 0, s, p);
        MMI_CHECK_LAUNCH(who);
        return MMI_OK;
      }
    }
    if (f.bm == 128 && f.bn == 128) LAUNCH_UNI(128, 128);
    else if (f.bm == 128 && f.bn == 64) LAUNCH_UNI(128, 64);
    else LAUNCH_UNI(64, 64);
#undef LAUNCH_UNI
  } else if (f.bm == 128 && f.bn == 128) LAUNCH(128, 128, true);
  else if (f.bm == 128 && f.bn == 64) LAUNCH(128, 64, true);
  else LAUNCH(64, 64, true);
#undef LAUNCH
  MMI_CHECK_LAUNCH(who);
  return MMI_OK;
}

int check_desc(const mmi_conv_desc* d, const char* who) {
  MMI_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
  MMI_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "%s: non-positive dims", who);
  MMI_CHECK_ARG(d->KH == d->KW && (d->KH == 1 || d->KH == 3), "%s: kernel %dx%d unsupported (1x1, 3x3)", who, d->KH, d->KW);
  MMI_CHECK_ARG(d->stride == 1 || d->stride == 2, "%s: stride %d unsupported", who, d->stride);
  MMI_CHECK_ARG(d->pad == d->KH / 2, "%s: pad %d != k/2", who, d->pad);
  MMI_CHECK_ARG(d->Ho == (d->H + 2 * d->pad - d->KH) / d->stride + 1 && d->Wo == (d->W + 2 * d->pad - d->KW) / d->stride + 1,
                "%s: output dims (%d,%d) inconsistent", who, d->Ho, d->Wo);
  MMI_CHECK_ARG(d->ldx >= d->Cin && d->ldy >= d->Cout, "%s: row strides smaller than channel counts", who);
  MMI_CHECK_ARG((int64_t)d->N * d->H * d->W < (1LL << 31) && (int64_t)d->KH * d->KW * d->Cin < (1 << 24), "%s: size overflow", who);
  return MMI_OK;
}

bool fwd_vec(const mmi_conv_desc* d) { return d->Cin % 4 == 0 && d->ldx % 4 == 0; }
FwdPlan fwd_plan(const mmi_conv_desc* d) {
  return plan_igemm<false>((int64_t)d->N * d->Ho * d->Wo, d->Cout, d->KH * d->KW * d->Cin, fwd_vec(d), true);
}
bool dgrad_vec(const mmi_conv_desc* d) { return d->Cout % 4 == 0 && d->ldy % 4 == 0 && d->Cin % 4 == 0; }
bool dgrad_par(const mmi_conv_desc* d) { return dgrad_vec(d) && d->stride == 2 && d->KH == 3; }
FwdPlan dgrad_plan(const mmi_conv_desc* d) {
  // parity mode: the grid is sized for the largest class (ceil(H/2) x ceil(W/2) pixels per image); it keeps the
  // data-parallel schedule (four K extents in one launch)
  const bool par = dgrad_par(d);
  const int64_t mrows = par ? (int64_t)d->N * ((d->H + 1) / 2) * ((d->W + 1) / 2) : (int64_t)d->N * d->H * d->W;
  return plan_igemm<true>(mrows, d->Cin, d->KH * d->KW * d->Cout, dgrad_vec(d), !par);
}

}  // namespace

extern "C" int mmi_set_gemm_precision(int mode) {
  MMI_CHECK_ARG((mode >= 0 && mode <= 3) || mode == 5,
                "mmi_set_gemm_precision: mode %d (0 = fp32 MFMA, 1 = bf16x3, 2 = bf16x6, 3 = bf16x9, 5 = bf16x1)", mode);
  g_gemm_prec = mode;
  return MMI_OK;
}

extern "C" int mmi_set_uniform_loaders(int on) {
  const int old = g_uniform_loaders;
  g_uniform_loaders = on ? 1 : 0;
  return old;
}

extern "C" int mmi_set_tile_override(int bm, int bn) {
  const bool ok = (bm == 0 && bn == 0) || (bm == 128 && (bn == 128 || bn == 64)) || (bm == 64 && bn == 64);
  MMI_CHECK_ARG(ok, "mmi_set_tile_override: (%d,%d) is not a kernel variant (128x128, 128x64, 64x64; 0,0 = automatic)", bm, bn);
  g_tile_bm = bm;
  g_tile_bn = bn;
  return MMI_OK;
}

extern "C" int mmi_set_wgrad_override(int bm, int bn, int splits) {
  const bool ok = (bm == 0 && bn == 0) || ((bm == 128 || bm == 64) && (bn == 128 || bn == 64));
  MMI_CHECK_ARG(ok && splits >= 0, "mmi_set_wgrad_override: (%d,%d,%d): tiles are 128/64 x 128/64 (0,0 = automatic), splits >= 0 (0 = off)", bm, bn, splits);
  g_wgrad_force[0] = bm; g_wgrad_force[1] = bn; g_wgrad_force[2] = splits;
  return MMI_OK;
}

extern "C" int mmi_set_streamk_slots(int slots) {
  const int old = g_sk_slots;
  g_sk_slots = slots;
  return old;
}

extern "C" int mmi_conv_fwd_row_blocks(const mmi_conv_desc* d) {
  if (check_desc(d, "mmi_conv_fwd_row_blocks") != MMI_OK) return MMI_ERR_ARG;
  if (mmi_smallconv_supported(d)) return mmi_smallconv_blocks(d);  // CEM layers: direct VALU conv (cem.hip)
  return fwd_plan(d).mtiles;
}

extern "C" size_t mmi_conv_fwd_workspace(const mmi_conv_desc* d) {
  if (check_desc(d, "mmi_conv_fwd_workspace") != MMI_OK || mmi_smallconv_supported(d)) return 0;
  return fwd_workspace_bytes(fwd_plan(d), d->Cout);
}

extern "C" size_t mmi_conv_dgrad_workspace(const mmi_conv_desc* d) {
  if (check_desc(d, "mmi_conv_dgrad_workspace") != MMI_OK || mmi_smallconv_dgrad_supported(d)) return 0;
  return sk_workspace_bytes(dgrad_plan(d));
}

namespace {
int conv_fwd_impl(const float* x, const float* w, const float* bias, const float* residual, int ldr, int act, float* y,
                  float* stat_partials, void* workspace, size_t workspace_bytes, const mmi_conv_desc* d, void* stream,
                  const char* who, const mmi_bn_stats* bn = nullptr) {
  if (int e = check_desc(d, who)) return e;
  MMI_CHECK_ARG(x && w && y, "%s: null pointer", who);
  MMI_CHECK_ARG(!(bias && stat_partials), "%s: bias and BN statistics are mutually exclusive", who);
  const bool plain = residual == nullptr && act == MMI_ACT_NONE;
  const int64_t rows = (int64_t)d->N * d->Ho * d->Wo;
  if (mmi_smallconv_supported(d) && plain) {
    MMI_CHECK_ARG(d->Cin != 24 || ((uintptr_t)x & 15) == 0, "%s: operands must be 16-byte aligned", who);
    if (int e = mmi_smallconv_fwd(x, w, bias, y, stat_partials, d, (hipStream_t)stream)) return e;
    if (bn == nullptr) return MMI_OK;   // the CEM's direct convolutions keep the separate fold
    if (int e = mmi_bn_finalize(stat_partials, mmi_smallconv_blocks(d), rows, d->Cout, bn->eps, bn->momentum, bn->running_mean,
                                bn->running_var, bn->num_batches_tracked, bn->mean_invstd, stream)) return e;
    if (bn->num_batches_tracked2 != nullptr) {
      mmi_set_error("%s: two num_batches_tracked counters are not supported on this path", who);
      return MMI_ERR_ARG;
    }
    return MMI_OK;
  }
  const bool vec = fwd_vec(d);
  MMI_CHECK_ARG(!vec || (((uintptr_t)x | (uintptr_t)w) & 15) == 0, "%s: operands must be 16-byte aligned", who);
  IgemmP p{};
  p.A = x; p.B = w; p.C = y; p.bias = bias; p.stat_part = stat_partials;
  p.res = residual; p.ldr = ldr; p.act = act;
  p.M = d->N * d->Ho * d->Wo; p.Ncol = d->Cout; p.Kc = d->Cin; p.KH = d->KH; p.KW = d->KW;
  p.P = d->Ho; p.Q = d->Wo; p.Hs = d->H; p.Ws = d->W; p.lda = d->ldx; p.ldc = d->ldy;
  p.stride = d->stride; p.pad = d->pad; p.Ktot = d->KH * d->KW * d->Cin; p.ldb = p.Ktot;
  const FwdPlan f = fwd_plan(d);
  static const bool bn_fold_off = getenv("MMIDET_BN_FOLD") != nullptr && atoi(getenv("MMIDET_BN_FOLD")) == 0;  // (A/B switch)
  const bool fold = bn != nullptr && !bn_fold_off && bn_fold_fits(f) &&
                    (bn->num_batches_tracked2 == nullptr || bn->num_batches_tracked2 == bn->num_batches_tracked + 1);
  if (fold) {
    MMI_CHECK_ARG(stat_partials && bn->mean_invstd, "%s: BN statistics need the partials buffer and mean_invstd", who);
    MMI_CHECK_ARG((bn->running_mean == nullptr) == (bn->running_var == nullptr), "%s: running stats must come in pairs", who);
    if (workspace == nullptr || workspace_bytes < fwd_workspace_bytes(f, d->Cout) || ((uintptr_t)workspace & 15)) {
      mmi_set_error("%s: needs a 16-byte aligned zero-initialised workspace of %zu bytes (got %zu)", who, fwd_workspace_bytes(f, d->Cout),
                    workspace_bytes);
      return MMI_ERR_WORKSPACE;
    }
    const int G = stat_group_size(f.mtiles);
    p.bn_fold = StatFold{stat_partials, (float*)((char*)workspace + WS_HEADER_BYTES), (int*)((char*)workspace + SK_COUNTER_BYTES),
                         f.mtiles, d->Cout, f.ntiles, G};
    p.bn_mi = bn->mean_invstd;
    p.bn_rmean = bn->running_mean; p.bn_rvar = bn->running_var;
    p.bn_nbt = bn->num_batches_tracked;
    p.bn_nnbt = bn->num_batches_tracked == nullptr ? 0 : (bn->num_batches_tracked2 != nullptr ? 2 : 1);
    p.bn_eps = bn->eps; p.bn_momentum = bn->momentum;
    p.bn_inv_rows = 1.0 / (double)rows; p.bn_unbias = rows > 1 ? (double)rows / (double)(rows - 1) : 1.0;
  }
  if (int e = launch_igemm<false>(p, f, vec, workspace, workspace_bytes, (hipStream_t)stream, WS_HEADER_BYTES + bn_l1_bytes(f, d->Cout)))
    return e;
  if (bn != nullptr && !fold) {  // (a list too long for the counter block, or counters that are not adjacent: separate fold)
    if (int e = mmi_bn_finalize(stat_partials, f.mtiles, rows, d->Cout, bn->eps, bn->momentum, bn->running_mean, bn->running_var,
                                bn->num_batches_tracked, bn->mean_invstd, stream)) return e;
    if (bn->num_batches_tracked2 != nullptr) {   // (rare path: the second counter takes a launch of its own)
      if (int e = mmi_i64_increment(bn->num_batches_tracked2, stream)) return e;
    }
  }
  return MMI_OK;
}
}  // namespace

extern "C" int mmi_conv_bn_fwd(const float* x, const float* w, float* y, float* stat_partials, const mmi_bn_stats* bn,
                               void* workspace, size_t workspace_bytes, const mmi_conv_desc* d, void* stream) {
  MMI_CHECK_ARG(bn != nullptr && stat_partials != nullptr, "mmi_conv_bn_fwd: null BN arguments");
  return conv_fwd_impl(x, w, nullptr, nullptr, 0, MMI_ACT_NONE, y, stat_partials, workspace, workspace_bytes, d, stream,
                       "mmi_conv_bn_fwd", bn);
}

namespace {
int fill_epilogue(IgemmP& p, const mmi_conv_desc* d, const mmi_linear_epilogue* e, bool dgrad, const float* out, const char* who) {
  MMI_CHECK_ARG(d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0, "%s: 1x1 descriptors only", who);
  if (e == nullptr || e->kind == MMI_EPI_NONE) return MMI_OK;
  const int cols = dgrad ? d->Cin : d->Cout;
  const bool fwd_kind = e->kind == MMI_EPI_DROPOUT_RESIDUAL || e->kind == MMI_EPI_GELU;
  const bool bwd_kind = e->kind == MMI_EPI_GELU_GRAD || e->kind == MMI_EPI_ACCUMULATE;
  MMI_CHECK_ARG(dgrad ? bwd_kind : fwd_kind, "%s: epilogue kind %d does not belong to this direction", who, e->kind);
  if (e->kind == MMI_EPI_GELU) {
    MMI_CHECK_ARG(e->aux_out != nullptr && e->ldaux_out >= cols, "%s: GELU epilogue needs aux_out with ldaux_out >= %d", who, cols);
  } else {
    MMI_CHECK_ARG(e->aux != nullptr && e->ldaux >= cols, "%s: epilogue needs aux with ldaux >= %d", who, cols);
    // only the accumulate form may read what it writes (element by element, same thread)
    MMI_CHECK_ARG(e->kind == MMI_EPI_ACCUMULATE || e->aux != out, "%s: aux aliases the output", who);
  }
  if (e->kind == MMI_EPI_DROPOUT_RESIDUAL)
    MMI_CHECK_ARG(e->p_drop >= 0.f && e->p_drop < 1.f, "%s: dropout probability %g", who, (double)e->p_drop);
  p.epi = e->kind; p.aux = e->aux; p.ldaux = e->ldaux; p.aux_out = e->aux_out; p.ldaux_out = e->ldaux_out;
  p.seed = e->seed; p.seed_dev = e->seed_dev;
  p.drop_thresh = e->kind == MMI_EPI_DROPOUT_RESIDUAL ? drop_thresh(e->p_drop) : 0u;
  p.inv_keep = e->kind == MMI_EPI_DROPOUT_RESIDUAL ? 1.0f / (1.0f - e->p_drop) : 1.0f;
  return MMI_OK;
}
}  // namespace

extern "C" int mmi_linear_fwd_fused(const float* x, const float* w, const float* bias, float* y, void* workspace,
                                    size_t workspace_bytes, const mmi_conv_desc* d, const mmi_linear_epilogue* e,
                                    void* stream) {
  if (int err = check_desc(d, "mmi_linear_fwd_fused")) return err;
  MMI_CHECK_ARG(x && w && y, "mmi_linear_fwd_fused: null pointer");
  const bool vec = fwd_vec(d);
  MMI_CHECK_ARG(!vec || (((uintptr_t)x | (uintptr_t)w) & 15) == 0, "mmi_linear_fwd_fused: operands must be 16-byte aligned");
  IgemmP p{};
  if (int err = fill_epilogue(p, d, e, false, y, "mmi_linear_fwd_fused")) return err;
  p.A = x; p.B = w; p.C = y; p.bias = bias;
  p.M = d->N * d->Ho * d->Wo; p.Ncol = d->Cout; p.Kc = d->Cin; p.KH = 1; p.KW = 1;
  p.P = d->Ho; p.Q = d->Wo; p.Hs = d->H; p.Ws = d->W; p.lda = d->ldx; p.ldc = d->ldy;
  p.stride = 1; p.pad = 0; p.Ktot = d->Cin; p.ldb = p.Ktot;
  if (p.epi == MMI_EPI_NONE)  // a plain Linear: the convolution instantiation (buffer-store epilogue, fewer registers)
    return launch_igemm<false, false>(p, fwd_plan(d), vec, workspace, workspace_bytes, (hipStream_t)stream);
  return launch_igemm<false, true>(p, fwd_plan(d), vec, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int mmi_linear_dgrad_fused(const float* dy, const float* w, float* dx, void* workspace, size_t workspace_bytes,
                                      const mmi_conv_desc* d, const mmi_linear_epilogue* e, void* stream) {
  if (int err = check_desc(d, "mmi_linear_dgrad_fused")) return err;
  MMI_CHECK_ARG(dy && w && dx, "mmi_linear_dgrad_fused: null pointer");
  const bool vec = dgrad_vec(d);
  MMI_CHECK_ARG(!vec || (((uintptr_t)dy | (uintptr_t)w) & 15) == 0, "mmi_linear_dgrad_fused: operands must be 16-byte aligned");
  IgemmP p{};
  if (int err = fill_epilogue(p, d, e, true, dx, "mmi_linear_dgrad_fused")) return err;
  p.A = dy; p.B = w; p.C = dx;
  p.M = d->N * d->H * d->W; p.Ncol = d->Cin; p.Kc = d->Cout; p.KH = 1; p.KW = 1;
  p.P = d->H; p.Q = d->W; p.Hs = d->Ho; p.Ws = d->Wo; p.lda = d->ldy; p.ldc = d->ldx;
  p.stride = 1; p.pad = 0; p.Ktot = d->Cout; p.ldb = d->Cin;
  if (p.epi == MMI_EPI_NONE)
    return launch_igemm<true, false>(p, dgrad_plan(d), vec, workspace, workspace_bytes, (hipStream_t)stream);
  return launch_igemm<true, true>(p, dgrad_plan(d), vec, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int mmi_conv_fwd(const float* x, const float* w, const float* bias, float* y, float* stat_partials,
                            void* workspace, size_t workspace_bytes, const mmi_conv_desc* d, void* stream) {
  return conv_fwd_impl(x, w, bias, nullptr, 0, MMI_ACT_NONE, y, stat_partials, workspace, workspace_bytes, d, stream,
                       "mmi_conv_fwd");
}

extern "C" int mmi_conv_bias_act_fwd(const float* x, const float* w, const float* bias, const float* residual, int ldr,
                                     int act, float* y, void* workspace, size_t workspace_bytes, const mmi_conv_desc* d,
                                     void* stream) {
  MMI_CHECK_ARG(act == MMI_ACT_NONE || act == MMI_ACT_SILU || act == MMI_ACT_LEAKY, "mmi_conv_bias_act_fwd: unknown activation %d", act);
  MMI_CHECK_ARG(residual == nullptr || (d && ldr >= d->Cout), "mmi_conv_bias_act_fwd: residual row stride < Cout");
  return conv_fwd_impl(x, w, bias, residual, ldr, act, y, nullptr, workspace, workspace_bytes, d, stream,
                       "mmi_conv_bias_act_fwd");
}

extern "C" int mmi_conv_dgrad(const float* dy, const float* w, float* dx, void* workspace, size_t workspace_bytes,
                              const mmi_conv_desc* d, void* stream) {
  if (int e = check_desc(d, "mmi_conv_dgrad")) return e;
  MMI_CHECK_ARG(dy && w && dx, "mmi_conv_dgrad: null pointer");
  if (mmi_smallconv_dgrad_supported(d)) return mmi_smallconv_dgrad(dy, w, dx, d, (hipStream_t)stream);
  // A = dy (channels Cout), output columns = Cin
  const bool vec = dgrad_vec(d);
  MMI_CHECK_ARG(!vec || (((uintptr_t)dy | (uintptr_t)w) & 15) == 0, "mmi_conv_dgrad: operands must be 16-byte aligned");
  IgemmP p{};
  p.A = dy; p.B = w; p.C = dx; p.bias = nullptr; p.stat_part = nullptr;
  p.M = d->N * d->H * d->W; p.Ncol = d->Cin; p.Kc = d->Cout; p.KH = d->KH; p.KW = d->KW;
  p.P = d->H; p.Q = d->W; p.Hs = d->Ho; p.Ws = d->Wo; p.lda = d->ldy; p.ldc = d->ldx;
  p.stride = d->stride; p.pad = d->pad; p.Ktot = d->KH * d->KW * d->Cout; p.ldb = d->KH * d->KW * d->Cin;
  p.par = dgrad_par(d) ? 1 : 0;
  return launch_igemm<true>(p, dgrad_plan(d), vec, workspace, workspace_bytes, (hipStream_t)stream);
}

namespace {
struct WgPlan {
  int bm, bn, mtiles, ntiles, splits, chunk;
  bool vec;
};
// resident workgroups of a wgrad variant on the whole chip (registers / LDS decide: 3 per CU for 128x128, 8 for 64x64)
int wgrad_slots(int bm, int bn, bool vec) {
  static int cache[5] = {0, 0, 0, 0, 0};
  const int idx = !vec ? 0 : (bm == 128 ? (bn == 128 ? 1 : 2) : (bn == 128 ? 3 : 4));
  if (cache[idx] == 0) {
    int n = 0;
    hipError_t e;
    switch (idx) {
      case 0: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wgrad_kernel<64, 64, false>, 256, 0); break;
      case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wgrad_kernel<128, 128, true>, 256, 0); break;
      case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wgrad_kernel<128, 64, true>, 256, 0); break;
      case 3: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wgrad_kernel<64, 128, true>, 256, 0); break;
      default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wgrad_kernel<64, 64, true>, 256, 0); break;
    }
    static const int guess[5] = {5, 3, 5, 5, 8};
    cache[idx] = (e == hipSuccess && n > 0) ? n : guess[idx];
    if (idx == 0 || idx == 4) cache[idx] = 2;  // 64x64: long K chunks stream better than many short ones (measured)
    (void)hipGetLastError();
  }
  // (the three-term split variants hold 1.5x the LDS and more registers: two workgroups per CU for the wide tiles)
  const int per_cu = ((g_gemm_prec == 2 || g_gemm_prec == 3) && idx >= 1 && idx <= 3 && cache[idx] > 2) ? 2 : cache[idx];
  return per_cu * device_cus();
}

WgPlan wgrad_plan(const mmi_conv_desc* d) {
  WgPlan g;
  const int Ntot = d->KH * d->KW * d->Cin;
  const int64_t Mpix = (int64_t)d->N * d->Ho * d->Wo;
  g.vec = d->Cin % 4 == 0 && d->Cout % 4 == 0 && d->ldx % 4 == 0 && d->ldy % 4 == 0;
  g.bm = d->Cout > 64 ? 128 : 64;
  g.bn = Ntot > 64 ? 128 : 64;
  if (!g.vec) g.bm = g.bn = 64;
  g.mtiles = cdiv(d->Cout, g.bm);
  g.ntiles = cdiv(Ntot, g.bn);
  int tiles = g.mtiles * g.ntiles;
  // Split K (pixels) so that tiles*splits fills whole waves of the resident workgroups of the variant (wgrad_slots): a grid of 1.5 waves
  // wastes a quarter of the chip.  Fewer splits win ties (less slab traffic).
  int max_splits = (int)((Mpix + 511) / 512);                       // >= 512 pixels (16 K-steps) per split
  if (g.vec && (int64_t)tiles * max_splits < 256) {
    // a launch-bound GEMM (the token projections: 2048 rows x 128..512 channels): 64x64 tiles and K chunks of 128 pixels
    // put ~10x more workgroups on the chip; the slab traffic is kept below 8 MB
    g.bm = g.bn = 64;
    g.mtiles = cdiv(d->Cout, 64);
    g.ntiles = cdiv(Ntot, 64);
    tiles = g.mtiles * g.ntiles;
    max_splits = (int)((Mpix + 127) / 128);
    const int64_t by_bytes = (int64_t)(8 << 20) / ((int64_t)d->Cout * Ntot * 4 + 1);
    if (max_splits > by_bytes) max_splits = (int)by_bytes;
  }
  const int slots = wgrad_slots(g.bm, g.bn, g.vec);
  int cap = tiles > 64 ? 16 : cdiv(2 * slots, tiles);
  if (cap > max_splits) cap = max_splits;
  if (cap < 1) cap = 1;
  // Every split costs a slab of dw to write and to read back: a split count is charged `pen` of wave efficiency per split
  // (MMIDET_WGRAD_SPLIT_PENALTY, default 0: the round-1 rule -- fill whole waves, fewer splits win ties)
  static const double pen = getenv("MMIDET_WGRAD_SPLIT_PENALTY") ? atof(getenv("MMIDET_WGRAD_SPLIT_PENALTY")) : 0.0;
  int splits = 1;
  double best = -1e9;
  for (int sp = 1; sp <= cap; ++sp) {
    const int blocks = tiles * sp;
    const double eff = (double)blocks / (double)(cdiv(blocks, slots) * slots) - pen * sp;
    if (eff > best + 1e-9) best = eff, splits = sp;
  }
  // Short-K GEMMs (the token projections: 2048 rows, i.e. at most 64 K-steps): tools/sweep_wgrad.py,
  // profiles/r02_sweep_wgrad.txt.  What wins there is enough workgroups WITHOUT leaving the in-launch fold (<= 4 splits): the
  // largest tile variant that gives >= 512 tiles unsplit (1024 -> 4096: 64x128, 1.21x over 128x128 x 3 splits), else 64x64
  // tiles with up to four splits (1024 -> 1024: 1.30x, 512 <-> 2048: 1.33x).
  // OFF by default: stand-alone the two rules below take 6 % off the swept shapes (profiles/r02_sweep_wgrad.txt: 30.95 -> 28.0 ms
  // summed over a step), inside the step -- where every wgrad shares the chip with the lane's dgrad -- they cost 0.6 ms
  // (profiles/r02_ab_wgrad_rules.txt, three interleaved pairs): more, smaller workgroups interfere more with the co-runner.
  static const bool sweep_rules = getenv("MMIDET_WGRAD_RULES") && atoi(getenv("MMIDET_WGRAD_RULES")) == 1;   // (A/B switch)
  if (sweep_rules && g.vec && Mpix <= 4096 && cdiv(d->Cout, 64) * cdiv(Ntot, 64) >= 256 && g_wgrad_force[2] == 0) {
    static const int cand[3][2] = {{128, 128}, {64, 128}, {64, 64}};
    int pick = 2;
    for (int c = 0; c < 3; ++c)
      if ((int64_t)cdiv(d->Cout, cand[c][0]) * cdiv(Ntot, cand[c][1]) >= 512) {
        pick = c;
        break;
      }
    g.bm = cand[pick][0];
    g.bn = cand[pick][1];
    g.mtiles = cdiv(d->Cout, g.bm);
    g.ntiles = cdiv(Ntot, g.bn);
    tiles = g.mtiles * g.ntiles;
    splits = tiles >= 512 ? 1 : min(4, cdiv(1024, tiles));
    if (splits > max_splits) splits = max_splits < 1 ? 1 : max_splits;
  }
  // 1x1 convolutions over many pixels (the same sweep): the output is a few tiles and everything is split-K; 64x64 tiles put
  // 512..1024 workgroups on the chip with a third to a tenth of the splits -- i.e. of the slab traffic -- of one or two wide
  // tiles (128 -> 64 @160x160: 1.45x, 256 -> 128 @80x80: 1.25x, 512 -> 256 @40x40 and 1024 -> 512 @20x20: 1.22x).
  if (sweep_rules && g.vec && d->KH * d->KW == 1 && Mpix > 4096 && g_wgrad_force[2] == 0) {
    const int t64 = cdiv(d->Cout, 64) * cdiv(Ntot, 64);
    if (t64 >= 8 || d->Cout <= 64) {
      g.bm = g.bn = 64;
      g.mtiles = cdiv(d->Cout, 64);
      g.ntiles = cdiv(Ntot, 64);
      tiles = t64;
      const int by_pixels = (int)((Mpix + 511) / 512);
      if (t64 >= 8) {                    // 512..1024 workgroups, about 800 pixels (25 K-steps) per split where that fits
        const int lo = cdiv(512, t64), hi = cdiv(1024, t64), want = (int)(Mpix / 800);
        splits = want < lo ? lo : (want > hi ? hi : want);
      } else {
        splits = cdiv(512, t64);
      }
      if (splits > by_pixels) splits = by_pixels;
      if (splits < 1) splits = 1;
    }
  }
  if (g_wgrad_force[2] > 0) {       // mmi_set_wgrad_override (tuning, tools/sweep_wgrad.py): force tile variant and split count
    if (g.vec && g_wgrad_force[0] > 0) {
      g.bm = g_wgrad_force[0];
      g.bn = g_wgrad_force[1];
      g.mtiles = cdiv(d->Cout, g.bm);
      g.ntiles = cdiv(Ntot, g.bn);
    }
    splits = g_wgrad_force[2];
    if (splits > (int)((Mpix + BK - 1) / BK)) splits = (int)((Mpix + BK - 1) / BK);
  }
  g.chunk = cdiv(cdiv(Mpix, splits), BK) * BK;
  g.splits = cdiv(Mpix, g.chunk);
  return g;
}
}  // namespace

// Workspace of a wgrad launch: [0, WG_COUNTER_BYTES) per-tile arrival counters of the in-launch split-K fold (zero-filled
// when first handed over, self-cleaning afterwards), then the splits' partial slabs (no zero-fill needed).
constexpr int WG_MAX_TILES = 4096;
constexpr size_t WG_COUNTER_BYTES = (size_t)WG_MAX_TILES * sizeof(int);
extern "C" size_t mmi_conv_wgrad_workspace(const mmi_conv_desc* d) {
  if (check_desc(d, "mmi_conv_wgrad_workspace") != MMI_OK) return 0;
  const WgPlan g = wgrad_plan(d);
  const size_t generic = g.splits > 1 ? (size_t)g.splits * ((size_t)d->Cout * d->KH * d->KW * d->Cin + d->Cout) * sizeof(float) : 0;
  const size_t small = mmi_smallconv_supported(d) ? mmi_smallconv_wgrad_workspace(d) : 0;
  const size_t body = generic > small ? generic : small;
  return body ? WG_COUNTER_BYTES + body : 0;
}

namespace {
// does this shape run the pixel-table loaders with a table (not the 1x1 stride-1 layers, whose x rows need none)?
bool wgrad_uses_table(const mmi_conv_desc* d) {
  if (mmi_smallconv_supported(d)) return false;
  const WgPlan g = wgrad_plan(d);
  if (!(g.vec && g_uniform_loaders && g_gemm_prec == 0 && d->KH * d->KW <= 32)) return false;
  if (d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0) return false;
  const int64_t margin = ((int64_t)d->KH * d->W + d->KW) * d->ldx;
  return (margin + ((int64_t)d->N * d->H * d->W - 1) * d->ldx + d->Cin) * 4 < (1LL << 31);
}
int64_t wgrad_table_entries(const mmi_conv_desc* d) { return (int64_t)d->N * d->Ho * d->Wo + 4 * BK; }
}  // namespace

extern "C" size_t mmi_conv_wgrad_table_bytes(const mmi_conv_desc* d) {
  if (check_desc(d, "mmi_conv_wgrad_table_bytes") != MMI_OK || !wgrad_uses_table(d)) return 0;
  return (size_t)wgrad_table_entries(d) * sizeof(uint2);
}

extern "C" int mmi_conv_wgrad_table_build(void* table, const mmi_conv_desc* d, void* stream) {
  if (int e = check_desc(d, "mmi_conv_wgrad_table_build")) return e;
  MMI_CHECK_ARG(table != nullptr && ((uintptr_t)table & 7) == 0, "mmi_conv_wgrad_table_build: null or misaligned table");
  const int total = (int)wgrad_table_entries(d);
  hipLaunchKernelGGL(wgrad_table_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (uint2*)table,
                     d->N * d->Ho * d->Wo, total, d->Ho, d->Wo, d->H, d->W, d->KH, d->KW, d->stride, d->pad, d->ldx);
  MMI_CHECK_LAUNCH("mmi_conv_wgrad_table_build");
  return MMI_OK;
}

namespace {
int conv_wgrad_impl(const float* dy, const float* x, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                    const void* table, const mmi_conv_desc* d, void* stream, bool bf16_io = false);
}
extern "C" int mmi_conv_wgrad(const float* dy, const float* x, float* dw, float* dbias, void* workspace,
                              size_t workspace_bytes, const mmi_conv_desc* d, void* stream) {
  return conv_wgrad_impl(dy, x, dw, dbias, workspace, workspace_bytes, nullptr, d, stream);
}
extern "C" int mmi_conv_wgrad_tab(const float* dy, const float* x, float* dw, float* dbias, void* workspace,
                                  size_t workspace_bytes, const void* table, const mmi_conv_desc* d, void* stream) {
  return conv_wgrad_impl(dy, x, dw, dbias, workspace, workspace_bytes, table, d, stream);
}

namespace {
int conv_wgrad_impl(const float* dy, const float* x, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                    const void* table, const mmi_conv_desc* d, void* stream, bool bf16_io) {
  if (int e = check_desc(d, "mmi_conv_wgrad")) return e;
  MMI_CHECK_ARG(dy && x && dw, "mmi_conv_wgrad: null pointer");
  if (mmi_smallconv_supported(d) && dbias == nullptr && !bf16_io) {
    if (workspace == nullptr || workspace_bytes < WG_COUNTER_BYTES + mmi_smallconv_wgrad_workspace(d)) {
      mmi_set_error("mmi_conv_wgrad: workspace too small (%zu < %zu)", workspace_bytes, WG_COUNTER_BYTES + mmi_smallconv_wgrad_workspace(d));
      return MMI_ERR_WORKSPACE;
    }
    return mmi_smallconv_wgrad(dy, x, dw, (char*)workspace + WG_COUNTER_BYTES, d, (hipStream_t)stream);   // (never the counter block)
  }
  const WgPlan g = wgrad_plan(d);
  MMI_CHECK_ARG(!bf16_io || g.vec, "mmi_conv_wgrad_bf16: channel counts and row strides must be multiples of 4");
  MMI_CHECK_ARG(!g.vec || (((uintptr_t)dy | (uintptr_t)x) & (bf16_io ? 7 : 15)) == 0, "mmi_conv_wgrad: operands must be 16-byte aligned");
  const int64_t wsize = (int64_t)d->Cout * d->KH * d->KW * d->Cin;
  const int64_t slab = wsize + d->Cout;  // weight gradient + bias-gradient tail
  if (g.splits > 1 && (workspace == nullptr || workspace_bytes < WG_COUNTER_BYTES + (size_t)g.splits * slab * sizeof(float) ||
                       ((uintptr_t)workspace & 15))) {
    mmi_set_error("mmi_conv_wgrad: workspace too small or misaligned (%zu < %zu)", workspace_bytes,
                  WG_COUNTER_BYTES + (size_t)g.splits * slab * sizeof(float));
    return MMI_ERR_WORKSPACE;
  }
  float* slabs = g.splits > 1 ? (float*)((char*)workspace + WG_COUNTER_BYTES) : nullptr;
  static const bool fold_off = getenv("MMIDET_WGRAD_FOLD") != nullptr && atoi(getenv("MMIDET_WGRAD_FOLD")) == 0;  // (A/B switch)
  // The fold runs on ONE workgroup per tile, serially over the splits (a dependent round of loads per four of them), while
  // the reduce kernel spreads the same reads over the whole chip: measured (profiles/r02_wgrad_fold_microbench.txt) the fold
  // only wins up to a handful of splits, so long split lists keep the separate reduce launch.
  // Measured twice (profiles/r02_wgrad_fold_microbench.txt: one workgroup walking all splits; profiles/r02_ab_wgrad_fold_tree.txt:
  // the fan-in-4 tree of the kernel's epilogue): the in-launch fold wins up to FOUR splits (one level of the tree) and loses
  // beyond -- every level is a dependent round of device-coherent loads of slabs written on other XCDs (~6 us), against one
  // chip-wide reduce launch that streams them: 3x3 128->128@80x80 0.266 -> 0.351 ms, the step 123.2 -> 125.3 ms with the tree
  // for every split count.  So longer split lists keep the separate reduce launch; MMIDET_WGRAD_FOLD_MAX (<= 256) moves the limit.
  static const int fold_max = getenv("MMIDET_WGRAD_FOLD_MAX") ? atoi(getenv("MMIDET_WGRAD_FOLD_MAX")) : 4;
  int cnt_per_tile = 0;
  for (int n = g.splits; n > 1; n = (n + 3) / 4) cnt_per_tile += (n + 3) / 4;
  const bool fold = g.splits > 1 && g.splits <= fold_max && (int64_t)g.mtiles * g.ntiles * cnt_per_tile <= WG_MAX_TILES && !fold_off;
  WgradP p{};
  p.DY = dy; p.X = x; p.OUT = g.splits > 1 ? slabs : dw;
  p.OUTB = dbias == nullptr ? nullptr : (g.splits > 1 ? slabs + wsize : dbias);
  p.cnt = fold ? (int*)workspace : nullptr;
  p.cnt_per_tile = cnt_per_tile;
  p.DW = dw; p.DB = dbias;
  p.zero = zero_src();
  if (p.zero == nullptr) {
    mmi_set_error("mmi_conv_wgrad: cannot resolve the zero-source symbol");
    return MMI_ERR_LAUNCH;
  }
  p.Mpix = d->N * d->Ho * d->Wo; p.Cout = d->Cout; p.Cin = d->Cin; p.KH = d->KH; p.KW = d->KW;
  p.Ho = d->Ho; p.Wo = d->Wo; p.H = d->H; p.W = d->W; p.stride = d->stride; p.pad = d->pad;
  p.ldx = d->ldx; p.ldy = d->ldy; p.Ntot = d->KH * d->KW * d->Cin; p.chunk = g.chunk;
  p.mtiles = g.mtiles; p.ntiles = g.ntiles; p.splits = g.splits; p.slab_stride = g.splits > 1 ? slab : 0;
  const dim3 grid(g.mtiles * g.ntiles, g.splits), block(256);
  hipStream_t s = (hipStream_t)stream;
  // pixel-table loaders (wgrad_kernel<..., TAB>): tap mask in 32 bits, 31-bit byte offsets into x
  bool tab = false;
  if (g.vec && g_uniform_loaders && g_gemm_prec == 0 && d->KH * d->KW <= 32 && !bf16_io) {
    const int64_t margin = ((int64_t)d->KH * d->W + d->KW) * d->ldx;
    const int64_t x_bytes = (margin + ((int64_t)d->N * d->H * d->W - 1) * d->ldx + d->Cin) * 4;
    if (x_bytes < (1LL << 31)) {
      tab = true;
      p.x_bytes = (uint32_t)x_bytes;
      p.tab = (const uint2*)table;   // (null: the kernel builds its table slab by slab)
    }
  }
#define LAUNCHW(BM_, BN_, VEC_) \
  hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, VEC_>), grid, block, 0, s, p)
  if (bf16_io) {
#define LAUNCHWB(BM_, BN_) hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, true, 4>), grid, block, 0, s, p)
    if (g.bm == 128 && g.bn == 128) LAUNCHWB(128, 128);
    else if (g.bm == 128) LAUNCHWB(128, 64);
    else if (g.bn == 128) LAUNCHWB(64, 128);
    else LAUNCHWB(64, 64);
#undef LAUNCHWB
  } else if (g.vec && g_gemm_prec >= 1) {
#define LAUNCHW3(BM_, BN_)                                                                              \
  do {                                                                                                  \
    if (g_gemm_prec == 1) hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, true, 1>), grid, block, 0, s, p);  \
    else if (g_gemm_prec == 2) hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, true, 2>), grid, block, 0, s, p); \
    else if (g_gemm_prec == 5) hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, true, 5>), grid, block, 0, s, p); \
    else hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, true, 3>), grid, block, 0, s, p);                   \
  } while (0)
    if (g.bm == 128 && g.bn == 128) LAUNCHW3(128, 128);
    else if (g.bm == 128) LAUNCHW3(128, 64);
    else if (g.bn == 128) LAUNCHW3(64, 128);
    else LAUNCHW3(64, 64);
#undef LAUNCHW3
  } else if (!g.vec) LAUNCHW(64, 64, false);
  else if (tab) {
#define LAUNCHWT(BM_, BN_) hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, true, 0, true>), grid, block, 0, s, p)
    if (g.bm == 128 && g.bn == 128) LAUNCHWT(128, 128);
    else if (g.bm == 128) LAUNCHWT(128, 64);
    else if (g.bn == 128) LAUNCHWT(64, 128);
    else LAUNCHWT(64, 64);
#undef LAUNCHWT
  } else if (g.bm == 128 && g.bn == 128) LAUNCHW(128, 128, true);
  else if (g.bm == 128) LAUNCHW(128, 64, true);
  else if (g.bn == 128) LAUNCHW(64, 128, true);
  else LAUNCHW(64, 64, true);
#undef LAUNCHW
  MMI_CHECK_LAUNCH("mmi_conv_wgrad");
  if (g.splits > 1 && !fold) {
    // without dbias only the weight part [0, wsize) of every slab is reduced
    const int64_t count = dbias != nullptr ? slab : wsize;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(count, 1024)), dim3(256), 0, s, (const float*)slabs, dw, dbias, wsize, slab, count,
                       g.splits);
    MMI_CHECK_LAUNCH("mmi_conv_wgrad(reduce)");
  }
  return MMI_OK;
}
}  // namespace

// ---- bf16 storage: activations and activation gradients are bf16 in HBM, weights / weight gradients /
// BatchNorm statistics fp32; one bf16 MFMA product per element pair with fp32 accumulation (igemm_kernel<..., PREC = 4>).
// One workgroup per tile (no stream-K: these launches are HBM-bound, not wave-quantisation-bound).
namespace {
template <bool DGRAD, bool EPI>
int launch_igemm_bf16(IgemmP p, const FwdPlan& f, hipStream_t s, const char* who) {
  p.zero = zero_src();
  if (p.zero == nullptr) {
    mmi_set_error("%s: cannot resolve the zero-source symbol", who);
    return MMI_ERR_LAUNCH;
  }
  p.mtiles = f.mtiles;
  p.ntiles = f.ntiles;
  const dim3 grid(f.mtiles * f.ntiles, p.par ? 4 : 1), block(256);
  if (f.bm == 128 && f.bn == 128) hipLaunchKernelGGL((igemm_kernel<128, 128, DGRAD, true, false, 4, EPI>), grid, block, 0, s, p);
  else if (f.bm == 128 && f.bn == 64) hipLaunchKernelGGL((igemm_kernel<128, 64, DGRAD, true, false, 4, EPI>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((igemm_kernel<64, 64, DGRAD, true, false, 4, EPI>), grid, block, 0, s, p);
  MMI_CHECK_LAUNCH(who);
  return MMI_OK;
}
FwdPlan fwd_plan_bf16(const mmi_conv_desc* d) { return plan_tiles((int64_t)d->N * d->Ho * d->Wo, d->Cout); }
FwdPlan dgrad_plan_bf16(const mmi_conv_desc* d) {
  const bool par = dgrad_par(d);
  const int64_t mrows = par ? (int64_t)d->N * ((d->H + 1) / 2) * ((d->W + 1) / 2) : (int64_t)d->N * d->H * d->W;
  return plan_tiles(mrows, d->Cin);
}
}  // namespace

extern "C" int mmi_conv_fwd_row_blocks_bf16(const mmi_conv_desc* d) {
  if (check_desc(d, "mmi_conv_fwd_row_blocks_bf16") != MMI_OK) return MMI_ERR_ARG;
  return fwd_plan_bf16(d).mtiles;
}

extern "C" size_t mmi_conv_fwd_workspace_bf16(const mmi_conv_desc* d) {
  if (check_desc(d, "mmi_conv_fwd_workspace_bf16") != MMI_OK) return 0;
  return WS_HEADER_BYTES + bn_l1_bytes(fwd_plan_bf16(d), d->Cout);
}

extern "C" int mmi_conv_fwd_bf16(const void* x, const float* w, const float* bias, void* y, float* stat_partials,
                                 const mmi_bn_stats* bn, void* workspace, size_t workspace_bytes, const mmi_conv_desc* d,
                                 void* stream) {
  const char* who = "mmi_conv_fwd_bf16";
  if (int e = check_desc(d, who)) return e;
  MMI_CHECK_ARG(x && w && y, "%s: null pointer", who);
  MMI_CHECK_ARG(!(bias && stat_partials), "%s: bias and BN statistics are mutually exclusive", who);
  MMI_CHECK_ARG(fwd_vec(d) && d->ldy % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 7) == 0 && ((uintptr_t)w & 15) == 0,
                "%s: channel counts / row strides must be multiples of 4 and the operands aligned", who);
  IgemmP p{};
  p.A = (const float*)x; p.B = w; p.C = (float*)y; p.bias = bias; p.stat_part = stat_partials;
  p.M = d->N * d->Ho * d->Wo; p.Ncol = d->Cout; p.Kc = d->Cin; p.KH = d->KH; p.KW = d->KW;
  p.P = d->Ho; p.Q = d->Wo; p.Hs = d->H; p.Ws = d->W; p.lda = d->ldx; p.ldc = d->ldy;
  p.stride = d->stride; p.pad = d->pad; p.Ktot = d->KH * d->KW * d->Cin; p.ldb = p.Ktot;
  const FwdPlan f = fwd_plan_bf16(d);
  const int64_t rows = (int64_t)d->N * d->Ho * d->Wo;
  bool fold = false;
  if (bn != nullptr) {
    MMI_CHECK_ARG(stat_partials && bn->mean_invstd, "%s: BN statistics need the partials buffer and mean_invstd", who);
    fold = bn_fold_fits(f) && (bn->num_batches_tracked2 == nullptr || bn->num_batches_tracked2 == bn->num_batches_tracked + 1);
    if (fold) {
      if (workspace == nullptr || workspace_bytes < WS_HEADER_BYTES + bn_l1_bytes(f, d->Cout) || ((uintptr_t)workspace & 15)) {
        mmi_set_error("%s: needs a 16-byte aligned zero-initialised workspace of %zu bytes (got %zu)", who,
                      WS_HEADER_BYTES + bn_l1_bytes(f, d->Cout), workspace_bytes);
        return MMI_ERR_WORKSPACE;
      }
      p.bn_fold = StatFold{stat_partials, (float*)((char*)workspace + WS_HEADER_BYTES), (int*)((char*)workspace + SK_COUNTER_BYTES),
                           f.mtiles, d->Cout, f.ntiles, stat_group_size(f.mtiles)};
      p.bn_mi = bn->mean_invstd;
      p.bn_rmean = bn->running_mean; p.bn_rvar = bn->running_var;
      p.bn_nbt = bn->num_batches_tracked;
      p.bn_nnbt = bn->num_batches_tracked == nullptr ? 0 : (bn->num_batches_tracked2 != nullptr ? 2 : 1);
      p.bn_eps = bn->eps; p.bn_momentum = bn->momentum;
      p.bn_inv_rows = 1.0 / (double)rows; p.bn_unbias = rows > 1 ? (double)rows / (double)(rows - 1) : 1.0;
    }
  }
  if (int e = launch_igemm_bf16<false, false>(p, f, (hipStream_t)stream, who)) return e;
  if (bn != nullptr && !fold) {
    if (int e = mmi_bn_finalize(stat_partials, f.mtiles, rows, d->Cout, bn->eps, bn->momentum, bn->running_mean, bn->running_var,
                                bn->num_batches_tracked, bn->mean_invstd, stream)) return e;
    if (bn->num_batches_tracked2 != nullptr)
      if (int e = mmi_i64_increment(bn->num_batches_tracked2, stream)) return e;
  }
  return MMI_OK;
}

// dx = conv_transpose(dy, w) [+ skip]: dy, dx, skip bf16 (skip: 1x1 stride-1 layers only, row stride ldskip; may be NULL)
extern "C" int mmi_conv_dgrad_bf16(const void* dy, const float* w, void* dx, const void* skip, int ldskip, const mmi_conv_desc* d,
                                   void* stream) {
  const char* who = "mmi_conv_dgrad_bf16";
  if (int e = check_desc(d, who)) return e;
  MMI_CHECK_ARG(dy && w && dx, "%s: null pointer", who);
  MMI_CHECK_ARG(dgrad_vec(d) && d->ldx % 4 == 0 && (((uintptr_t)dy | (uintptr_t)dx) & 7) == 0 && ((uintptr_t)w & 15) == 0,
                "%s: channel counts / row strides must be multiples of 4 and the operands aligned", who);
  IgemmP p{};
  p.A = (const float*)dy; p.B = w; p.C = (float*)dx;
  p.M = d->N * d->H * d->W; p.Ncol = d->Cin; p.Kc = d->Cout; p.KH = d->KH; p.KW = d->KW;
  p.P = d->H; p.Q = d->W; p.Hs = d->Ho; p.Ws = d->Wo; p.lda = d->ldy; p.ldc = d->ldx;
  p.stride = d->stride; p.pad = d->pad; p.Ktot = d->KH * d->KW * d->Cout; p.ldb = d->KH * d->KW * d->Cin;
  p.par = dgrad_par(d) ? 1 : 0;
  if (skip != nullptr) {
    MMI_CHECK_ARG(d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && ldskip >= d->Cin && skip != dx,
                  "%s: the skip accumulation exists for 1x1 stride-1 layers", who);
    p.epi = MMI_EPI_ACCUMULATE; p.aux = (const float*)skip; p.ldaux = ldskip;
    return launch_igemm_bf16<true, true>(p, dgrad_plan_bf16(d), (hipStream_t)stream, who);
  }
  return launch_igemm_bf16<true, false>(p, dgrad_plan_bf16(d), (hipStream_t)stream, who);
}

extern "C" int mmi_conv_wgrad_bf16(const void* dy, const void* x, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                                   const mmi_conv_desc* d, void* stream) {
  return conv_wgrad_impl((const float*)dy, (const float*)x, dw, dbias, workspace, workspace_bytes, nullptr, d, stream, true);
}

