// The forward / data-gradient implicit-GEMM kernel template and its device helpers (see igemm.hip for the design notes).
#pragma once
#include <type_traits>

#include "igemm_defs.h"

namespace mmi_ig {
namespace {

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

// x -> NP bf16 terms, each the rounded residual of the previous ones.  Written on element PAIRS: one v_cvt_pk_bf16_f32 per pair and
// term, the residual as a packed subtract (18 VALU per float4 and three terms; element by element the compiler spent 24).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
template <int NP>
__device__ __forceinline__ void split_bf16(const f32x4& v, bf16x4 (&t)[NP]) {
  f32x2 r0 = {v[0], v[1]}, r1 = {v[2], v[3]};
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const bf16x2 h0 = __builtin_convertvector(r0, bf16x2), h1 = __builtin_convertvector(r1, bf16x2);
    t[k][0] = h0[0]; t[k][1] = h0[1]; t[k][2] = h1[0]; t[k][3] = h1[1];
    if (k + 1 < NP) {
      r0 -= __builtin_convertvector(h0, f32x2);
      r1 -= __builtin_convertvector(h1, f32x2);
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
// Order of the K slabs of a 3x3 layer under the uniform-tap loaders: MMI_KORD = 1 (default) = channel-slab major -- the nine taps
// of channels 0..31, then of 32..63, ... -- instead of tap major (all channel slabs of tap 0, then tap 1, ...: the general loaders'
// order, MMI_KORD = 0).  A workgroup's loads of one channel slab re-touch the same cache lines of its three-row input window nine
// times in a row, so what its XCD's L2 must hold while the window is being re-read is one slab deep (1/4 of the window at 128
// channels, 1/16 at 512) and not all channels.  Stand-alone, forward +5..8 % on every 3x3 layer (64 channels at 160x160: +24 %),
// stride-1 dgrad +3 %, the stride-2 parity classes -5 % (six launches per step); in the step -1.2 % on a box whose memory side is
// slow (133.6 -> 132.0 ms) and -0.2 % on a fast one (118.58 -> 118.36): profiles/r04_k_order_ab.txt.  Same products, another
// summation order than the general loaders (which these loaders used to equal bit for bit); 1x1 layers have one tap and do not
// change.  A compile-time constant (an A/B build compiles the igemm_fwd* / igemm_dgrad* units with -DMMI_KORD=0 and is loaded through MMIDET_HIP_LIB).
#ifndef MMI_KORD
#define MMI_KORD 1
#endif
#ifndef MMI_UNI_OCC
#define MMI_UNI_OCC 3
#endif
// W41 (narrow outputs: Focus' input gradient has N = 12): the four waves are stacked along M, each owning 32 rows x the whole
// tile width, and skip the 32-column blocks beyond the last output column -- in the 2 x 2 layout half of the waves would own
// nothing but padding and leave their SIMDs' matrix pipes idle.
// T8 (pre-split operands, PREC = 2 / 3 with the uniform-tap loaders): bit 0 = the weight operand B, bit 1 = the activation operand A
// arrive as their three bf16 terms, split ONCE by whoever produced them (t8.hip: the optimizer for the weights, the BatchNorm
// passes for activations and their gradients) instead of by every workgroup that stages a tile of them -- a weight tile is
// staged by every row tile of the launch, an activation row by every column tile and tap.  Format "T8": per 8 consecutive
// channels 48 bytes = [term 0: 8 bf16 | term 1 | term 2] (6 bytes per element, rows keep their element stride).  A loader thread
// fetches one such group (three 16-byte loads) and stores each term to its plane of the LDS row record: no v_cvt, no subtract;
// the MFMA side does not change, and the terms are the same bits the in-kernel split produces, so results are bit-identical.
// PF2 (deep prefetch): the LDS tile is double-buffered and the global loads run TWO slabs ahead in two register sets -- slab s + 3
// is requested at the top of iteration s, right after slab s + 1 (requested two iterations earlier) has gone from its registers
// to the idle LDS buffer -- so a load has two whole MFMA phases to arrive instead of a fraction of one, and an iteration has one
// barrier instead of two.  Costs a second LDS stage and a second register set: two workgroups per CU.
template <int BM, int BN, bool DGRAD, bool VEC, bool SK, int PREC = 0, bool EPI = false, bool UNI = false, bool W41 = false, int T8 = 0, bool PF2 = false>
__global__ __launch_bounds__(256, (!PF2 && MMI_IGEMM_STAGES == 1 && BK == 32 && (PREC < 2 || PREC >= 4)) ? (UNI ? MMI_UNI_OCC : 3) : 2) void igemm_kernel(IgemmP p, IgemmDelta dz) {
  // twin launches (two problems of one shape, e.g. the RGB and IR backbone layers of the two-stream model): blockIdx.z picks the
  // problem; problem 1 moves its copy of the operand pointers (igemm_defs.h::IgemmDelta), nothing per lane
  if (blockIdx.z != 0) {  // uniform
    p.A = shift_ptr(p.A, dz.A); p.B = shift_ptr(p.B, dz.B); p.C = shift_ptr(p.C, dz.C);
    p.stat_part = shift_ptr(p.stat_part, dz.stat_part);
    p.sk_slots = shift_ptr(p.sk_slots, dz.sk_slots); p.sk_count = shift_ptr(p.sk_count, dz.sk_count);
    p.aux = shift_ptr(p.aux, dz.aux); p.aux_out = shift_ptr(p.aux_out, dz.aux_out);
    p.bn_fold.part = shift_ptr(p.bn_fold.part, dz.fold_part); p.bn_fold.l1 = shift_ptr(p.bn_fold.l1, dz.fold_l1);
    p.bn_fold.cnt = shift_ptr(p.bn_fold.cnt, dz.fold_cnt);
    p.bn_mi = shift_ptr(p.bn_mi, dz.bn_mi); p.bn_rmean = shift_ptr(p.bn_rmean, dz.bn_rmean); p.bn_rvar = shift_ptr(p.bn_rvar, dz.bn_rvar);
    p.bn_nbt = shift_ptr(p.bn_nbt, dz.bn_nbt);
    p.A8 = shift_ptr(p.A8, dz.A8); p.B8 = shift_ptr(p.B8, dz.B8);
    if constexpr (DGRAD) {
      p.bnr_y = shift_ptr(p.bnr_y, dz.bnr_y); p.bnr_g = shift_ptr(p.bnr_g, dz.bnr_g); p.bnr_b = shift_ptr(p.bnr_b, dz.bnr_b);
    }
  }
  static_assert(!PF2 || (VEC && UNI), "deep prefetch: a form of the uniform-tap loaders");
  constexpr int NSTG = PF2 ? 2 : MMI_IGEMM_STAGES, NS = PF2 ? 2 : 1;   // LDS stages, register sets of staged slabs
  static_assert(T8 == 0 || (UNI && (PREC == 2 || PREC == 3) && BK == 32), "pre-split operands: three-term modes, uniform-tap loaders");
  constexpr bool A8 = (T8 & 2) != 0, B8 = (T8 & 1) != 0;
  static_assert(!W41 || (DGRAD && !SK && PREC == 0 && !EPI && BM == 128), "the stacked wave layout exists for the plain fp32 dgrad tiles");
  static_assert(PREC == 0 || (VEC && BK == 32), "the split-bf16 forms exist for the vector loaders only");
  static_assert(!UNI || VEC, "uniform-tap loaders are a form of the vector loaders");
  // PREC = 4 (bf16 STORAGE): the activation operand A and the output C live in HBM as bf16 (the weights stay
  // fp32 master copies, rounded when a tile is staged), one bf16 MFMA product per element pair, fp32 accumulation, fp32
  // BatchNorm statistics taken from the accumulators.  Same tile machinery as the split forms with a single plane.
  constexpr bool BF = PREC == 4;
  constexpr int ES = BF ? 2 : 4;   // bytes per element of the activation operand A (UNI: byte offsets against a buffer resource)
  // PREC = 5 ("bf16x1"): fp32 operands in HBM, each rounded to ONE bf16 term when staged, one bf16 MFMA product: the arithmetic of
  // the bf16-storage mode for the GEMMs whose operands stay fp32 (the token-side Linear layers, Focus, Detect)
  constexpr bool ONE = BF || PREC == 5;
  constexpr int NP = PREC == 0 || ONE ? 1 : (PREC == 3 ? 3 : PREC + 1);      // bf16 planes per operand
  constexpr int OL = ONE ? 0 : (PREC == 3 ? 2 * (NP - 1) : NP - 1);          // highest total order of the products kept
  // floats per [row][k] LDS record: fp32 32 + 4 pad; split forms NP x 64 B of bf16 + 16 B pad (20, 36 or 52 floats: each makes
  // the ds_read_b128 of 8 consecutive rows hit 8 different 16-byte bank groups)
  constexpr int RSF = ONE ? 20 : (PREC >= 2 ? 52 : LDS_PAD);
  constexpr int WM = W41 ? BM / 4 : BM / 2, WN = W41 ? BN : BN / 2, TM = WM / 32, TN = WN / 32;
  // loader geometry per operand: fp32 = 8 threads per tile row with 4 k each (KT, RPP); pre-split = 4 threads with one 8-k group each
  constexpr int KTA = A8 ? BK / 8 : KT, RPPA = 256 / KTA, KEA = A8 ? 8 : 4;
  constexpr int KTB = B8 ? BK / 8 : KT, RPPB = 256 / KTB, KEB = B8 ? 8 : 4;
  constexpr int ESA = A8 ? 6 : (BF ? 2 : 4), ESB = B8 ? 6 : 4;   // bytes per element (byte offsets against the buffer resources)
  constexpr int RA = BM / RPPA;                     // A rows per loader thread
  constexpr int A_ELEMS = BM * RSF;
  // split-bf16 dgrad: the weight tile stays k-major ([k][n], as it comes from OHWI memory) in two bf16 planes whose rows are
  // padded by 64 B (conflict-free ds_read_b64_tr_b16: the MFMA B operand is fetched with the hardware transpose read)
  constexpr int B_RSB = BN * 2 + 64;                                  // bytes per k row of one plane
  constexpr int B_ELEMS = DGRAD ? (PREC >= 1 ? NP * BK * B_RSB / 4 : BK * BN) : BN * RSF;
  constexpr int STAGE = A_ELEMS + B_ELEMS;
  constexpr int RB = BN / RPPB;                     // fwd: B rows per loader thread
  constexpr int VPR = BN / KEB, RPI = 256 / VPR, KB_IT = BK / RPI;  // dgrad B loader geometry (KEB columns per thread)
  __shared__ __align__(16) float smem[NSTG * STAGE];
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
  const int kq = (t % KTA) * KEA;  // this thread's k offset inside a slab ([row][k] tiles), A operand
  const int lrow = t / KTA;        // 0..RPPA-1
  const int kqb = (t % KTB) * KEB, lrowb = t / KTB;   // the same for the forward B operand
  const int l31 = lane & 31, lh = lane >> 5;
  __amdgpu_buffer_rsrc_t srd_a, srd_b;
  if constexpr (UNI) {
    const int64_t margin = ((int64_t)p.KH * p.Ws + p.KW) * p.lda;  // elements in front of A that row offsets may reach into
    srd_a = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(A8 ? p.A8 : (const void*)p.A) - margin * ESA), 0, (int)p.a_bytes, 0x00020000);
    srd_b = __builtin_amdgcn_make_buffer_rsrc(B8 ? (void*)p.B8 : (void*)p.B, 0, (int)p.b_bytes, 0x00020000);
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
      const int m = m0 + lrow + RPPA * i;
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
      if (par && (t % KTA) == 0) rowmap[lrow + RPPA * i] = orow;  // visible after the K loop's barriers
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[NS][RA];
    bf16x4 rab[NS][BF ? RA : 1];  // bf16 storage: the A operand arrives as 4 bf16 per load
    constexpr int NB = DGRAD ? KB_IT : RB;
    f32x4 rb[NS][NB];
    bf16x8 ra8[NS][A8 ? RA : 1][3], rb8[NS][B8 ? NB : 1][3];   // pre-split operands: one 8-k group = three terms of 8 bf16

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
            aoff[i] = (uint32_t)((((int64_t)(m0 + lrow + RPPA * i) + p.Ws + 1) * p.lda + kq) * ESA);  // margin = KH*Ws + KW pixels
            amask[i] = 0u;
          }
        } else if (rows[i].base >= 0) {
          const int ihb = DGRAD ? ((rows[i].ph - tp.kh0) >> u_sh) : rows[i].ph + tp.kh0;
          const int iwb = DGRAD ? ((rows[i].qw - tp.kw0) >> u_sh) : rows[i].qw + tp.kw0;
          const int ihlo = DGRAD ? ihb - u_dh * (u_nth - 1) : ihb, iwlo = DGRAD ? iwb - u_dw * (tp.ntw - 1) : iwb;
          const int64_t pix = rows[i].base + (int64_t)(ihlo + p.KH) * p.Ws + iwlo + p.KW;
          aoff[i] = (uint32_t)((pix * p.lda + kq) * ESA);
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
          const int n = n0 + lrowb + RPPB * i;
          boff[i] = n < p.Ncol ? (uint32_t)(((int64_t)n * p.ldb + kqb) * ESB) : OOB;
        } else {
          const int n = n0 + (t % VPR) * KEB;
          boff[i] = n < p.Ncol ? (uint32_t)(((int64_t)(t / VPR + RPI * i) * p.ldb + n) * ESB) : OOB;
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
      if (MMI_KORD) {   // channel-slab major: slab s = (channel slab s / ntaps, tap s % ntaps)
        const int s0 = k0cur / BK, cs = s0 / ntaps, tap0 = s0 - cs * ntaps;
        u_c0 = cs * BK;
        u_ti = tap0 / tp.ntw;
        u_tj = tap0 - u_ti * tp.ntw;
      } else {
        const int tap0 = k0cur / p.Kc;
        u_c0 = k0cur - tap0 * p.Kc;
        u_ti = tap0 / tp.ntw;
        u_tj = tap0 - u_ti * tp.ntw;
      }
    }
    if (VEC && !UNI) {
      ca.init(k0cur + kq, p.Kc, tp.ntw);
      if (DGRAD) {
#pragma unroll
        for (int i = 0; i < NB; ++i) cb[i].init(k0cur + t / VPR + RPI * i, p.Kc, tp.ntw);
      }
    }
    auto load_a_row = [&](int i, int rs = 0) {
      if (!VEC) {
        ra[rs][i] = load_a<DGRAD, VEC>(p, tp, rows[i], k0cur + kq);
        return;
      }
      if constexpr (UNI) {
        const int tap = u_ti * tp.ntw + u_tj;
        const int dpix = DGRAD ? (u_nth - 1 - u_ti) * u_dh * p.Ws + (tp.ntw - 1 - u_tj) * u_dw : u_ti * u_dh * p.Ws + u_tj * u_dw;
        const uint32_t soff = (uint32_t)(dpix * p.lda + u_c0) * (uint32_t)ESA;
        const uint32_t inv = (uint32_t)__builtin_amdgcn_sbfe(amask[i], tap, 1);   // -1 where this tap leaves the image
        if constexpr (A8) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            ra8[rs][i][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(srd_a, (aoff[i] | (inv & OOB)) + 16u * pl, soff, 0));
        } else if constexpr (BF) rab[rs][i] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(srd_a, aoff[i] | (inv & OOB), soff, 0));
        else ra[rs][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_a, aoff[i] | (inv & OOB), soff, 0));
        return;
      }
      int64_t pix = 0;
      const bool ok = (ca.tap < ntaps) & src_pixel<DGRAD>(p, rows[i], tp.kh0 + tp.khs * ca.ti, tp.kw0 + tp.kws * ca.tj, pix);
      if constexpr (BF)
        rab[rs][i] = *reinterpret_cast<const bf16x4*>(ok ? reinterpret_cast<const char*>(p.A) + (pix * p.lda + ca.c) * 2
                                                         : reinterpret_cast<const char*>(ZERO_SRC));
      else
        ra[rs][i] = *reinterpret_cast<const f32x4*>(ok ? p.A + pix * p.lda + ca.c : ZERO_SRC);
    };
    auto load_b_row = [&](int i, int rs = 0) {
      if (!DGRAD) {
        if (!VEC) {
          rb[rs][i] = load_b_nk<VEC>(p, n0 + lrow + RPP * i, k0cur + kq);
          return;
        }
        if constexpr (UNI) {
          const int tapw = (tp.kh0 + tp.khs * u_ti) * p.KW + tp.kw0 + tp.kws * u_tj;   // (forward: all taps, so tapw = tap)
          if constexpr (B8) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
              rb8[rs][i][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(srd_b, boff[i] + 16u * pl, (uint32_t)(tapw * p.Kc + u_c0) * (uint32_t)ESB, 0));
          } else
          rb[rs][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_b, boff[i], (uint32_t)(tapw * p.Kc + u_c0) * 4u, 0));
          return;
        }
        const int n = n0 + lrow + RPP * i, k = k0cur + kq;
        const bool ok = (n < p.Ncol) & (k < tp.Ktot);
        rb[rs][i] = *reinterpret_cast<const f32x4*>(ok ? p.B + (int64_t)n * p.ldb + k : ZERO_SRC);
      } else {
        if (!VEC) {
          rb[rs][i] = load_b_kn<VEC>(p, tp, k0cur + t / VPR + RPI * i, n0 + (t % VPR) * 4);
          return;
        }
        if constexpr (UNI) {
          const int tapw = (tp.kh0 + tp.khs * u_ti) * p.KW + tp.kw0 + tp.kws * u_tj;
          if constexpr (B8) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
              rb8[rs][i][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(srd_b, boff[i] + 16u * pl, (uint32_t)(u_c0 * p.ldb + tapw * p.Ncol) * (uint32_t)ESB, 0));
          } else
          rb[rs][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_b, boff[i], (uint32_t)(u_c0 * p.ldb + tapw * p.Ncol) * 4u, 0));
          return;
        }
        const int n = n0 + (t % VPR) * 4;
        const bool ok = (cb[i].tap < ntaps) & (n < p.Ncol);
        const int tapw = (tp.kh0 + tp.khs * cb[i].ti) * p.KW + tp.kw0 + tp.kws * cb[i].tj;
        rb[rs][i] = *reinterpret_cast<const f32x4*>(ok ? p.B + (int64_t)cb[i].c * p.ldb + (int64_t)tapw * p.Ncol + n : ZERO_SRC);
      }
    };
    auto advance = [&]() {  // move every cursor to the next slab
      k0cur += BK;
      if constexpr (UNI) {
        // Past the last slab the cursor stays where it is: the surplus prefetch of the last iteration then re-reads the last
        // slab (a scalar offset beyond the tap table would leave the buffer's range check, which covers the lane offset).
        if (k0cur < tp.Ktot) {
          if (MMI_KORD) {        // next tap of the same channel slab; after the last tap the next channel slab
            if (++u_tj == tp.ntw) {
              u_tj = 0;
              if (++u_ti == u_nth) {
                u_ti = 0;
                u_c0 += BK;
              }
            }
          } else {
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
      }
      if (VEC && !UNI) {
        ca.advance(p.Kc, tp.ntw);
        if (DGRAD) {
#pragma unroll
          for (int i = 0; i < NB; ++i) cb[i].advance(p.Kc, tp.ntw);
        }
      }
    };
    auto gload = [&](int rs = 0) {
#pragma unroll
      for (int i = 0; i < RA; ++i) load_a_row(i, rs);
#pragma unroll
      for (int i = 0; i < NB; ++i) load_b_row(i, rs);
    };
    auto lstore = [&](int stage, int rs = 0) {
      float* As = smem + stage * STAGE;
      float* Bs = As + A_ELEMS;
      if constexpr (PREC >= 1) {
        // row record: NP planes of 32 bf16 (64 B each) | 16 B pad
#pragma unroll
        for (int i = 0; i < RA; ++i) {
          __bf16* row = reinterpret_cast<__bf16*>(As + (lrow + RPPA * i) * RSF);
          if constexpr (A8) {
#pragma unroll
            for (int k = 0; k < NP; ++k) *reinterpret_cast<bf16x8*>(row + 32 * k + kq) = ra8[rs][i][k];
          } else {
            bf16x4 tm[NP];
            if constexpr (BF) tm[0] = rab[rs][i];
            else split_bf16<NP>(ra[rs][i], tm);
#pragma unroll
            for (int k = 0; k < NP; ++k) *reinterpret_cast<bf16x4*>(row + 32 * k + kq) = tm[k];
          }
        }
        if constexpr (!DGRAD) {
#pragma unroll
          for (int i = 0; i < RB; ++i) {
            __bf16* row = reinterpret_cast<__bf16*>(Bs + (lrowb + RPPB * i) * RSF);
            if constexpr (B8) {
#pragma unroll
              for (int k = 0; k < NP; ++k) *reinterpret_cast<bf16x8*>(row + 32 * k + kqb) = rb8[rs][i][k];
            } else {
              bf16x4 tm[NP];
              split_bf16<NP>(rb[rs][i], tm);
#pragma unroll
              for (int k = 0; k < NP; ++k) *reinterpret_cast<bf16x4*>(row + 32 * k + kqb) = tm[k];
            }
          }
        } else {
          char* base = reinterpret_cast<char*>(Bs);
#pragma unroll
          for (int i = 0; i < KB_IT; ++i) {
            char* dst = base + (t / VPR + RPI * i) * B_RSB + (t % VPR) * (KEB * 2);
            if constexpr (B8) {
#pragma unroll
              for (int k = 0; k < NP; ++k) *reinterpret_cast<bf16x8*>(dst + k * BK * B_RSB) = rb8[rs][i][k];
            } else {
              bf16x4 tm[NP];
              split_bf16<NP>(rb[rs][i], tm);
#pragma unroll
              for (int k = 0; k < NP; ++k) *reinterpret_cast<bf16x4*>(dst + k * BK * B_RSB) = tm[k];
            }
          }
        }
        return;
      }
#pragma unroll
      for (int i = 0; i < RA; ++i) *reinterpret_cast<f32x4*>(As + (lrow + RPP * i) * LDS_PAD + kq) = ra[rs][i];
      if (!DGRAD) {
#pragma unroll
        for (int i = 0; i < RB; ++i) *reinterpret_cast<f32x4*>(Bs + (lrow + RPP * i) * LDS_PAD + kq) = rb[rs][i];
      } else {
#pragma unroll
        for (int i = 0; i < KB_IT; ++i) *reinterpret_cast<f32x4*>(Bs + (t / VPR + RPI * i) * BN + (t % VPR) * 4) = rb[rs][i];
      }
    };

    // one slab's MFMA phase from LDS stage `stg` (spread: the next slab's global loads go out piecewise between the MFMA groups)
    auto mfma_phase = [&](int stg, auto spread) {
      constexpr bool SPREAD = decltype(spread)::value;
      const float* As = smem + stg * STAGE;
      const float* Bs = As + A_ELEMS;
      if constexpr (PREC >= 1) {
#pragma unroll
        for (int kb = 0; kb < BK / 16; ++kb) {
          // the next slab's global loads: two thirds ahead of the first 16-k block, the rest ahead of the second
#pragma unroll
          for (int i = 0; i < RA; ++i)
            if (SPREAD && (kb == 0 ? (i % 3 != 2) : (i % 3 == 2))) load_a_row(i);
#pragma unroll
          for (int i = 0; i < NB; ++i)
            if (SPREAD && (kb == 0 ? ((RA + i) % 3 != 2) : ((RA + i) % 3 == 2))) load_b_row(i);
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
          if (SPREAD && MMI_LOAD_SPREAD(i, g)) load_a_row(i);
#pragma unroll
        for (int i = 0; i < NB; ++i)
          if (SPREAD && MMI_LOAD_SPREAD(RA + i, g)) load_b_row(i);
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
    };
    if constexpr (PF2) {
      // slab r (relative to ks0) is computed from LDS stage r & 1; register set (r + 1) & 1 holds slab r + 1 on entry to iteration r
      gload(0);
      lstore(0, 0);
      advance();
      gload(1);
      advance();
      gload(0);
      __syncthreads();
      for (int ks = ks0; ks < ks1; ks += 2) {
        lstore(1, 1);          // slab r + 1 (requested two iterations ago) -> the idle stage
        advance();
        gload(1);              // slab r + 3
        mfma_phase(0, std::false_type{});
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (ks + 1 < ks1) {
          lstore(0, 0);
          advance();
          gload(0);
          mfma_phase(1, std::false_type{});
          __builtin_amdgcn_sched_barrier(0);
          __syncthreads();
        }
      }
    } else {
    gload();
    lstore(0);
    __syncthreads();

    for (int ks = ks0; ks < ks1; ++ks) {
      // The next slab is fetched unconditionally (straight-line code, counted waits): past the end of K every lane is
      // masked to the zero source, so the last iteration only stages zeros (or, in a stream-K segment that stops short
      // of the tile's end, an unused slab) into the idle buffer.
      advance();
      mfma_phase(MMI_IGEMM_STAGES == 2 ? ((ks - ks0) & 1) : 0, std::true_type{});
      __builtin_amdgcn_sched_barrier(0);  // keep the LDS stores (and their vmcnt waits) behind every MFMA of the slab
      if (MMI_IGEMM_STAGES == 1) __syncthreads();  // single LDS stage: everyone is done reading before it is overwritten
      lstore(MMI_IGEMM_STAGES == 2 ? ((ks - ks0 + 1) & 1) : 0);
      __syncthreads();
    }
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
    // (dgrad) the BatchNorm backward reduction of the layer below rides along: see IgemmP::bnr_y
    const bool bnr = DGRAD && !W41 && p.bnr_y != nullptr;  // uniform
    const bool fast_store = UNI && !EPI && !par && p.c_bytes != 0 && p.act == MMI_ACT_NONE && p.res == nullptr && !bnr &&
                            m0 + BM <= Mc && n0 + BN <= p.Ncol;  // uniform
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * WN + j * 32 + l31;
      const bool cok = col < p.Ncol;
      const float bv = (p.bias != nullptr && cok) ? p.bias[col] : 0.f;
      float s1 = 0.f, s2 = 0.f;
      float bn_m = 0.f, bn_is = 0.f, bn_g = 0.f, bn_b = 0.f, r1 = 0.f, r2 = 0.f;
      if (DGRAD && bnr && cok) {
        bn_m = p.bn_mi[col]; bn_is = p.bn_mi[p.mi_stride + col]; bn_g = p.bnr_g[col]; bn_b = p.bnr_b[col];
      }
      if (UNI && fast_store) {
        const __amdgpu_buffer_rsrc_t srd_c = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (int)p.c_bytes, 0x00020000);
        constexpr int CS = BF ? 2 : 4;   // bytes per output element
        const uint32_t voff = (uint32_t)(((m0 + wm * WM + 4 * lh) * p.ldc + col) * CS);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[i][j][r] + bv;
            s1 += v;
            s2 = __builtin_fmaf(v, v, s2);  // (explicit, so that both epilogue forms round alike)
            const uint32_t so = (uint32_t)((i * 32 + (r & 3) + 8 * (r >> 2)) * p.ldc) * (uint32_t)CS;
            if constexpr (BF) __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, (__bf16)v), srd_c, voff, so, 0);
            else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), srd_c, voff, so, 0);
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
          if (DGRAD && !BF && bnr && cok && row < Mc) {      // (the arithmetic of bn_bwd_reduce_kernel, on the value just stored)
            const float xh = (p.bnr_y[(int64_t)row * p.bnr_ldy + col] - bn_m) * bn_is;
            // (SiLU' with the hardware exp / reciprocal: one ulp each, where act_grad's expf and IEEE division are ~20 instructions
            //  per element of an epilogue that runs beside other workgroups' MFMA streams)
            const float zv = xh * bn_g + bn_b;
            float ag = 1.0f;
            if (p.bnr_act == MMI_ACT_SILU) {
              const float sg = __frcp_rn(1.0f + __expf(-zv));
              ag = sg * (1.0f + zv * (1.0f - sg));
            } else if (p.bnr_act == MMI_ACT_LEAKY) {
              ag = zv > 0.f ? 1.0f : 0.1f;
            }
            const float dzv = v * ag;
            r1 += dzv;
            r2 = __builtin_fmaf(dzv, xh, r2);
          }
        }
      }
      if (DGRAD && bnr) {
        s1 = r1;
        s2 = r2;
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
              p.bn_mi[p.mi_stride + col] = 1.0f / sqrtf((float)(var + (double)p.bn_eps));
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

}  // namespace
}  // namespace mmi_ig
