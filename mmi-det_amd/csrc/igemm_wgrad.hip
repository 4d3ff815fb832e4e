// Weight-gradient implicit GEMM: kernels, planner and C ABI.  See igemm.hip for the design notes.
#include "igemm_kernel.h"

namespace mmi_ig {
namespace {

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
// MMI_WGRAD_PAIR = 1 (fp32 MFMA path): a wave's MFMA sub-tiles INTERLEAVE their rows -- sub-tile i of the TM owns the tile rows
// TM * r + i, sub-tile j of the TN the columns TN * c + j -- so the TM (TN) fragments a lane feeds to one k-step are TM (TN)
// CONSECUTIVE floats of the k-major LDS rows: one ds_read_b64 instead of two ds_read_b32 per operand, half the LDS
// instructions of the K loop (what the in-step A/B above charged for), the same products summed in the same order (results
// bit-identical); the epilogue writes column pairs as 8-byte stores.  Only which lane holds which output element changes.
#ifndef MMI_WGRAD_PAIR
#define MMI_WGRAD_PAIR 1
#endif

// TAB (pixel-table loaders, the wgrad counterpart of the uniform-tap loaders above).  Here K runs over output pixels, so
// what every thread of a row has in common is the pixel: per slab ONE wave (taking turns) writes a 32-entry LDS table
// {byte offset of the pixel's top-left source position, bit mask of the taps that leave the image (all ones past the
// split's end)}; a loader thread adds its own constant tap/channel displacement, tests its own tap bit (2 VALU) and issues
// a buffer load whose masked lanes return zero.  The dy rows need nothing per slab: constant lane offsets against a buffer
// resource that is re-based (scalar arithmetic) to the slab's first pixel and ends at the split's last one.
// T8: dy and x arrive pre-split (t8.h: three bf16 terms per value, 48 bytes per 8 channels, written by the BatchNorm passes that
// produced them): a loader thread fetches one 8-channel group of a pixel (three 16-byte loads) and stores each term to its plane --
// the K loop carries no v_cvt / subtract; same terms, same products, bit-identical to the in-kernel split.  No bias gradient.
template <int BM, int BN, bool VEC, int PREC = 0, bool TAB = false, bool T8 = false>
__global__ __launch_bounds__(256, (BK == 32 && PREC == 0) ? MMI_WGRAD_OCC : ((BK == 32 && (PREC < 2 || PREC >= 4)) ? 3 : 2)) void wgrad_kernel(WgradP p, WgradDelta dz) {
  if (blockIdx.z != 0) {  // twin launches: blockIdx.z = problem (see igemm_kernel); uniform
    p.DY = shift_ptr(p.DY, dz.DY); p.X = shift_ptr(p.X, dz.X); p.OUT = shift_ptr(p.OUT, dz.OUT); p.OUTB = shift_ptr(p.OUTB, dz.OUTB);
    p.cnt = shift_ptr(p.cnt, dz.cnt); p.DW = shift_ptr(p.DW, dz.DW); p.DB = shift_ptr(p.DB, dz.DB);
    p.DY8 = shift_ptr(p.DY8, dz.DY8); p.X8 = shift_ptr(p.X8, dz.X8);
  }
  static_assert(!T8 || (TAB && (PREC == 2 || PREC == 3)), "pre-split operands: three-term modes, pixel-table loaders");
  static_assert(PREC == 0 || (VEC && BK == 32), "the split-bf16 forms exist for the vector loaders only");
  static_assert(!TAB || (VEC && BK == 32), "pixel-table loaders are a form of the vector loaders");
  constexpr bool BF = PREC == 4;   // bf16 storage: dy and x are bf16 in HBM, dw stays fp32 (see igemm_kernel)
  constexpr int ES = T8 ? 6 : (BF ? 2 : 4);   // bytes per element of dy and x (TAB: byte offsets against buffer resources)
  constexpr int KE = T8 ? 8 : 4;              // channels per loader thread and row
  constexpr bool ONE = BF || PREC == 5;   // PREC = 5: fp32 operands, one bf16 term each (see igemm_kernel)
  constexpr int NP = PREC == 0 || ONE ? 1 : (PREC == 3 ? 3 : PREC + 1);
  constexpr int OL = ONE ? 0 : (PREC == 3 ? 2 * (NP - 1) : NP - 1);
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  constexpr bool PAIR = MMI_WGRAD_PAIR && PREC == 0 && !MMI_WGRAD_LDS_B32;   // interleaved sub-tile rows / columns (see above)
  constexpr int VA = BM / KE, RPA = 256 / VA, ITA = BK / RPA;
  constexpr int VB = BN / KE, RPB = 256 / VB, ITB = BK / RPB;
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
  const int am = m0 + (t % VA) * KE, akr = t / VA;
  // B loader: float4 along (tap,ci): fixed per thread
  const int bn = n0 + (t % VB) * KE, bkr = t / VB;
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
  bf16x8 ra8[T8 ? ITA : 1][3], rb8[T8 ? ITB : 1][3];   // pre-split operands: one 8-channel group = three terms of 8 bf16
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
    for (int i = 0; i < ITA; ++i) a_voff[i] = am < p.Cout ? (uint32_t)(((akr + RPA * i) * p.ldy + am) * ES) : OOB;
    b_tapbit = b_kh[0] * p.KW + b_kw[0];
    b_tapoff = b_ok[0] ? (uint32_t)(((b_kh[0] * p.W + b_kw[0]) * p.ldx + b_ci[0]) * ES) : OOB;
    const int64_t margin = ((int64_t)p.KH * p.W + p.KW) * p.ldx;
    srd_x = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(T8 ? p.X8 : (const void*)p.X) - margin * ES), 0, (int)p.x_bytes, 0x00020000);
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
      if (gtab && !lin1w && wave == (j & 3) && lane < BK) {
        tnext = p.tab[(int64_t)kbeg + (int64_t)j * BK + lane];
        if constexpr (T8) tnext.x = (tnext.x >> 2) * 6u;   // (the table holds byte offsets of fp32 elements)
      }
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
          e.x = (uint32_t)(((((int64_t)timg * p.H + ih0 + p.KH) * p.W + iw0 + p.KW) * p.ldx) * ES);
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
      const int64_t left = (int64_t)(kend - k0cur) * p.ldy * ES;
      const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(reinterpret_cast<const char*>(T8 ? p.DY8 : (const void*)p.DY) + (int64_t)k0cur * p.ldy * ES), 0, left > 0 ? (left < 0x7FFFFFFF ? (int)left : 0x7FFFFFFF) : 0,
          0x00020000);
      if constexpr (T8) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) ra8[i][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(srd_a, a_voff[i] + 16u * pl, 0, 0));
      } else if constexpr (BF) rab[i] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(srd_a, a_voff[i], 0, 0));
      else ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_a, a_voff[i], 0, 0));
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
        const int64_t left = (int64_t)(kend - k0cur) * p.ldx * ES;
        const __amdgpu_buffer_rsrc_t srd_xs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(reinterpret_cast<const char*>(T8 ? p.X8 : (const void*)p.X) + (int64_t)k0cur * p.ldx * ES), 0, left > 0 ? (left < 0x7FFFFFFF ? (int)left : 0x7FFFFFFF) : 0,
            0x00020000);
        const uint32_t voff = b_ok[0] ? (uint32_t)(((bkr + RPB * i) * p.ldx + b_ci[0]) * ES) : OOB;
        if constexpr (T8) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) rb8[i][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(srd_xs, voff + 16u * pl, 0, 0));
        } else if constexpr (BF) rbb[i] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(srd_xs, voff, 0, 0));
        else rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_xs, voff, 0, 0));
        return;
      }
      const uint2 e = ptab[tab_sel][bkr + RPB * i];
      const uint32_t inv = (uint32_t)__builtin_amdgcn_sbfe((int)e.y, b_tapbit, 1);  // -1: this thread's tap leaves the image
      if constexpr (T8) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          rb8[i][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(srd_x, ((e.x + b_tapoff) | (inv & OOB)) + 16u * pl, 0, 0));
      } else if constexpr (BF) rbb[i] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(srd_x, (e.x + b_tapoff) | (inv & OOB), 0, 0));
      else rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_x, (e.x + b_tapoff) | (inv & OOB), 0, 0));
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
        char* dst = ab + (akr + RPA * i) * A_RSB + (t % VA) * (KE * 2);
        if constexpr (T8) {
#pragma unroll
          for (int k = 0; k < NP; ++k) *reinterpret_cast<bf16x8*>(dst + k * BK * A_RSB) = ra8[i][k];
          continue;
        }
        bf16x4 tm[NP];
        if constexpr (BF) tm[0] = rab[i];
        else split_bf16<NP>(ra[i], tm);
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
        char* dst = bb + (bkr + RPB * i) * B_RSB + (t % VB) * (KE * 2);
        if constexpr (T8) {
#pragma unroll
          for (int k = 0; k < NP; ++k) *reinterpret_cast<bf16x8*>(dst + k * BK * B_RSB) = rb8[i][k];
          continue;
        }
        bf16x4 tm[NP];
        if constexpr (BF) tm[0] = rbb[i];
        else split_bf16<NP>(rb[i], tm);
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
        if constexpr (PAIR && TM == 2) {
          const f32x2 v = *reinterpret_cast<const f32x2*>(As + (2 * (4 * g + e) + lh) * BM + wm * WM + 2 * l31);
          a[e][0] = v[0], a[e][TM - 1] = v[1];
        } else {
#pragma unroll
          for (int i = 0; i < TM; ++i) a[e][i] = As[(2 * (4 * g + e) + lh) * BM + wm * WM + i * 32 + l31];
        }
        if constexpr (PAIR && TN == 2) {
          const f32x2 v = *reinterpret_cast<const f32x2*>(Bs + (2 * (4 * g + e) + lh) * BN + wn * WN + 2 * l31);
          b[e][0] = v[0], b[e][TN - 1] = v[1];
        } else {
#pragma unroll
          for (int j = 0; j < TN; ++j) b[e][j] = Bs[(2 * (4 * g + e) + lh) * BN + wn * WN + j * 32 + l31];
        }
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
  // element (i, j, r) of this lane: row (i * 32 + rr) and column (j * 32 + l31) of the wave's sub-tile grid, or -- PAIR -- row
  // (TM * rr + i) and column (TN * l31 + j), rr = (r & 3) + 8 * (r >> 2) + 4 * lh
  constexpr bool PM = PAIR && TM == 2, PN = PAIR && TN == 2;
  const bool pair_store = PN && (p.Ntot & 1) == 0 && ((uintptr_t)out & 7) == 0;  // uniform: column pairs as one 8-byte store
  if (pair_store) {
    const int col = n0 + wn * WN + 2 * l31;
    if (col < p.Ntot) {   // (Ntot even: col + 1 < Ntot as well)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int row = m0 + wm * WM + (PM ? 2 * rr + i : i * 32 + rr);
          if (row < p.Cout) {
            const f32x2 v = {acc[i][0][r], acc[i][TN - 1][r]};
            float* dst = out + (int64_t)row * p.Ntot + col;
            if (fold) __hip_atomic_store(reinterpret_cast<uint64_t*>(dst), __builtin_bit_cast(uint64_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *reinterpret_cast<f32x2*>(dst) = v;
          }
        }
    }
  } else {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * WN + (PN ? 2 * l31 + j : j * 32 + l31);
      if (col < p.Ntot) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int row = m0 + wm * WM + (PM ? 2 * rr + i : i * 32 + rr);
            if (row < p.Cout) {
              if (fold) st_agent(out + (int64_t)row * p.Ntot + col, acc[i][j][r]);
              else out[(int64_t)row * p.Ntot + col] = acc[i][j][r];
            }
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

// out = sum over splits of slabs[z].  A workgroup owns 64 sixteen-byte columns; its four waves each sum every fourth split (z = wave,
// wave + 4, ...) with four independent loads in flight, and wave 0 adds the four partial sums in wave order -- a fixed order, so
// the result is run-to-run bit-identical -- four times the workgroups and a quarter of the serial chain of the one-thread-per-
// column form (a 42-split list of 128 x 1152 slabs: 144 workgroups walking 42 loads each -> 576 walking 11).
// (elements [0, n1) go to out, the bias tail [n1, n) to out2; blockIdx.y = 1: the twin problem's slabs and outputs)
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs0, float* __restrict__ out0, float* __restrict__ out20,
                                                          const float* __restrict__ slabs1, float* __restrict__ out1, float* __restrict__ out21,
                                                          int64_t n1, int64_t n, int64_t count, int splits) {
  __shared__ f32x4 part[3][64];
  const float* __restrict__ slabs = blockIdx.y ? slabs1 : slabs0;
  float* __restrict__ out = blockIdx.y ? out1 : out0;
  float* __restrict__ out2 = blockIdx.y ? out21 : out20;
  const int g = threadIdx.x >> 6, col = threadIdx.x & 63;
  const int64_t i = ((int64_t)blockIdx.x * 64 + col) * 4;
  const bool vec = i + 4 <= n1 && (n & 3) == 0;      // count = n (with the bias tail) or n1 (without); n is the slab stride
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  if (i < count && vec) {
    int z = g;
    for (; z + 12 < splits; z += 16) {
      s0 += *reinterpret_cast<const f32x4*>(slabs + (int64_t)z * n + i);
      s1 += *reinterpret_cast<const f32x4*>(slabs + (int64_t)(z + 4) * n + i);
      s2 += *reinterpret_cast<const f32x4*>(slabs + (int64_t)(z + 8) * n + i);
      s3 += *reinterpret_cast<const f32x4*>(slabs + (int64_t)(z + 12) * n + i);
    }
    for (; z < splits; z += 4) s0 += *reinterpret_cast<const f32x4*>(slabs + (int64_t)z * n + i);
    s0 = (s0 + s1) + (s2 + s3);
    if (g > 0) part[g - 1][col] = s0;
  }
  __syncthreads();
  if (g != 0 || i >= count) return;
  if (vec) {
    *reinterpret_cast<f32x4*>(out + i) = (s0 + part[0][col]) + (part[1][col] + part[2][col]);
  } else {
    for (int64_t j = i; j < count && j < i + 4; ++j) {
      float s = 0.f;
      for (int z = 0; z < splits; ++z) s += slabs[(int64_t)z * n + j];
      if (j < n1) out[j] = s;
      else out2[j - n1] = s;
    }
  }
}

}  // namespace
}  // namespace mmi_ig
using namespace mmi_ig;

namespace {
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

// nprob: problems of this shape sharing the launch (twin launches: 2): their tiles fill the chip together, so each needs
// fewer splits -- longer K chunks per workgroup and half the slab traffic per problem
WgPlan wgrad_plan(const mmi_conv_desc* d, int nprob = 1) {
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
  if (g.vec && (int64_t)tiles * nprob * max_splits < 256) {
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
  int cap = tiles * nprob > 64 ? 16 : cdiv(2 * slots, tiles * nprob);
  if (cap > max_splits) cap = max_splits;
  if (cap < 1) cap = 1;
  // Every split costs a slab of dw to write and to read back: a split count is charged `pen` of wave efficiency per split
  // (MMIDET_WGRAD_SPLIT_PENALTY, default 0: the round-1 rule -- fill whole waves, fewer splits win ties)
  static const double pen = getenv("MMIDET_WGRAD_SPLIT_PENALTY") ? atof(getenv("MMIDET_WGRAD_SPLIT_PENALTY")) : 0.0;
  int splits = 1;
  double best = -1e9;
  for (int sp = 1; sp <= cap; ++sp) {
    const int blocks = tiles * nprob * sp;
    const double eff = (double)blocks / (double)(cdiv(blocks, slots) * slots) - pen * sp;
    if (eff > best + 1e-9) best = eff, splits = sp;
  }
  if (nprob > 1) {
    // Twin launches: every split is a slab to write and read back for BOTH problems, and two problems fill a wave of the chip
    // with half the splits each -- take the fewest splits within 5 % of the best wave efficiency (3x3 128->128 @80x80: 42 splits
    // of 74 K slabs per problem instead of 85 of 37; the single-problem rule above is unchanged)
    for (int sp = 1; sp <= cap; ++sp) {
      const int blocks = tiles * nprob * sp;
      const double eff = (double)blocks / (double)(cdiv(blocks, slots) * slots) - pen * sp;
      if (eff >= best - 0.05) {
        splits = sp;
        break;
      }
    }
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
extern "C" size_t mmi_conv_wgrad_workspace(const mmi_conv_desc* d) {
  const PrecScope prec_scope_(d, 2);
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
  if (!(g.vec && g_uniform_loaders && (g_gemm_prec == 0 || g_gemm_prec == 2 || g_gemm_prec == 3) && d->KH * d->KW <= 32)) return false;
  if (d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0) return false;
  const int64_t margin = ((int64_t)d->KH * d->W + d->KW) * d->ldx;
  return (margin + ((int64_t)d->N * d->H * d->W - 1) * d->ldx + d->Cin) * 4 < (1LL << 31);
}
int64_t wgrad_table_entries(const mmi_conv_desc* d) { return (int64_t)d->N * d->Ho * d->Wo + 4 * BK; }
}  // namespace

extern "C" size_t mmi_conv_wgrad_table_bytes(const mmi_conv_desc* d) {
  const PrecScope prec_scope_(d, 2);
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
size_t align256w(size_t n) { return (n + 255) & ~(size_t)255; }
size_t wgrad_ws_one(const mmi_conv_desc* d, const WgPlan& g) {   // bytes of ONE problem's workspace
  const size_t generic = g.splits > 1 ? (size_t)g.splits * ((size_t)d->Cout * d->KH * d->KW * d->Cin + d->Cout) * sizeof(float) : 0;
  return generic ? WG_COUNTER_BYTES + generic : 0;
}
int conv_wgrad_n(int nprob, const float* const* dy, const float* const* x, float* const* dw, float* const* dbias, void* workspace,
                 size_t workspace_bytes, const void* table, const mmi_conv_desc* d, void* stream, bool bf16_io = false);
int conv_wgrad_impl(const float* dy, const float* x, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                    const void* table, const mmi_conv_desc* d, void* stream, bool bf16_io = false) {
  return conv_wgrad_n(1, &dy, &x, &dw, &dbias, workspace, workspace_bytes, table, d, stream, bf16_io);
}
}  // namespace
extern "C" int mmi_conv_wgrad(const float* dy, const float* x, float* dw, float* dbias, void* workspace,
                              size_t workspace_bytes, const mmi_conv_desc* d, void* stream) {
  return conv_wgrad_impl(dy, x, dw, dbias, workspace, workspace_bytes, nullptr, d, stream);
}
extern "C" int mmi_conv_wgrad_tab(const float* dy, const float* x, float* dw, float* dbias, void* workspace,
                                  size_t workspace_bytes, const void* table, const mmi_conv_desc* d, void* stream) {
  return conv_wgrad_impl(dy, x, dw, dbias, workspace, workspace_bytes, table, d, stream);
}
// twin form (see igemm.hip, "twin launches"): two problems of one shape in one launch; dbias may be NULL (or hold NULLs);
// workspace = mmi_conv_wgrad_workspace_n(d, 2) bytes, 256-byte aligned; the pixel table (shapes only) serves both
extern "C" size_t mmi_conv_wgrad_workspace_n(const mmi_conv_desc* d, int nprob) {
  const PrecScope prec_scope_(d, 2);
  if (check_desc(d, "mmi_conv_wgrad_workspace_n") != MMI_OK || nprob < 1 || nprob > 2 || mmi_smallconv_supported(d)) return 0;
  if (nprob == 1) return wgrad_ws_one(d, wgrad_plan(d, 1));
  const size_t one = wgrad_ws_one(d, wgrad_plan(d, 2));
  return one ? 2 * WG_COUNTER_BYTES + 2 * align256w(one - WG_COUNTER_BYTES) : 0;   // [counters 0 | counters 1 | slabs 0 | slabs 1]
}
extern "C" int mmi_conv_wgrad2(const float* const* dy, const float* const* x, float* const* dw, float* const* dbias, void* workspace,
                               size_t workspace_bytes, const void* table, const mmi_conv_desc* d, void* stream) {
  MMI_CHECK_ARG(dy && x && dw, "mmi_conv_wgrad2: null argument arrays");
  float* const nob[2] = {nullptr, nullptr};
  return conv_wgrad_n(2, dy, x, dw, dbias != nullptr ? dbias : nob, workspace, workspace_bytes, table, d, stream);
}

namespace {
int conv_wgrad_n(int nprob, const float* const* dy, const float* const* x, float* const* dw, float* const* dbias, void* workspace,
                 size_t workspace_bytes, const void* table, const mmi_conv_desc* d, void* stream, bool bf16_io) {
  const PrecScope prec_scope_(d, 2);
  if (int e = check_desc(d, "mmi_conv_wgrad")) return e;
  for (int q = 0; q < nprob; ++q) MMI_CHECK_ARG(dy[q] && x[q] && dw[q], "mmi_conv_wgrad: null pointer");
  const bool want_bias = dbias[0] != nullptr;
  if (nprob == 1 && mmi_smallconv_supported(d) && !want_bias && !bf16_io) {
    if (workspace == nullptr || workspace_bytes < WG_COUNTER_BYTES + mmi_smallconv_wgrad_workspace(d)) {
      mmi_set_error("mmi_conv_wgrad: workspace too small (%zu < %zu)", workspace_bytes, WG_COUNTER_BYTES + mmi_smallconv_wgrad_workspace(d));
      return MMI_ERR_WORKSPACE;
    }
    return mmi_smallconv_wgrad(dy[0], x[0], dw[0], (char*)workspace + WG_COUNTER_BYTES, d, (hipStream_t)stream);   // (never the counter block)
  }
  MMI_CHECK_ARG(nprob == 1 || (!mmi_smallconv_supported(d) && !bf16_io), "mmi_conv_wgrad2: no twin form for this shape / storage type");
  const WgPlan g = wgrad_plan(d, nprob);
  MMI_CHECK_ARG(!bf16_io || g.vec, "mmi_conv_wgrad_bf16: channel counts and row strides must be multiples of 4");
  const int64_t wsize = (int64_t)d->Cout * d->KH * d->KW * d->Cin;
  const int64_t slab = wsize + d->Cout;  // weight gradient + bias-gradient tail
  // one problem: [counters | slabs]; two: [counters 0 | counters 1 | slabs 0 | slabs 1] (the counter blocks at fixed offsets: they
  // must never be another shape's scratch, see igemm.hip "twin launches")
  const size_t one = wgrad_ws_one(d, g), slab_bytes = one ? align256w(one - WG_COUNTER_BYTES) : 0;
  const size_t need = nprob > 1 ? (one ? 2 * WG_COUNTER_BYTES + 2 * slab_bytes : 0) : one;
  if (g.splits > 1 && (workspace == nullptr || workspace_bytes < need || ((uintptr_t)workspace & (nprob > 1 ? 255 : 15)))) {
    mmi_set_error("mmi_conv_wgrad: workspace too small or misaligned (%zu < %zu)", workspace_bytes, need);
    return MMI_ERR_WORKSPACE;
  }
  static const bool fold_off = getenv("MMIDET_WGRAD_FOLD") != nullptr && atoi(getenv("MMIDET_WGRAD_FOLD")) == 0;  // (A/B switch)
  // The fold runs on ONE workgroup per tile, serially over the splits (a dependent round of loads per four of them), while
  // the reduce kernel spreads the same reads over the whole chip: measured twice (profiles/r02_wgrad_fold_microbench.txt: one
  // workgroup walking all splits; profiles/r02_ab_wgrad_fold_tree.txt: the fan-in-4 tree of the kernel's epilogue) the
  // in-launch fold wins up to FOUR splits (one level of the tree) and loses beyond -- every level is a dependent round of
  // device-coherent loads of slabs written on other XCDs (~6 us), against one chip-wide reduce launch that streams them:
  // 3x3 128->128@80x80 0.266 -> 0.351 ms, the step 123.2 -> 125.3 ms with the tree for every split count.  So longer split
  // lists keep the separate reduce launch; MMIDET_WGRAD_FOLD_MAX (<= 256) moves the limit.
  static const int fold_max = getenv("MMIDET_WGRAD_FOLD_MAX") ? atoi(getenv("MMIDET_WGRAD_FOLD_MAX")) : 4;
  int cnt_per_tile = 0;
  for (int n = g.splits; n > 1; n = (n + 3) / 4) cnt_per_tile += (n + 3) / 4;
  const bool fold = g.splits > 1 && g.splits <= fold_max && (int64_t)g.mtiles * g.ntiles * cnt_per_tile <= WG_MAX_TILES && !fold_off;
  // pixel-table loaders (wgrad_kernel<..., TAB>): tap mask in 32 bits, 31-bit byte offsets into x
  bool tab = false;
  uint32_t x_bytes_u = 0;
  if (g.vec && g_uniform_loaders && (g_gemm_prec == 0 || g_gemm_prec == 2 || g_gemm_prec == 3 || bf16_io) && d->KH * d->KW <= 32) {
    const int64_t margin = ((int64_t)d->KH * d->W + d->KW) * d->ldx;
    const int64_t x_bytes = (margin + ((int64_t)d->N * d->H * d->W - 1) * d->ldx + d->Cin) * (bf16_io ? 2 : 4);
    if (x_bytes < (1LL << 31)) {
      tab = true;
      x_bytes_u = (uint32_t)x_bytes;
    }
  }
  // pre-split operands announced for this launch (mmi_gemm_operands_t8: dy image, x image, and the twin problem's): taken and cleared
  const void* t8p[4];
  {
    const void** pend = t8_pending();
    for (int i = 0; i < 4; ++i) t8p[i] = pend[i], pend[i] = nullptr;
  }
  bool t8 = tab && !bf16_io && (g_gemm_prec == 2 || g_gemm_prec == 3) && !want_bias && t8p[0] != nullptr && t8p[1] != nullptr &&
            (nprob == 1 || (t8p[2] != nullptr && t8p[3] != nullptr)) && d->ldx % 8 == 0 && d->ldy % 8 == 0 && d->Cin % 8 == 0 && d->Cout % 8 == 0;
  if (t8) {
    const int64_t x8 = (int64_t)x_bytes_u / 4 * 6;
    for (int i = 0; i < 2 * nprob; ++i) t8 = t8 && ((uintptr_t)t8p[i] & 15) == 0;
    if (x8 < (1LL << 31) && t8) x_bytes_u = (uint32_t)x8;
    else t8 = false;
  }
  WgradP pp[2];
  float* slabs_of[2] = {nullptr, nullptr};
  for (int q = 0; q < nprob; ++q) {
    MMI_CHECK_ARG(!g.vec || (((uintptr_t)dy[q] | (uintptr_t)x[q]) & (bf16_io ? 7 : 15)) == 0, "mmi_conv_wgrad: operands must be 16-byte aligned");
    MMI_CHECK_ARG((dbias[q] != nullptr) == want_bias, "mmi_conv_wgrad2: bias gradients for both problems or for none");
    char* ws = (char*)workspace + (size_t)q * WG_COUNTER_BYTES;   // this problem's counter block
    float* slabs = g.splits > 1 ? (float*)((char*)workspace + (size_t)nprob * WG_COUNTER_BYTES + (size_t)q * slab_bytes) : nullptr;
    slabs_of[q] = slabs;
    WgradP& p = pp[q];
    p = WgradP{};
    p.DY = dy[q]; p.X = x[q]; p.OUT = g.splits > 1 ? slabs : dw[q];
    p.OUTB = dbias[q] == nullptr ? nullptr : (g.splits > 1 ? slabs + wsize : dbias[q]);
    p.cnt = fold ? (int*)ws : nullptr;
    p.cnt_per_tile = cnt_per_tile;
    p.DW = dw[q]; p.DB = dbias[q];
    p.zero = zero_src();
    if (p.zero == nullptr) {
      mmi_set_error("mmi_conv_wgrad: cannot resolve the zero-source symbol");
      return MMI_ERR_LAUNCH;
    }
    p.Mpix = d->N * d->Ho * d->Wo; p.Cout = d->Cout; p.Cin = d->Cin; p.KH = d->KH; p.KW = d->KW;
    p.Ho = d->Ho; p.Wo = d->Wo; p.H = d->H; p.W = d->W; p.stride = d->stride; p.pad = d->pad;
    p.ldx = d->ldx; p.ldy = d->ldy; p.Ntot = d->KH * d->KW * d->Cin; p.chunk = g.chunk;
    p.mtiles = g.mtiles; p.ntiles = g.ntiles; p.splits = g.splits; p.slab_stride = g.splits > 1 ? slab : 0;
    if (tab) {
      p.x_bytes = x_bytes_u;
      p.tab = bf16_io ? nullptr : (const uint2*)table;   // (null: the kernel builds its table slab by slab; the precomputed tables hold 4-byte offsets)
    }
    if (t8) p.DY8 = t8p[2 * q], p.X8 = t8p[2 * q + 1];
  }
  const WgradP& p = pp[0];
  WgradDelta q{};
  if (nprob > 1) {
    const WgradP& t = pp[1];
    q.DY = ptr_delta(t.DY, p.DY); q.X = ptr_delta(t.X, p.X); q.OUT = ptr_delta(t.OUT, p.OUT); q.OUTB = ptr_delta(t.OUTB, p.OUTB);
    q.cnt = ptr_delta(t.cnt, p.cnt); q.DW = ptr_delta(t.DW, p.DW); q.DB = ptr_delta(t.DB, p.DB);
    q.DY8 = ptr_delta(t.DY8, p.DY8); q.X8 = ptr_delta(t.X8, p.X8);
  }
  const dim3 grid(g.mtiles * g.ntiles, g.splits, nprob), block(256);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCHW(BM_, BN_, VEC_) \
  hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, VEC_>), grid, block, 0, s, p, q)
  if (t8) {
#define LAUNCHW8(BM_, BN_)                                                                                        \
  do {                                                                                                            \
    if (g_gemm_prec == 2) hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, true, 2, true, true>), grid, block, 0, s, p, q); \
    else hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, true, 3, true, true>), grid, block, 0, s, p, q);                  \
  } while (0)
    if (g.bm == 128 && g.bn == 128) LAUNCHW8(128, 128);
    else if (g.bm == 128) LAUNCHW8(128, 64);
    else if (g.bn == 128) LAUNCHW8(64, 128);
    else LAUNCHW8(64, 64);
#undef LAUNCHW8
  } else if (bf16_io && tab) {
#define LAUNCHWBT(BM_, BN_) hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, true, 4, true>), grid, block, 0, s, p, q)
    if (g.bm == 128 && g.bn == 128) LAUNCHWBT(128, 128);
    else if (g.bm == 128) LAUNCHWBT(128, 64);
    else if (g.bn == 128) LAUNCHWBT(64, 128);
    else LAUNCHWBT(64, 64);
#undef LAUNCHWBT
  } else if (bf16_io) {
#define LAUNCHWB(BM_, BN_) hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, true, 4>), grid, block, 0, s, p, q)
    if (g.bm == 128 && g.bn == 128) LAUNCHWB(128, 128);
    else if (g.bm == 128) LAUNCHWB(128, 64);
    else if (g.bn == 128) LAUNCHWB(64, 128);
    else LAUNCHWB(64, 64);
#undef LAUNCHWB
  } else if (g.vec && (g_gemm_prec == 2 || g_gemm_prec == 3) && tab) {
#define LAUNCHWT2(BM_, BN_)                                                                                  \
  do {                                                                                                       \
    if (g_gemm_prec == 2) hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, true, 2, true>), grid, block, 0, s, p, q); \
    else hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, true, 3, true>), grid, block, 0, s, p, q);                  \
  } while (0)
    if (g.bm == 128 && g.bn == 128) LAUNCHWT2(128, 128);
    else if (g.bm == 128) LAUNCHWT2(128, 64);
    else if (g.bn == 128) LAUNCHWT2(64, 128);
    else LAUNCHWT2(64, 64);
#undef LAUNCHWT2
  } else if (g.vec && g_gemm_prec >= 1) {
#define LAUNCHW3(BM_, BN_)                                                                              \
  do {                                                                                                  \
    if (g_gemm_prec == 1) hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, true, 1>), grid, block, 0, s, p, q);  \
    else if (g_gemm_prec == 2) hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, true, 2>), grid, block, 0, s, p, q); \
    else if (g_gemm_prec == 5) hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, true, 5>), grid, block, 0, s, p, q); \
    else hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, true, 3>), grid, block, 0, s, p, q);                   \
  } while (0)
    if (g.bm == 128 && g.bn == 128) LAUNCHW3(128, 128);
    else if (g.bm == 128) LAUNCHW3(128, 64);
    else if (g.bn == 128) LAUNCHW3(64, 128);
    else LAUNCHW3(64, 64);
#undef LAUNCHW3
  } else if (!g.vec) LAUNCHW(64, 64, false);
  else if (tab) {
#define LAUNCHWT(BM_, BN_) hipLaunchKernelGGL((wgrad_kernel<BM_, BN_, true, 0, true>), grid, block, 0, s, p, q)
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
    const int64_t count = want_bias ? slab : wsize;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(count, 256), nprob), dim3(256), 0, s, (const float*)slabs_of[0], dw[0], dbias[0],
                       (const float*)slabs_of[nprob - 1], dw[nprob - 1], dbias[nprob - 1], wsize, slab, count, g.splits);
    MMI_CHECK_LAUNCH("mmi_conv_wgrad(reduce)");
  }
  return MMI_OK;
}
}  // namespace

extern "C" int mmi_conv_wgrad_bf16(const void* dy, const void* x, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                                   const mmi_conv_desc* d, void* stream) {
  return conv_wgrad_impl((const float*)dy, (const float*)x, dw, dbias, workspace, workspace_bytes, nullptr, d, stream, true);
}
