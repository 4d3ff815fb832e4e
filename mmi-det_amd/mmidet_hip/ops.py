"""autograd glue over the C ABI.  Activations are NHWC tensors (N,H,W,C) (or (rows,C) token matrices) whose channel
axis is contiguous; a tensor may be a channel slice of a wider buffer (row stride ld > C).  Every function here only
enqueues HIP kernels of libmmidet_hip.so on the current torch stream."""
import torch
from torch.autograd import Function

from . import alloc, lib
from .lib import ACT_LEAKY, ACT_NONE, ACT_SILU, ConvDesc  # noqa: F401

_scratch = {}
_retired = []     # workspaces replaced by a larger one: kept until the streams that may still use them have been joined


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """hipStream_t of the current torch stream as an integer (the raw getter is ~10x cheaper than building a Stream object;
    it is called several thousand times per step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def scratch(nfloats, device, slot=0, stream=None):
    """Grow-only fp32 workspace per (device, slot, stream): kernels on one stream run in order, so a stream may reuse its
    buffer for every call; the two backbone lanes and the wgrad side streams each get their own."""
    key = (device, slot, _stream() if stream is None else stream)
    buf = _scratch.get(key)
    if buf is None or buf.numel() < nfloats:
        if buf is not None:
            # A workspace of ANOTHER stream (stream=side) is allocated while the lane stream is current, so the caching
            # allocator would hand its memory to the next lane-stream allocation the moment the last reference drops --
            # while an earlier wgrad on the side stream may still be reducing into it.  Keep it until join_pending().
            _retired.append(buf)
        buf = alloc.workspace(max(int(nfloats), 1 << 16), torch.float32, device)
        _scratch[key] = buf
    return buf


_zeroed = {}


def zeroed_scratch(nbytes, device, stream=None, tag=0):
    """Grow-only, zero-filled-at-birth byte workspace per (device, tag, stream): the kernels keep the arrival counters at its
    head at zero between launches, so the fill happens once per buffer (include/mmidet_hip.h).  One tag per workspace layout
    (0: conv forward / dgrad, 'w': wgrad, 'bn': BatchNorm backward) -- what is a counter in one layout is scratch in another."""
    key = (device, tag, _stream() if stream is None else stream)
    buf = _zeroed.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None and stream is not None and stream != _stream():
            _retired.append(buf)        # (see scratch(): a side stream may still be using it)
        buf = alloc.workspace(int(nbytes), torch.uint8, device, zero=True)
        _zeroed[key] = buf
    return buf


def check_counters():
    """The arrival counters at the head of every zero-initialised workspace are zero between launches: each last-arriver election
    (stream-K tiles, BatchNorm statistics folds, weight-gradient split folds) resets the counters it used.  Synchronises; raises
    if an election was left half done.  Header sizes from the library (mmi_workspace_header_bytes)."""
    torch.cuda.synchronize()
    kinds = {0: 0, 'twin': 1, 'w': 2, 'twinw': 3, 'bn': 4}
    bad = []
    for (device, tag, stream), buf in _zeroed.items():
        nb = lib.workspace_header_bytes(kinds[tag])
        p = buf.data_ptr()
        off = ((p + 255) & ~255) - p if tag in ('twin', 'twinw') else 0     # (twin_ops._aligned)
        head = buf[off:off + min(nb, buf.numel() - off)]
        nz = int(torch.count_nonzero(head))
        if nz:
            bad.append('%d non-zero counter bytes in the %r workspace of stream %#x' % (nz, tag, stream))
    if bad:
        raise AssertionError('arrival counters not reset: ' + '; '.join(bad))
    return len(_zeroed)


def t8_image(t, stream=None):
    """The pre-split ("T8") image of an fp32 rows tensor (include/mmidet_hip.h: mmi_split_t8): a bf16 tensor (..., C/8, 3, 8) with
    the tensor's leading shape, written by the stand-alone converter on the current stream.  Channel count a multiple of 8."""
    t, ld = rows_of(t)
    c = t.shape[-1]
    assert c % 8 == 0 and t.dtype == torch.float32
    img = alloc.empty((*t.shape[:-1], c // 8, 3, 8), dtype=torch.bfloat16, device=t.device)
    lib.split_t8(t.data_ptr(), ld, img.data_ptr(), c, _nrows(t), c, _stream() if stream is None else stream)
    return img


def conv_fwd(x, w, bias, y, part, d, s):
    """mmi_conv_fwd on tensors (None -> NULL) with the stream-K workspace the shape asks for."""
    nb = fwd_plan(d)[0]
    ws = zeroed_scratch(nb, x.device, s) if nb else None
    lib.conv_fwd(x.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
                 part.data_ptr() if part is not None else None, ws.data_ptr() if nb else None, nb, d, s)


def conv_fwd_raw(xp, wp, yp, d, device, s):
    """mmi_conv_fwd on raw addresses (no bias, no statistics): the inference form of a lane of a twin layer."""
    nb = fwd_plan(d)[0]
    ws = zeroed_scratch(nb, device, s) if nb else None
    lib.conv_fwd(xp, wp, None, yp, None, ws.data_ptr() if nb else None, nb, d, s)


_dgrad_ws = {}
_wgrad_ws = {}
_bnbwd_ws = {}


def conv_dgrad(dy, w, dx, d, s):
    k = _desc_key(d)
    nb = _dgrad_ws.get(k)
    if nb is None:
        nb = _dgrad_ws[k] = lib.conv_dgrad_workspace(d)
    ws = zeroed_scratch(nb, dy.device, s) if nb else None
    lib.conv_dgrad(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), ws.data_ptr() if nb else None, nb, d, s)


def conv_dgrad_bnred(dy, w, dx, d, hook, s, skip=None, lds=0):
    """conv_dgrad with the BatchNorm backward reduction of the layer below in its epilogue (lib.BnReduceHook; include/mmidet_hip.h);
    skip: rows tensor added in the epilogue (1x1 stride-1 layers)."""
    k = _desc_key(d)
    nb = _dgrad_ws.get(k)
    if nb is None:
        nb = _dgrad_ws[k] = lib.conv_dgrad_workspace(d)
    ws = zeroed_scratch(nb, dy.device, s) if nb else None
    lib.conv_dgrad_bnred(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), skip.data_ptr() if skip is not None else None, lds, hook,
                         ws.data_ptr() if nb else None, nb, d, s)


def rows_of(t):
    """Return (tensor, ld): `tensor` is `t` (or a compacted copy) viewed as rows x C with unit channel stride and a
    uniform row stride ld."""
    assert t.dtype in (torch.float32, torch.bfloat16) and t.is_cuda, 'mmidet_hip ops need fp32 (or bf16-storage) tensors on the MI355X'
    C = t.shape[-1]
    if t.is_contiguous():          # the common case
        return t, C
    if t.dim() == 1:
        return (t if t.stride(0) == 1 else t.contiguous()), C
    ok = t.stride(-1) == 1 or C == 1
    ld = t.stride(-2) if t.shape[-2] != 1 else C
    if ok and t.dim() > 2:
        expect = ld * t.shape[-2]
        for d in range(t.dim() - 3, -1, -1):
            if t.shape[d] != 1 and t.stride(d) != expect:
                ok = False
                break
            expect *= t.shape[d]
    if ok and t.shape[-2] != 1 and ld < C:
        ok = False
    if not ok:
        t = t.contiguous()
        ld = C
    return t, ld


def _nrows(t):
    return t.numel() // t.shape[-1]


def _desc(x_shape, cout, k, stride, ldx, ldy):
    n, h, w, cin = x_shape
    pad = k // 2
    ho = (h + 2 * pad - k) // stride + 1
    wo = (w + 2 * pad - k) // stride + 1
    return ConvDesc(n, h, w, cin, ho, wo, cout, k, k, stride, pad, ldx, ldy)


def _ohwi(w):
    """Conv weights are (Cout,Cin,KH,KW) tensors in channels_last memory = OHWI; Linear weights are (N,K)."""
    if w.dim() == 2:
        return w if w.is_contiguous() else w.contiguous()
    if not w.is_contiguous(memory_format=torch.channels_last):
        w = w.contiguous(memory_format=torch.channels_last)
    return w


OVERLAP_WGRAD = __import__("os").environ.get("MMIDET_OVERLAP_WGRAD", "1") != "0"   # weight-gradient GEMM on a side HIP stream next to dgrad
PACK_C3 = __import__("os").environ.get("MMIDET_PACK_C3", "1") != "0"             # A/B: C3's cv1 | cv2 as one GEMM, no concat copy
SKIP_FUSE = __import__("os").environ.get("MMIDET_SKIP_FUSE", "1") != "0"         # A/B: Bottleneck shortcut gradient in the dgrad epilogue
SHARED_SIDE = __import__("os").environ.get("MMIDET_SHARED_SIDE", "0") == "1"   # one wgrad stream for both backbone lanes
# Deferred join: the lane never waits for its wgrad stream per layer; the operands are kept alive in _pending and the
# caller joins once after backward (join_pending).  Nobody may read a weight gradient before that: valid while .grad is
# None at backward time (autograd then adopts dw without touching it), i.e. not with the DDP flat buckets.
DEFER_JOIN = False   # switched on by TrainStep around its own backward only
# Data parallel: weight data_ptr -> view of the reducer's flat bucket.  The wgrad kernel then writes the gradient where
# the collective reads it, and autograd adopts that view as .grad (no accumulate pass, no pack copy): ddp.GradReducer.
GRAD_SLOTS = {}
SLOT_HANDED_OUT = set()   # weights whose gradient was written into its slot this step (checked by the reducer's hook)
_side_streams = {}
_side_rr = 0
# wgrad side streams per lane, used round-robin: two let a small weight-gradient GEMM (token projections, 1x1 convs) run beside
# a large one instead of behind it (125.3 -> 123.4 ms/step; three are worse: 127.9)
# (Round 2: MMIDET_NSIDE=1 -- every wgrad of a lane behind one another on one stream -- used to end in a GPU memory access fault in
#  the first warm-up step, three runs of three.  Cause: the pixel tables of wgrad_table() are shared by the two backbone lanes
#  (same geometries), and the lane that did NOT build a table used it without being ordered behind the other lane's build kernel
#  -- in the first step a wgrad could walk a table that was still uninitialised memory; likewise a table replaced by a larger one
#  was freed while a queued wgrad could still read it.  Both gaps are closed in wgrad_table(); with them closed the setting runs
#  (122.07 ms/step, gpurun_out/r2_nside1b.json) and tests/test_step_gpu.py covers it.  The default was exposed to the same race;
#  two wgrad streams per lane merely made the window small.)
NSIDE = max(1, int(__import__('os').environ.get('MMIDET_NSIDE', '2')))
_pending = []
_pending_sides = {}


def _side_stream(device):
    """The wgrad companion of the CURRENT stream (each backbone lane has its own)."""
    global _side_rr
    _side_rr = (_side_rr + 1) % NSIDE
    key = (device, 0 if SHARED_SIDE else _stream(), _side_rr)
    s = _side_streams.get(key)
    if s is None:
        s = torch.cuda.Stream(device=device)   # (HIP stream priorities, two levels here, made no measurable difference)
        _side_streams[key] = s
    return s


def grad_like(t):
    """Output tensor for the gradient of parameter tensor `t`: its bucket view under data parallelism (see GRAD_SLOTS),
    else fresh memory of the parameter's layout."""
    slot = GRAD_SLOTS.get(t.data_ptr())
    if slot is not None and slot.shape == t.shape and slot.stride() == t.stride():
        SLOT_HANDED_OUT.add(t.data_ptr())
        return slot.detach()        # a fresh alias (sole owner), so that AccumulateGrad adopts it instead of cloning
    if (slot is not None and t.dim() == 2 and slot.dim() == 4 and tuple(slot.shape) == (*t.shape, 1, 1) and t.is_contiguous()
            and slot.is_contiguous(memory_format=torch.channels_last)):
        # a 1x1 conv weight used as a matrix (GPT1_fourier's gates: `conv.weight.view(8, c)`): the same bytes in the same order,
        # so the kernel writes the bucket view here too and the view's backward hands autograd an alias of it.  (Round 3: a
        # fresh tensor here was copied into the slot by the reducer's hook on the LANE stream while its wgrad kernel was still
        # running on the side stream -- deferred join -- which a captured step with wgrad overlap then exposed.)
        SLOT_HANDED_OUT.add(t.data_ptr())
        return slot.detach().as_strided(t.shape, t.stride())
    return alloc.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=t.device)


_wgrad_tabs = {}       # (device, geometry) -> uint8 tensor holding the layer's pixel table
_wgrad_tab_use = {}    # descriptor -> table bytes (0: this shape takes none)
WGRAD_TABLE = __import__("os").environ.get("MMIDET_WGRAD_TABLE", "1") != "0"      # A/B: precomputed pixel tables for wgrad


def wgrad_table(d, device):
    """The precomputed pixel table of this layer geometry (include/mmidet_hip.h: mmi_conv_wgrad_table_build) or None.  Built
    once, on the current stream, the first time a geometry is seen; it depends on shapes and strides only."""
    if not WGRAD_TABLE:
        return None
    k = _desc_key(d)
    nb = _wgrad_tab_use.get(k)
    if nb is None:
        nb = _wgrad_tab_use[k] = lib.conv_wgrad_table_bytes(d)
    if nb == 0:
        return None
    g = (device, d.N, d.H, d.W, d.Ho, d.Wo, d.KH, d.KW, d.stride, d.pad, d.ldx)
    ent = _wgrad_tabs.get(g)
    cur = _stream()
    if torch.cuda.is_current_stream_capturing():
        # Never build inside a capture (the build would be a node of ONE lane, the other lane's use in the same graph would not be
        # ordered behind it, and the table would live in the graph's private pool while this cache outlives the graph).  A table
        # built by the warm-up steps is a constant by now (capture follows a device synchronisation); a geometry first seen
        # here takes the kernel's own slab-by-slab builder, which gives bit-identical results.
        return ent[0] if ent is not None and ent[0].numel() >= nb else None
    if ent is None or ent[0].numel() < nb:
        if ent is not None:
            _retired.append(ent[0])      # a queued wgrad on a side stream may still read the smaller table (see scratch())
        t = alloc.workspace(nb, torch.uint8, device)
        lib.conv_wgrad_table_build(t.data_ptr(), d, cur)
        ev = torch.cuda.Event()
        ev.record()
        ent = _wgrad_tabs[g] = (t, ev, {cur})
    elif cur not in ent[2]:
        # The twin backbones share geometries, hence tables: the lane that did not build this one orders itself behind the
        # build (once per stream; afterwards the table is constant).
        torch.cuda.current_stream().wait_event(ent[1])
        ent[2].add(cur)
    return ent[0]


def _wgrad(dy, lddy, x, ldx, w, d, overlap=False, want_bias=False, bias=None, out=None):
    """dw = dy^T x.  With overlap=True the kernel is enqueued on the side stream behind everything already on the
    current stream; the caller must `_join_side()` before the current stream (or anyone else) touches dw.  dgrad and wgrad
    of one layer are independent, and two co-running grids fill each other's partial last wave (the fp32-MFMA kernels
    lose up to a third of the chip to wave quantisation when they run alone).
    out = (dw, db or None): caller-owned outputs (several parameters that lie back to back and take one GEMM: pack_qkv)."""
    if out is not None:
        dw, db = out
        assert (db is not None) == bool(want_bias)
    else:
        dw = grad_like(w)
        db = (grad_like(bias) if bias is not None else alloc.empty(w.shape[0], dtype=w.dtype, device=w.device)) if want_bias else None
    dbp = db.data_ptr() if want_bias else None
    k = _desc_key(d)
    nbytes = _wgrad_ws.get(k)
    if nbytes is None:
        nbytes = _wgrad_ws[k] = lib.conv_wgrad_workspace(d)
    bf = dy.dtype == BF16
    assert x.dtype == dy.dtype, 'wgrad operands must share the storage type'
    tab = None if bf else wgrad_table(d, w.device)
    tabp = tab.data_ptr() if tab is not None else None

    def launch(wsp, st):
        if bf:
            lib.conv_wgrad_bf16(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), dbp, wsp, nbytes, d, st)
        else:
            lib.conv_wgrad_tab(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), dbp, wsp, nbytes, tabp, d, st)
    if overlap:
        main, side = torch.cuda.current_stream(), _side_stream(w.device)
        ws = zeroed_scratch(nbytes, w.device, side.cuda_stream, tag='w') if nbytes else None
        side.wait_stream(main)
        launch(ws.data_ptr() if ws is not None else None, side.cuda_stream)
        if DEFER_JOIN:
            _pending.append((dy, x))
            _pending_sides[side.cuda_stream] = side
    else:
        ws = zeroed_scratch(nbytes, w.device, tag='w') if nbytes else None
        launch(ws.data_ptr() if ws is not None else None, _stream())
    return (dw, db) if want_bias else dw


def _join_side(device):
    if not DEFER_JOIN:
        cur = torch.cuda.current_stream()
        lane = 0 if SHARED_SIDE else cur.cuda_stream
        for r in range(NSIDE):                      # every side stream of this lane that exists
            s = _side_streams.get((device, lane, r))
            if s is not None:
                cur.wait_stream(s)


def side_streams_in_flight():
    """wgrad streams with work the current stream has not joined yet (deferred-join mode)."""
    return list(_pending_sides.values())


def join_pending():
    """After backward (deferred-join mode): the current stream waits for every wgrad stream, then the operands go."""
    cur = torch.cuda.current_stream()
    for side in _pending_sides.values():
        cur.wait_stream(side)
    _pending_sides.clear()
    _pending.clear()
    if _retired and not torch.cuda.is_current_stream_capturing():
        for side in _side_streams.values():      # (per-layer-join mode included: every wgrad stream, once)
            cur.wait_stream(side)
        _retired.clear()


BF16 = torch.bfloat16


def raw_cast(t, dtype):
    """fp32 <-> bf16 copy of an NHWC / rows tensor (no autograd): the storage-mode boundary (csrc/bf16_ops.hip)."""
    if t.dtype == dtype:
        return t
    t, ld = rows_of(t)
    out = alloc.empty(tuple(t.shape), dtype=dtype, device=t.device)
    c = t.shape[-1]
    fn = lib.cast_f32_bf16 if dtype == BF16 else lib.cast_bf16_f32
    fn(t.data_ptr(), ld, out.data_ptr(), c, _nrows(t), c, _stream())
    return out


class _Cast(Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return raw_cast(x, dtype)

    @staticmethod
    def backward(ctx, g):
        return raw_cast(g, ctx.src), None


def cast(x, dtype):
    return x if x.dtype == dtype else _Cast.apply(x, dtype)


class Dest:
    """A pre-allocated NHWC buffer that producers write channel slices of (C3's concat buffer: models/common.py:650 of the
    reference without the copy).  Passed to the ops as (Dest, first channel): a plain Python object, invisible to autograd."""

    def __init__(self, t):
        self.t = t


def _dest_view(dest, shape):
    holder, off = dest
    out = holder.t[..., off:off + shape[-1]]
    assert tuple(out.shape) == tuple(shape), 'destination slice %s does not fit the output %s' % (tuple(out.shape), tuple(shape))
    return out


_plan_cache = {}


def _desc_key(d):
    return (lib.plan_epoch[0], d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout, d.KH, d.KW, d.stride, d.pad, d.ldx, d.ldy)


def fwd_plan(d):
    """(workspace bytes, statistics row blocks) of a forward descriptor: planner queries cached per shape (two C calls per
    convolution per step otherwise)."""
    k = _desc_key(d)
    v = _plan_cache.get(k)
    if v is None:
        v = _plan_cache[k] = (lib.conv_fwd_workspace(d), lib.conv_fwd_row_blocks(d))
    return v


def bn_bwd_ws(rows, c):
    v = _bnbwd_ws.get((rows, c))
    if v is None:
        v = _bnbwd_ws[(rows, c)] = lib.bn_act_bwd_workspace(rows, c)
    return v


def _bn_forward(x, w, y, d, cout, rows, training, eps, momentum, rmean, rvar, nbt, nbt2, s):
    """conv + BatchNorm statistics (training: folded inside the conv launch) -> mean_invstd (2*cout)."""
    mi = alloc.empty(2 * cout, dtype=torch.float32, device=x.device)
    bf = x.dtype == BF16
    nb, nrb = fwd_plan_bf16(d) if bf else fwd_plan(d)
    ws = zeroed_scratch(nb, x.device, s) if nb else None
    wsp = ws.data_ptr() if nb else None
    if training:
        part = scratch((nrb + 64) * 2 * cout, x.device)     # + MMI_BN_FOLD_ROWS spare rows (the CEM's separate fold)
        bn = lib.BnStats(eps, momentum, rmean.data_ptr(), rvar.data_ptr(), nbt.data_ptr() if nbt is not None else None,
                         nbt2.data_ptr() if nbt2 is not None else None, mi.data_ptr())
        if bf:
            lib.conv_fwd_bf16(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), part.data_ptr(), bn, wsp, nb, d, s)
        else:
            lib.conv_bn_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), part.data_ptr(), bn, wsp, nb, d, s)
    else:
        if bf:
            lib.conv_fwd_bf16(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), None, None, wsp, nb, d, s)
        else:
            lib.conv_fwd(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), None, wsp, nb, d, s)
        lib.bn_eval_stats(rmean.data_ptr(), rvar.data_ptr(), cout, eps, mi.data_ptr(), s)
    return mi


def fwd_plan_bf16(d):
    k = ('bf16',) + _desc_key(d)
    v = _plan_cache.get(k)
    if v is None:
        v = _plan_cache[k] = (lib.conv_fwd_workspace_bf16(d), lib.conv_fwd_row_blocks_bf16(d))
    return v


def _bn_act_fwd(y, ldy, mi, gamma, beta, residual, ldr, out, ldo, out1, ldo1, split, rows, c, act, s):
    fn = lib.bn_act_fwd_split_bf16 if y.dtype == BF16 else lib.bn_act_fwd_split
    fn(y.data_ptr(), ldy, mi.data_ptr(), gamma.data_ptr(), beta.data_ptr(), residual.data_ptr() if residual is not None else None,
       ldr, out.data_ptr(), ldo, out1.data_ptr() if out1 is not None else None, ldo1, split, rows, c, act, s)


def _bn_act_bwd(y, ldy, dout, ldd, dout1, ldd1, split, mi, gamma, beta, dy, dgs, rows, c, act, frozen, s):
    nbw = bn_bwd_ws(rows, c)
    ws = zeroed_scratch(nbw, y.device, s, tag='bn')
    fn = lib.bn_act_bwd_bf16 if y.dtype == BF16 else lib.bn_act_bwd
    fn(y.data_ptr(), ldy, dout.data_ptr(), ldd, dout1.data_ptr() if dout1 is not None else None, ldd1, split, mi.data_ptr(),
       gamma.data_ptr(), beta.data_ptr(), ws.data_ptr(), nbw, dy.data_ptr(), c, dgs[0].data_ptr(), dgs[1].data_ptr(),
       dgs[2].data_ptr() if dgs[2] is not None else None, dgs[3].data_ptr() if dgs[3] is not None else None, rows, c, act, frozen, s)


def _conv_dgrad_any(dy, w, dx, dd, s, skip=None, lds=0):
    if dy.dtype == BF16:
        lib.conv_dgrad_bf16(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), skip.data_ptr() if skip is not None else None, lds, dd, s)
    else:
        assert skip is None
        conv_dgrad(dy, w, dx, dd, s)


def _dgrad_accumulate(dy, w, dx, d, dd, skip, s):
    """dx = conv_transpose(dy, w) + skip.  1x1 stride-1 layers add in the GEMM epilogue (MMI_EPI_ACCUMULATE); others in a pass
    of their own."""
    skip, lds = rows_of(skip)
    if dy.dtype == BF16:
        skip = raw_cast(skip, BF16)
        if d.KH == 1 and d.stride == 1 and lds % 4 == 0:
            _conv_dgrad_any(dy, w, dx, dd, s, skip, lds)
        else:
            _conv_dgrad_any(dy, w, dx, dd, s)
            lib.add_bf16(dx.data_ptr(), d.Cin, skip.data_ptr(), lds, dx.data_ptr(), d.Cin, _nrows(dx), d.Cin, s)
        return
    if d.KH == 1 and d.stride == 1 and d.Cin % 4 == 0 and d.Cout % 4 == 0 and dd.ldy % 4 == 0 and lds % 4 == 0:
        nb = lib.conv_dgrad_workspace(dd)
        ws = zeroed_scratch(nb, dy.device, s) if nb else None
        e = lib.LinearEpilogue(lib.EPI_ACCUMULATE, lds, lds, 0.0, skip.data_ptr(), None, 0, None)
        lib.linear_dgrad_fused(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), ws.data_ptr() if nb else None, nb, dd, e, s)
    else:
        conv_dgrad(dy, w, dx, dd, s)
        lib.add(dx.data_ptr(), d.Cin, skip.data_ptr(), lds, dx.data_ptr(), d.Cin, _nrows(dx), d.Cin, s)


# The BatchNorm backward REDUCTION of a layer can ride in the epilogue of the dgrad that produces its incoming gradient (csrc/
# igemm_kernel.h, IgemmP::bnr_y; profiles/r04_bn_pass_knockout.txt: the separate pass costs ~6 ms of the 118 ms step).  Built,
# verified (tests/test_bnred_gpu.py) and measured SLOWER than the pass it removes -- step +2.2 ms with the 3x3 layers' dgrads
# carrying it, +5.5 ms with the 1x1 layers' too (profiles/r04_bnred_step_ab.txt): in the MFMA accumulator layout a lane owns one
# column of four rows, so y arrives as 4-byte loads and the 15-20 VALU instructions per element sit in an epilogue that the other
# workgroups' MFMA streams do not hide.  OFF by default ("1": on).
BNRED = __import__("os").environ.get("MMIDET_BNRED", "0") == "1"
BNRED_COUNT = {'taken': 0, 'rejected': 0}      # (tests / diagnostics)
# ... in the dgrad of 1x1 layers too ("1"): their tiles have 4-8 K slabs, the added epilogue work is as long as the whole K loop
BNRED_K1 = __import__("os").environ.get("MMIDET_BNRED_K1", "0") == "1"


class BnSrc:
    """What a consumer's dgrad needs to know about the Conv+BatchNorm+activation layer whose OUTPUT it differentiates, hung on that
    output tensor (`t._bnsrc`) by the producer's forward.  uses: hook-aware consumers seen in forward.  The consumer's backward leaves
    parts = (partials, nparts) and the identity of the gradient tensor it wrote (address, version counter); the producer's backward
    takes the short form only if the gradient it receives IS that tensor, unmodified -- any other reader of the output makes autograd
    hand over a sum (another tensor, or the same one with a bumped version), and the full reduction runs as before."""
    __slots__ = ('y', 'mi', 'gammas', 'betas', 'act', 'twin', 'uses', 'parts', 'dx_ptr', 'dx_version')

    def __init__(self, y, mi, gammas, betas, act, twin):
        self.y, self.mi, self.gammas, self.betas, self.act, self.twin = y, mi, gammas, betas, act, twin
        self.uses, self.parts, self.dx_ptr, self.dx_version = 0, None, 0, -1

    def wrote(self, dx, parts):
        self.parts, self.dx_ptr, self.dx_version = parts, dx.data_ptr(), dx._version

    def take(self, dout):
        """The partial sums, if `dout` is the tensor they were taken from; they are consumed either way."""
        parts, self.parts = self.parts, None
        if parts is None:
            return None
        if dout.data_ptr() == self.dx_ptr and dout._version == self.dx_version and dout.is_contiguous():
            BNRED_COUNT['taken'] += 1
            return parts
        BNRED_COUNT['rejected'] += 1      # (correct, but the epilogue's work was wasted: another reader of the output exists)
        return None


def bn_src_of(x, twin):
    """The BnSrc of a forward input, counted as used -- or None (no source, grad mode off, the feature off, layouts differ)."""
    src = getattr(x, '_bnsrc', None) if BNRED else None
    if src is None or src.twin != twin or x.dtype != torch.float32:
        return None
    src.uses += 1
    return src


class _ConvBnAct(Function):
    """act(BN(conv(x))) [+ residual]; training-mode BN statistics are finished inside the conv launch.
    skip=True also returns x itself as a second output: a Bottleneck hands that to its second conv as the residual, so the
    shortcut's gradient arrives here as a second incoming gradient and is added in the dgrad epilogue instead of by autograd's
    fan-out accumulation (an ATen kernel).  dest=(Dest, channel): write the output into that slice of a wider buffer."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, rmean, rvar, nbt, residual, stride, act, training, eps, momentum, skip, dest):
        x_in = x
        x, ldx = rows_of(x)
        w = _ohwi(w)
        cout, k = w.shape[0], w.shape[2]
        d = _desc(x.shape, cout, k, stride, ldx, cout)
        y = alloc.empty((d.N, d.Ho, d.Wo, cout), dtype=x.dtype, device=x.device)
        rows = d.N * d.Ho * d.Wo
        s = _stream()
        mi = _bn_forward(x, w, y, d, cout, rows, training, eps, momentum, rmean, rvar, nbt, None, s)
        out = _dest_view(dest, y.shape) if dest is not None else alloc.empty_like(y)
        ldo = rows_of(out)[1]
        assert out.stride(-1) == 1 and rows_of(out)[0] is out and out.dtype == y.dtype, 'the destination slice must be a strided NHWC view'
        ldr = 0
        if residual is not None:
            residual, ldr = rows_of(raw_cast(residual, y.dtype))
        _bn_act_fwd(y, cout, mi, gamma, beta, residual, ldr, out, ldo, None, 0, cout, rows, cout, act, s)
        ctx.save_for_backward(x, w, y, mi, gamma, beta)
        ctx.cfg = (d, act, training, residual is not None, skip)
        ctx.src_in = bn_src_of(x_in, False) if ctx.needs_input_grad[0] else None
        ctx.src_out = None
        if BNRED and dest is None and y.dtype == torch.float32 and any(ctx.needs_input_grad):
            ctx.src_out = out._bnsrc = BnSrc(y, mi, (gamma,), (beta,), act, False)
        return (out, x_in) if skip else out

    @staticmethod
    def backward(ctx, dout, dskip=None):
        x, w, y, mi, gamma, beta = ctx.saved_tensors
        d, act, training, has_res, skip = ctx.cfg
        dout, ldd = rows_of(raw_cast(dout, y.dtype))
        cout = d.Cout
        rows = d.N * d.Ho * d.Wo
        s = _stream()
        dy = alloc.empty_like(y)
        dgamma = grad_like(gamma)
        dbeta = grad_like(beta)
        parts = ctx.src_out.take(dout) if ctx.src_out is not None and ldd == cout else None
        if parts is not None:        # the reduction came out of the consumer's dgrad epilogue: fold + apply
            lib.bn_act_bwd_apply(y.data_ptr(), cout, dout.data_ptr(), ldd, mi.data_ptr(), gamma.data_ptr(), beta.data_ptr(), parts[0].data_ptr(),
                                 parts[1], dy.data_ptr(), cout, dgamma.data_ptr(), dbeta.data_ptr(), rows, cout, act, 0 if training else 1, s)
        else:
            _bn_act_bwd(y, cout, dout, ldd, None, 0, cout, mi, gamma, beta, dy, (dgamma, dbeta, None, None), rows, cout, act,
                        0 if training else 1, s)
        dx = None
        both = OVERLAP_WGRAD and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]
        dw = _wgrad(dy, cout, x, d.ldx, w, d, overlap=both) if ctx.needs_input_grad[1] else None
        if ctx.needs_input_grad[0]:
            dx = alloc.empty((d.N, d.H, d.W, d.Cin), dtype=x.dtype, device=x.device)
            dd = ConvDesc(d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout, d.KH, d.KW, d.stride, d.pad, d.Cin, cout)
            src, sk, lds = ctx.src_in, None, 0
            if skip and dskip is not None:
                sk, lds = rows_of(dskip)
            ok = (src is not None and src.uses == 1 and d.stride == 1 and (d.KH == 3 or (d.KH == 1 and BNRED_K1)) and d.ldx == d.Cin and d.Cin % 4 == 0 and cout % 4 == 0
                  and d.Cin >= 32 and cout >= 32 and           # (not the CEM's direct 3 / 24-channel convolutions)
                  (sk is None or (d.KH == 1 and lds % 4 == 0 and sk.dtype == torch.float32 and sk.data_ptr() % 16 == 0)))
            if ok:
                kb = ('dgb',) + _desc_key(dd)
                nparts = _dgrad_ws.get(kb)
                if nparts is None:
                    nparts = _dgrad_ws[kb] = lib.conv_dgrad_row_blocks_n(dd, 1)
                part = alloc.empty((nparts, 2, d.Cin), dtype=torch.float32, device=x.device)
                hook = lib.BnReduceHook(src.y.data_ptr(), d.Cin, src.mi.data_ptr(), d.Cin, src.gammas[0].data_ptr(), src.betas[0].data_ptr(),
                                        src.act, part.data_ptr())
                conv_dgrad_bnred(dy, w, dx, dd, hook, s, skip=sk, lds=lds)
                src.wrote(dx, (part, nparts))
            elif skip and dskip is not None:
                _dgrad_accumulate(dy, w, dx, d, dd, dskip, s)
            else:
                _conv_dgrad_any(dy, w, dx, dd, s)
        if both:
            _join_side(x.device)
        return dx, dw, dgamma, dbeta, None, None, None, (dout if has_res else None), None, None, None, None, None, None, None


def conv_bn_act(x, w, gamma, beta, rmean, rvar, nbt, stride=1, act=ACT_SILU, residual=None, training=True, eps=1e-3,
                momentum=0.03, skip=False, dest=None):
    return _ConvBnAct.apply(x, w, gamma, beta, rmean, rvar, nbt, residual, stride, act, training, eps, momentum, skip, dest)


def back_to_back(*ts):
    """Equally shaped dense tensors that lie one after the other in memory (pack_pair / fusion_ops.pack_qkv)."""
    a = ts[0]
    n = a.numel() * a.element_size()
    return all(t.shape == a.shape and t.dtype == a.dtype and t.data_ptr() == a.data_ptr() + i * n for i, t in enumerate(ts))


class _DualConvBnAct(Function):
    """C3's cv1 and cv2 (models/common.py:645-650 of the reference: two 1x1 Conv modules over the same input) as ONE GEMM with
    2*c_ output channels, one BatchNorm statistics pass and one normalise/activate pass, whose first half `a` feeds the
    bottleneck chain and whose second half `b` is written straight into the buffer cv3 reads (no torch.cat copy).  Backward:
    one BatchNorm backward over both halves (two incoming gradients), ONE input-gradient GEMM (no fan-out add), the two weight
    gradients separately (each parameter keeps its own .grad).  The caller guarantees that the two layers' parameters and
    buffers are back to back in memory (C3.packed())."""

    @staticmethod
    def forward(ctx, x, w1, w2, g1, b1, g2, b2, rm1, rv1, nbt1, nbt2, act, training, eps, momentum, dest):
        x, ldx = rows_of(x)
        w1, w2 = _ohwi(w1), _ohwi(w2)
        c_ = w1.shape[0]
        cout = 2 * c_
        d = _desc(x.shape, cout, 1, 1, ldx, cout)
        y = alloc.empty((d.N, d.Ho, d.Wo, cout), dtype=x.dtype, device=x.device)
        rows = d.N * d.Ho * d.Wo
        s = _stream()
        mi = _bn_forward(x, w1, y, d, cout, rows, training, eps, momentum, rm1, rv1, nbt1, nbt2, s)
        a = alloc.empty((d.N, d.Ho, d.Wo, c_), dtype=x.dtype, device=x.device)
        b = _dest_view(dest, a.shape)
        ldb = rows_of(b)[1]
        assert b.dtype == y.dtype
        _bn_act_fwd(y, cout, mi, g1, b1, None, 0, a, c_, b, ldb, c_, rows, cout, act, s)
        ctx.save_for_backward(x, w1, w2, y, mi, g1, b1, g2, b2)
        ctx.cfg = (d, act, training, c_)
        return a, b

    @staticmethod
    def backward(ctx, da, db):
        x, w1, w2, y, mi, g1, b1, g2, b2 = ctx.saved_tensors
        d, act, training, c_ = ctx.cfg
        cout, rows, s = 2 * c_, d.N * d.Ho * d.Wo, _stream()
        da, lda = rows_of(raw_cast(da, y.dtype))
        db, ldb = rows_of(raw_cast(db, y.dtype))
        dy = alloc.empty_like(y)
        dg1, dbt1, dg2, dbt2 = grad_like(g1), grad_like(b1), grad_like(g2), grad_like(b2)
        _bn_act_bwd(y, cout, da, lda, db, ldb, c_, mi, g1, b1, dy, (dg1, dbt1, dg2, dbt2), rows, cout, act, 0 if training else 1, s)
        both = OVERLAP_WGRAD and ctx.needs_input_grad[0]
        d1 = ConvDesc(d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, c_, 1, 1, 1, 0, d.ldx, cout)
        if ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and not GRAD_SLOTS and dy.dtype == torch.float32:
            # one (2c_, Cin) gradient matrix for the packed pair, each parameter's .grad a row block of it: one launch
            flat = alloc.empty(cout * d.Cin, dtype=w1.dtype, device=x.device)
            dd = ConvDesc(d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, cout, 1, 1, 1, 0, d.ldx, cout)
            _wgrad(dy, cout, x, d.ldx, w1, dd, overlap=both, out=(flat, None))
            dw1, dw2 = (flat[i * c_ * d.Cin:(i + 1) * c_ * d.Cin].as_strided(w1.shape, w1.stride()) for i in range(2))
        else:
            dw1 = _wgrad(dy[..., :c_], cout, x, d.ldx, w1, d1, overlap=both) if ctx.needs_input_grad[1] else None
            dw2 = _wgrad(dy[..., c_:], cout, x, d.ldx, w2, d1, overlap=both) if ctx.needs_input_grad[2] else None
        dx = None
        if ctx.needs_input_grad[0]:
            dx = alloc.empty((d.N, d.H, d.W, d.Cin), dtype=x.dtype, device=x.device)
            dd = ConvDesc(d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, cout, 1, 1, 1, 0, d.Cin, cout)
            _conv_dgrad_any(dy, w1, dx, dd, s)
        if both:
            _join_side(x.device)
        return dx, dw1, dw2, dg1, dbt1, dg2, dbt2, None, None, None, None, None, None, None, None, None


def dual_conv_bn_act(x, w1, w2, g1, b1, g2, b2, rm1, rv1, nbt1, nbt2, act, training, eps, momentum, dest):
    return _DualConvBnAct.apply(x, w1, w2, g1, b1, g2, b2, rm1, rv1, nbt1, nbt2, act, training, eps, momentum, dest)


class _CatAlias(Function):
    """The concat of two tensors that already ARE the two channel halves of one buffer: returns the buffer, copies nothing;
    the gradient goes back as two channel-slice views."""

    @staticmethod
    def forward(ctx, a, b, holder):
        cat = holder.t
        ca = a.shape[-1]
        assert a.data_ptr() == cat.data_ptr() and b.data_ptr() == cat.data_ptr() + cat.element_size() * ca and ca + b.shape[-1] == cat.shape[-1], \
            'cat_alias: the inputs are not the channel halves of the buffer'
        ctx.ca = ca
        return cat

    @staticmethod
    def backward(ctx, g):
        return g[..., :ctx.ca], g[..., ctx.ca:], None


def cat_alias(a, b, holder):
    return _CatAlias.apply(a, b, holder)


class _CatAliasN(Function):
    """Concat of tensors that already ARE consecutive channel slices of one buffer (the neck's Concat layers, whose producers
    wrote into the buffer: models/common.py:740-748 of the reference without the copy)."""

    @staticmethod
    def forward(ctx, holder, *xs):
        cat = holder.t
        off, es = 0, cat.element_size()
        for x in xs:
            assert x.data_ptr() == cat.data_ptr() + es * off and x.shape[:-1] == cat.shape[:-1], 'cat_alias: an input is not its slice of the buffer'
            off += x.shape[-1]
        assert off == cat.shape[-1]
        ctx.sizes = [x.shape[-1] for x in xs]
        return cat

    @staticmethod
    def backward(ctx, g):
        outs, off = [], 0
        for c in ctx.sizes:
            outs.append(g[..., off:off + c])
            off += c
        return (None, *outs)


def cat_alias_n(xs, holder):
    return _CatAliasN.apply(holder, *xs)


def pack_pair(model):
    """Re-seat cv1 / cv2 of every C3 (conv weights, BatchNorm weight / bias / running statistics / num_batches_tracked) on
    shared buffers, cv1's part first: same Parameters, same state_dict keys and values, but the two 1x1 convolutions over the
    same input now run as one GEMM (C3.forward checks the addresses on every call, so a model moved afterwards -- .to(), deep
    copy -- just takes the two-convolution path again).  Call before anything caches parameter addresses (optimizer pointer
    tables, gradient buckets).  Returns the number of C3 modules packed."""
    n = 0
    for m in model.modules():
        if type(m).__name__ != 'C3' or not hasattr(m.cv1, 'bn') or not hasattr(m.cv2, 'bn'):
            continue
        w1, w2 = m.cv1.conv.weight, m.cv2.conv.weight
        if w1.shape != w2.shape or w1.shape[2:] != (1, 1) or w1.grad is not None or w2.grad is not None:
            continue
        pairs = [(w1, w2)] + [(getattr(m.cv1.bn, k), getattr(m.cv2.bn, k))
                              for k in ('weight', 'bias', 'running_mean', 'running_var', 'num_batches_tracked')]
        for a, b in pairs:
            if back_to_back(a.data, b.data) and a.is_contiguous(memory_format=torch.channels_last if a.dim() == 4 else torch.contiguous_format):
                continue
            flat = torch.cat([a.data.reshape(-1), b.data.reshape(-1)]) if a.dim() != 4 else \
                torch.cat([a.data.permute(0, 2, 3, 1).reshape(-1), b.data.permute(0, 2, 3, 1).reshape(-1)])
            k = a.numel()
            for i, p in enumerate((a, b)):
                piece = flat[i * k:(i + 1) * k]
                if a.dim() == 4:      # (O,I,1,1) logical shape over OHWI memory
                    p.data = piece.view(a.shape[0], a.shape[2], a.shape[3], a.shape[1]).permute(0, 3, 1, 2)
                else:
                    p.data = piece.view(a.shape)
        n += 1
    return n


class _ConvBias(Function):
    """conv(x, w) [+ bias] without normalisation: Detect heads, the CEM stencil bank, FFM 1x1 gates, nn.Linear."""

    @staticmethod
    def forward(ctx, x, w, bias, stride):
        ctx.src = x.dtype
        if x.dtype == BF16 and (w.shape[0] % 4 != 0 or x.shape[-1] % 4 != 0):
            x = raw_cast(x, torch.float32)       # (a Detect head has 3*(nc+5) = 33 columns: the bf16 kernels want multiples of 4)
        x, ldx = rows_of(x)
        w = _ohwi(w)
        if w.dim() == 2:
            cout, k = w.shape[0], 1
            xs = (_nrows(x), 1, 1, x.shape[-1])
        else:
            cout, k = w.shape[0], w.shape[2]
            xs = tuple(x.shape)
        d = _desc(xs, cout, k, stride, ldx, cout)
        oshape = (*x.shape[:-1], cout) if w.dim() == 2 else (d.N, d.Ho, d.Wo, cout)
        y = alloc.empty(oshape, dtype=x.dtype, device=x.device)
        if x.dtype == BF16:
            # bf16 storage: the map is bf16, this layer's OUTPUT (a Detect head: what the loss reads) is handed on as fp32
            nb = fwd_plan_bf16(d)[0]
            ws = zeroed_scratch(nb, x.device, _stream())
            lib.conv_fwd_bf16(x.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(), None, None,
                              ws.data_ptr(), nb, d, _stream())
            y = raw_cast(y, torch.float32)
        else:
            conv_fwd(x, w, bias, y, None, d, _stream())
        ctx.save_for_backward(x, w)
        ctx.bias = bias            # (only its address is needed in backward: where the bias gradient goes)
        ctx.cfg = (d, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        d, has_bias = ctx.cfg
        dy, lddy = rows_of(raw_cast(dy, x.dtype))
        s = _stream()
        dd = ConvDesc(d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout, d.KH, d.KW, d.stride, d.pad, d.Cin, lddy)
        dx = None
        both = OVERLAP_WGRAD and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]
        dw = db = None
        want_db = has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            dwd = ConvDesc(d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout, d.KH, d.KW, d.stride, d.pad, d.ldx, lddy)
            if want_db:   # the bias gradient falls out of the weight-gradient kernel's dy tiles
                dw, db = _wgrad(dy, lddy, x, d.ldx, w, dwd, overlap=both, want_bias=True, bias=ctx.bias)
            else:
                dw = _wgrad(dy, lddy, x, d.ldx, w, dwd, overlap=both)
        if ctx.needs_input_grad[0]:
            dx = alloc.empty(tuple(x.shape), dtype=x.dtype, device=x.device)
            _conv_dgrad_any(dy, w, dx, dd, s)
        if want_db and db is None:
            rows = d.N * d.Ho * d.Wo
            db = grad_like(ctx.bias)
            part = scratch(lib.bn_bwd_parts(rows) * d.Cout, x.device)
            lib.colsum(dy.data_ptr(), lddy, rows, d.Cout, part.data_ptr(), db.data_ptr(), s)
        if both:
            _join_side(x.device)
        if dx is not None and dx.dtype != ctx.src:
            dx = raw_cast(dx, ctx.src)
        return dx, dw, db, None


def conv_bias(x, w, bias=None, stride=1):
    return _ConvBias.apply(x, w, bias, stride)


def conv_bias_act(x, w, bias, stride=1, act=ACT_SILU, residual=None):
    """Inference form of Conv after Model.fuse() (models/common.py:124-125): act(conv(x, w) + bias) [+ residual] in one
    kernel.  No autograd (the reference's fused conv is requires_grad_(False) as well)."""
    assert not (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad)), \
        'fused Conv+BN layers are inference-only (utils/torch_utils.py:189)'
    x, ldx = rows_of(x)
    w = _ohwi(w)
    cout, k = w.shape[0], w.shape[2]
    d = _desc(x.shape, cout, k, stride, ldx, cout)
    y = alloc.empty((d.N, d.Ho, d.Wo, cout), dtype=x.dtype, device=x.device)
    ldr = 0
    if residual is not None:
        residual, ldr = rows_of(residual)
    s = _stream()
    nb = lib.conv_fwd_workspace(d)
    ws = zeroed_scratch(nb, x.device, s) if nb else None
    lib.conv_bias_act_fwd(x.data_ptr(), w.data_ptr(), bias.data_ptr(), residual.data_ptr() if residual is not None else None,
                          ldr, act, y.data_ptr(), ws.data_ptr() if nb else None, nb, d, s)
    return y


def detect_decode(levels, strides, anchor_grid, no):
    """Detect eval decode (models/yolo_test.py:57-68): levels = permuted head outputs (B,na,ny,nx,no) -> (B, sum rows, no)."""
    b, na = levels[0].shape[0], levels[0].shape[1]
    rows = [x.shape[1] * x.shape[2] * x.shape[3] for x in levels]
    z = alloc.empty((b, sum(rows), no), dtype=torch.float32, device=levels[0].device)
    off = 0
    s = _stream()
    for i, x in enumerate(levels):
        x = x.contiguous()
        ag = anchor_grid[i].reshape(-1).contiguous()
        lib.detect_decode(x.data_ptr(), z.data_ptr(), b, na, x.shape[2], x.shape[3], no, sum(rows), off, float(strides[i]),
                          ag.data_ptr(), s)
        off += rows[i]
    return z


def nms(pred, conf_thres, iou_thres, classes, agnostic, multi_label, max_det=300, max_wh=4096.0):
    """utils/general.py:486-580 on the device; classes = iterable of kept class ids or None; returns
    (out (B,max_det,6), nout (B,) int32)."""
    pred = pred.contiguous()
    assert pred.dtype == torch.float32 and pred.is_cuda and pred.dim() == 3
    b, r, no = pred.shape
    nb = lib.nms_workspace(b, r, no - 5, int(multi_label))
    ws = alloc.empty(nb, dtype=torch.uint8, device=pred.device)
    out = torch.zeros((b, max_det, 6), dtype=torch.float32, device=pred.device)
    nout = torch.zeros(b, dtype=torch.int32, device=pred.device)
    allow = None
    if classes is not None:
        allow = torch.zeros(no - 5, dtype=torch.uint8)
        allow[[int(c) for c in classes if 0 <= int(c) < no - 5]] = 1
        allow = allow.to(pred.device)
    lib.nms(pred.data_ptr(), b, r, no - 5, conf_thres, iou_thres, allow.data_ptr() if allow is not None else None,
            int(agnostic), int(multi_label), max_det, max_wh, ws.data_ptr(), nb, out.data_ptr(), nout.data_ptr(), _stream())
    return out, nout


def linear(x, w, bias=None):
    return _ConvBias.apply(x, w, bias, 1)


class _Add(Function):
    """a + b; dest=(Dest, channel): write the sum into that channel slice of a wider buffer (a neck Concat's buffer)."""

    @staticmethod
    def forward(ctx, a, b, dest):
        a, lda = rows_of(a)
        b, ldb = rows_of(b)
        out = _dest_view(dest, a.shape) if dest is not None else alloc.empty(tuple(a.shape), dtype=a.dtype, device=a.device)
        c = a.shape[-1]
        assert a.dtype == b.dtype == out.dtype
        (lib.add_bf16 if a.dtype == BF16 else lib.add)(a.data_ptr(), lda, b.data_ptr(), ldb, out.data_ptr(), rows_of(out)[1], _nrows(a), c, _stream())
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g, None


def add(a, b, dest=None):
    return _Add.apply(a, b, dest)


class _Concat(Function):
    """Channel concat of NHWC tensors; the backward hands out channel-slice views (no copy)."""

    @staticmethod
    def forward(ctx, *xs):
        ctot = sum(x.shape[-1] for x in xs)
        out = alloc.empty((*xs[0].shape[:-1], ctot), dtype=xs[0].dtype, device=xs[0].device)
        off = 0
        s = _stream()
        es = out.element_size()
        cp = lib.copy2d_bf16 if out.dtype == BF16 else lib.copy2d
        for x in xs:
            x, ld = rows_of(raw_cast(x, out.dtype))
            c = x.shape[-1]
            cp(x.data_ptr(), ld, out.data_ptr() + es * off, ctot, _nrows(x), c, s)
            off += c
        ctx.sizes = [x.shape[-1] for x in xs]
        return out

    @staticmethod
    def backward(ctx, g):
        outs, off = [], 0
        for c in ctx.sizes:
            outs.append(g[..., off:off + c])
            off += c
        return tuple(outs)


def concat(xs):
    return _Concat.apply(*xs)


class _Upsample2x(Function):
    """nearest x2.  skip=True also returns x itself (see _ConvBnAct): the map's other consumer (a later Concat of the PANet head)
    takes that alias, so its gradient arrives here as `gskip` and is added inside the backward kernel instead of by the autograd
    engine (an ATen add); the incoming gradient is read through its row stride (a channel slice of a Concat gradient)."""

    @staticmethod
    def forward(ctx, x, skip, dest):
        ctx.set_materialize_grads(False)
        x_in = x
        x, ld = rows_of(x)
        n, h, w, c = x.shape
        if x.dtype == BF16:
            assert dest is None
            if ld != c:
                x = x.contiguous()
            y = alloc.empty((n, 2 * h, 2 * w, c), dtype=x.dtype, device=x.device)
            lib.upsample2x_bf16(x.data_ptr(), y.data_ptr(), n, h, w, c, _stream())
        else:      # rows with strides on both sides: the input may be a slice of a Concat buffer, the output written into one
            y = _dest_view(dest, (n, 2 * h, 2 * w, c)) if dest is not None else alloc.empty((n, 2 * h, 2 * w, c), dtype=x.dtype, device=x.device)
            lib.upsample2x_ld(x.data_ptr(), ld, y.data_ptr(), rows_of(y)[1], n, h, w, c, _stream())
        ctx.shape = (n, h, w, c)
        return (y, x_in) if skip else y

    @staticmethod
    def backward(ctx, g, gskip=None):
        n, h, w, c = ctx.shape
        if g is None:
            return gskip, None, None
        dx = alloc.empty((n, h, w, c), dtype=g.dtype, device=g.device)
        if g.dtype == BF16:
            g = g.contiguous()
            lib.upsample2x_bwd_bf16(g.data_ptr(), dx.data_ptr(), n, h, w, c, _stream())
            if gskip is not None:
                gs, lds = rows_of(raw_cast(gskip, BF16))
                lib.add_bf16(dx.data_ptr(), c, gs.data_ptr(), lds, dx.data_ptr(), c, _nrows(dx), c, _stream())
            return dx, None, None
        g, ldg = rows_of(g)
        gs, lds = rows_of(gskip) if gskip is not None else (None, 0)
        lib.upsample2x_bwd_acc(g.data_ptr(), ldg, gs.data_ptr() if gs is not None else None, lds, dx.data_ptr(), n, h, w, c, _stream())
        return dx, None, None


def upsample2x(x, skip=False, dest=None):
    return _Upsample2x.apply(x, skip, dest)


class _SppPool(Function):
    """x -> cat(x, mp5, mp9, mp13) along channels (cascaded 5x5 max-pools)."""

    @staticmethod
    def forward(ctx, x):
        ctx.src = x.dtype
        x, ld = rows_of(raw_cast(x, torch.float32))      # (bf16 storage: a P5-sized map; the pooling kernels are fp32)
        n, h, w, c = x.shape
        out = alloc.empty((n, h, w, 4 * c), dtype=x.dtype, device=x.device)
        lib.spp_pool_fwd(x.data_ptr(), ld, out.data_ptr(), 4 * c, n, h, w, c, _stream())
        ctx.save_for_backward(x)
        return raw_cast(out, ctx.src)

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        x, ld = rows_of(x)
        g, ldg = rows_of(raw_cast(g, torch.float32))
        n, h, w, c = x.shape
        dx = alloc.empty((n, h, w, c), dtype=x.dtype, device=x.device)
        lib.spp_pool_bwd(x.data_ptr(), ld, g.data_ptr(), ldg, dx.data_ptr(), c, n, h, w, c, _stream())
        return raw_cast(dx, ctx.src)


def spp_pool(x):
    return _SppPool.apply(x)


def u8_pair_to_nhwc(imgs_u8):
    """Loader batch uint8 (N,6,H,W) -> (rgb, ir) fp32 NHWC in [0,1], marked so that Model.forward skips its own
    NCHW->NHWC pass (train.py:743-745 fused; bit-identical to `.float() / 255` + slice + re-layout)."""
    assert imgs_u8.dtype == torch.uint8 and imgs_u8.dim() == 4 and imgs_u8.shape[1] == 6 and imgs_u8.is_cuda
    x = imgs_u8.contiguous()
    n, _, h, w = x.shape
    rgb = alloc.empty((n, h, w, 3), dtype=torch.float32, device=x.device)
    ir = alloc.empty_like(rgb)
    lib.u8_pair_to_nhwc(x.data_ptr(), rgb.data_ptr(), ir.data_ptr(), n, h, w, _stream())
    rgb.mmi_nhwc = ir.mmi_nhwc = True
    return rgb, ir


def nchw_to_nhwc(x):
    """Boundary op (no gradient: the model inputs are images).  Accepts strided NCHW views (train.py:744-745); tensors
    that u8_pair_to_nhwc produced are NHWC already."""
    if getattr(x, 'mmi_nhwc', False):
        return x
    assert x.dim() == 4 and x.dtype == torch.float32 and x.is_cuda
    n, c, h, w = x.shape
    y = alloc.empty((n, h, w, c), dtype=x.dtype, device=x.device)
    sn, sc, sh, sw = x.stride()
    lib.nchw_to_nhwc(x.data_ptr(), sn, sc, sh, sw, y.data_ptr(), n, c, h, w, _stream())
    return y


class _SpaceToDepth(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        n, h, w, c = x.shape
        y = alloc.empty((n, h // 2, w // 2, 4 * c), dtype=x.dtype, device=x.device)
        lib.space_to_depth(x.data_ptr(), y.data_ptr(), n, h, w, c, 0, _stream())
        ctx.shape = (n, h, w, c)
        return y

    @staticmethod
    def backward(ctx, g):
        n, h, w, c = ctx.shape
        g = g.contiguous()
        dx = alloc.empty((n, h, w, c), dtype=g.dtype, device=g.device)
        lib.space_to_depth(g.data_ptr(), dx.data_ptr(), n, h, w, c, 1, _stream())
        return dx


def space_to_depth(x):
    return _SpaceToDepth.apply(x)


class _HeadPermute(Function):
    """(B,ny,nx,na*no) -> (B,na,ny,nx,no) contiguous, as Detect returns it (yolo_test.py:54-55)."""

    @staticmethod
    def forward(ctx, x, na):
        x = x.contiguous()
        b, ny, nx, c = x.shape
        no = c // na
        y = alloc.empty((b, na, ny, nx, no), dtype=x.dtype, device=x.device)
        lib.head_permute(x.data_ptr(), y.data_ptr(), b, na, no, ny * nx, 0, _stream())
        ctx.cfg = (b, na, no, ny, nx)
        return y

    @staticmethod
    def backward(ctx, g):
        b, na, no, ny, nx = ctx.cfg
        g = g.contiguous()
        dx = alloc.empty((b, ny, nx, na * no), dtype=g.dtype, device=g.device)
        lib.head_permute(g.data_ptr(), dx.data_ptr(), b, na, no, ny * nx, 1, _stream())
        return dx, None


def head_permute(x, na):
    return _HeadPermute.apply(x, na)


class _SobelAdd(Function):
    """t = r + EnhanceConv2d(r) for the fixed CEM stencil bank: factor[o]*stencil_{o%8}(sum_c r_c) + bias[o] (+ r)."""

    @staticmethod
    def forward(ctx, r, factor, bias):
        r, ldr = rows_of(r)
        n, h, w, c = r.shape
        t = alloc.empty((n, h, w, c), dtype=r.dtype, device=r.device)
        chansum = alloc.empty((n, h, w), dtype=r.dtype, device=r.device)
        f = factor.reshape(-1).contiguous()
        lib.sobel_add_fwd(r.data_ptr(), ldr, f.data_ptr(), bias.data_ptr(), chansum.data_ptr(), t.data_ptr(), c, n, h, w, c,
                          _stream())
        ctx.save_for_backward(chansum, f)
        ctx.cfg = (n, h, w, c, tuple(factor.shape))
        return t

    @staticmethod
    def backward(ctx, dt):
        chansum, f = ctx.saved_tensors
        n, h, w, c, fshape = ctx.cfg
        dt, ldd = rows_of(dt)
        dr = alloc.empty((n, h, w, c), dtype=dt.dtype, device=dt.device)
        df = alloc.empty(c, dtype=dt.dtype, device=dt.device)
        db = alloc.empty(c, dtype=dt.dtype, device=dt.device)
        nbytes = lib.sobel_add_bwd_workspace(n, h, w, c)
        ws = scratch(nbytes // 4 + 4, dt.device, slot=5)
        lib.sobel_add_bwd(dt.data_ptr(), ldd, chansum.data_ptr(), f.data_ptr(), dr.data_ptr(), c, df.data_ptr(),
                          db.data_ptr(), ws.data_ptr(), n, h, w, c, _stream())
        return dr, df.view(fshape), db


def sobel_add(r, factor, bias):
    return _SobelAdd.apply(r, factor, bias)


CEM_FUSED = __import__("os").environ.get("MMIDET_CEM_FUSED", "1") != "0"      # A/B: the fused CEM forward
CEM_WGRAD_LATE = __import__("os").environ.get("MMIDET_CEM_WGRAD_LATE", "0") == "1"   # A/B: conv3's wgrad after the critical chain (no gain)
CEM_BWD_FUSED = __import__("os").environ.get("MMIDET_CEM_BWD_FUSED", "1") != "0"   # A/B: conv3 dgrad + stencil-bank backward in one kernel
# ... with BatchNorm2's backward reduction riding along.  Round 2: 48 more accumulators cost the kernel a workgroup per CU (233 VGPRs)
# and the module was slower WITH the apply pass still reading dr and y2 (3.69 -> 3.92 ms, profiles/r02_cem_backward_middle.txt).  Round 4:
# the apply pass lives in conv2's weight-gradient loader, so the reduction pass was the last separate reader of (dr, y2) -- without it
# the module takes 2.32 instead of 2.42 ms (profiles/r04_cem_bwd_forms.txt; a lower-register form of the kernel's last phase, a thread
# per (position, 4-channel group), measured the same 2.33 and was dropped).  On.
CEM_BWD_BN = __import__("os").environ.get("MMIDET_CEM_BWD_BN", "1") != "0"
# Training forward as conv2 (stored, with BN2's statistics) + the fused kernel reading y2: conv2 evaluated once instead of 2.56 times
# per pixel; same arithmetic on the same y2 (tests/test_cem_gpu.py).  "0": statistics pre-pass + recomputing fused kernel (rounds 2-3).
CEM_TWO_PASS = __import__("os").environ.get("MMIDET_CEM_TWO_PASS", "1") != "0"
# BatchNorm2's backward apply pass inside conv2's weight-gradient loader (the image takes no gradient, so dy2 has no other reader):
# dy2 never reaches HBM.  "0": the separate apply pass + the generic small-channel weight gradient.
CEM_WGRAD_BN = __import__("os").environ.get("MMIDET_CEM_WGRAD_BN", "1") != "0"


class _CemFused(Function):
    """AdaptiveModule3 with the reference's stencil bank (models/common.py:751-803, 806-911) as ONE forward kernel behind a
    statistics pre-pass: x -> conv2 -> BN2 + LeakyReLU -> r + stencil bank -> conv3 with r and t in LDS (csrc/cem.hip::
    cem_fused_fwd_kernel), then BN3 + LeakyReLU + x.  y2, t and the channel-sum map are written on the way, so the backward
    is the unfused chain's: BN3 backward, conv3 wgrad / dgrad, stencil-bank backward, BN2 backward, conv2 wgrad."""

    @staticmethod
    def forward(ctx, x, w2, g2, b2, rm2, rv2, nbt2, factor, sbias, w3, g3, b3, rm3, rv3, nbt3, training, eps, momentum):
        x, ldx = rows_of(x)
        w2, w3 = _ohwi(w2), _ohwi(w3)
        n, h, w, _ = x.shape
        dev, s = x.device, _stream()
        rows = n * h * w
        nblk = lib.cem_blocks(n, h, w)
        f = factor.reshape(-1).contiguous()
        mi2 = alloc.empty(48, dtype=torch.float32, device=dev)
        mi3 = alloc.empty(6, dtype=torch.float32, device=dev)
        keep = any(ctx.needs_input_grad)      # (grad mode itself is off inside forward)
        y2 = alloc.empty((n, h, w, 24), dtype=torch.float32, device=dev) if keep else None
        t = alloc.empty((n, h, w, 24), dtype=torch.float32, device=dev) if keep else None
        cs = alloc.empty((n, h, w), dtype=torch.float32, device=dev) if keep else None
        y3 = alloc.empty((n, h, w, 3), dtype=torch.float32, device=dev)
        part3 = scratch((nblk + 64) * 2 * 3, dev, slot=6) if training else None
        two_pass = training and keep and CEM_TWO_PASS
        if two_pass:
            nb2 = lib.cem_conv2_fwd_blocks(n, h, w)
            part = scratch((nb2 + 64) * 2 * 24, dev)
            lib.cem_conv2_fwd(x.data_ptr(), ldx, w2.data_ptr(), y2.data_ptr(), part.data_ptr(), n, h, w, s)
            lib.bn_finalize(part.data_ptr(), nb2, rows, 24, eps, momentum, rm2.data_ptr(), rv2.data_ptr(), nbt2.data_ptr(), mi2.data_ptr(), s)
            lib.cem_fwd_from_y2(y2.data_ptr(), mi2.data_ptr(), g2.data_ptr(), b2.data_ptr(), f.data_ptr(), sbias.data_ptr(), w3.data_ptr(),
                                t.data_ptr(), cs.data_ptr(), y3.data_ptr(), part3.data_ptr(), n, h, w, s)
        else:
            if training:
                part = scratch((nblk + 64) * 2 * 24, dev)
                lib.cem_conv2_stats(x.data_ptr(), ldx, w2.data_ptr(), part.data_ptr(), n, h, w, s)
                lib.bn_finalize(part.data_ptr(), nblk, rows, 24, eps, momentum, rm2.data_ptr(), rv2.data_ptr(), nbt2.data_ptr(), mi2.data_ptr(), s)
            else:
                lib.bn_eval_stats(rm2.data_ptr(), rv2.data_ptr(), 24, eps, mi2.data_ptr(), s)
            lib.cem_fused_fwd(x.data_ptr(), ldx, w2.data_ptr(), mi2.data_ptr(), g2.data_ptr(), b2.data_ptr(), f.data_ptr(), sbias.data_ptr(),
                              w3.data_ptr(), y2.data_ptr() if keep else None, t.data_ptr() if keep else None, cs.data_ptr() if keep else None,
                              y3.data_ptr(), part3.data_ptr() if training else None, n, h, w, s)
        if training:
            lib.bn_finalize(part3.data_ptr(), nblk, rows, 3, eps, momentum, rm3.data_ptr(), rv3.data_ptr(), nbt3.data_ptr(), mi3.data_ptr(), s)
        else:
            lib.bn_eval_stats(rm3.data_ptr(), rv3.data_ptr(), 3, eps, mi3.data_ptr(), s)
        out = alloc.empty_like(y3)
        lib.bn_act_fwd(y3.data_ptr(), 3, mi3.data_ptr(), g3.data_ptr(), b3.data_ptr(), x.data_ptr(), ldx, out.data_ptr(), 3, rows, 3,
                       ACT_LEAKY, s)
        if keep:
            ctx.save_for_backward(x, w2, y2, mi2, g2, b2, cs, f, t, w3, y3, mi3, g3, b3, sbias)
        ctx.cfg = (n, h, w, ldx, training, tuple(factor.shape))
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w2, y2, mi2, g2, b2, cs, f, t, w3, y3, mi3, g3, b3, sbias = ctx.saved_tensors
        n, h, w, ldx, training, fshape = ctx.cfg
        dout, ldd = rows_of(dout)
        dev, s = x.device, _stream()
        rows, frozen = n * h * w, 0 if training else 1
        # BN3 + LeakyReLU (+ the residual, whose gradient is dout itself)
        dy3 = alloc.empty_like(y3)
        dg3, db3 = grad_like(g3), grad_like(b3)
        _bn_act_bwd(y3, 3, dout, ldd, None, 0, 3, mi3, g3, b3, dy3, (dg3, db3, None, None), rows, 3, ACT_LEAKY, frozen, s)
        # conv3: t (24) -> y3 (3)
        d3 = ConvDesc(n, h, w, 24, h, w, 3, 3, 3, 1, 1, 24, 3)
        # conv3's weight gradient is independent of everything below.  Launched here it shares the chip with the critical chain
        # (dr -> BatchNorm2 backward -> conv2's weight gradient) and stretches it: cem_bwd_mid 0.29 -> 0.57 ms, the BatchNorm
        # reduction 0.31 -> 0.58 ms in the step's trace.  Launched LAST it runs beside conv2's weight gradient on the other wgrad
        # stream, where nothing is waiting for either.  Measured in the step (three interleaved pairs): 122.46 vs 122.55 ms, i.e. nothing --
        # MMIDET_CEM_WGRAD_LATE=1 keeps the variant, the default is the original order.
        dw3 = None if CEM_WGRAD_LATE else _wgrad(dy3, 3, t, 24, w3, d3, overlap=OVERLAP_WGRAD)
        dr = alloc.empty((n, h, w, 24), dtype=torch.float32, device=dev)
        df = alloc.empty(24, dtype=torch.float32, device=dev)
        dsb = grad_like(sbias)
        if CEM_BWD_FUSED:
            # conv3's input gradient and the stencil bank's backward in one kernel: dt never reaches HBM (csrc/cem.hip)
            nbytes = lib.cem_bwd_mid_workspace(n, h, w)
            ws = scratch(nbytes // 4 + 4, dev, slot=5)
            bnpart = scratch(lib.cem_bwd_mid_blocks(n, h, w) * 48 + 64, dev, slot=7) if CEM_BWD_BN else None
            lib.cem_bwd_mid(dy3.data_ptr(), w3.data_ptr(), cs.data_ptr(), f.data_ptr(), dr.data_ptr(), df.data_ptr(), dsb.data_ptr(),
                            ws.data_ptr(), *((y2.data_ptr(), mi2.data_ptr(), g2.data_ptr(), b2.data_ptr(), bnpart.data_ptr())
                                             if CEM_BWD_BN else (None,) * 5), n, h, w, s)
        else:
            dt = alloc.empty((n, h, w, 24), dtype=torch.float32, device=dev)
            conv_dgrad(dy3, w3, dt, ConvDesc(n, h, w, 24, h, w, 3, 3, 3, 1, 1, 24, 3), s)
            nbytes = lib.sobel_add_bwd_workspace(n, h, w, 24)
            ws = scratch(nbytes // 4 + 4, dev, slot=5)
            lib.sobel_add_bwd(dt.data_ptr(), 24, cs.data_ptr(), f.data_ptr(), dr.data_ptr(), 24, df.data_ptr(), dsb.data_ptr(),
                              ws.data_ptr(), n, h, w, 24, s)
        # BN2 + LeakyReLU
        dg2, db2 = grad_like(g2), grad_like(b2)
        if CEM_WGRAD_BN and not ctx.needs_input_grad[0]:
            # sums only, then the apply pass inside conv2's weight-gradient loader (csrc/cem.hip, BNF)
            if CEM_BWD_FUSED and CEM_BWD_BN:      # ... which came out of cem_bwd_mid: fold its partials
                lib.bn_act_bwd_apply(y2.data_ptr(), 24, dr.data_ptr(), 24, mi2.data_ptr(), g2.data_ptr(), b2.data_ptr(), bnpart.data_ptr(),
                                     lib.cem_bwd_mid_blocks(n, h, w), None, 24, dg2.data_ptr(), db2.data_ptr(), rows, 24, ACT_LEAKY, frozen, s)
            else:
                nbw = bn_bwd_ws(rows, 24)
                ws2 = zeroed_scratch(nbw, dev, s, tag='bn')
                lib.bn_act_bwd(y2.data_ptr(), 24, dr.data_ptr(), 24, None, 0, 24, mi2.data_ptr(), g2.data_ptr(), b2.data_ptr(), ws2.data_ptr(),
                               nbw, None, 24, dg2.data_ptr(), db2.data_ptr(), None, None, rows, 24, ACT_LEAKY, frozen, s)
            dw2 = grad_like(w2)
            nb = lib.cem_conv2_wgrad_bn_workspace(n, h, w)

            def launch(st, wsp):
                lib.cem_conv2_wgrad_bn(dr.data_ptr(), y2.data_ptr(), x.data_ptr(), ldx, mi2.data_ptr(), g2.data_ptr(), b2.data_ptr(),
                                       dg2.data_ptr(), db2.data_ptr(), frozen, dw2.data_ptr(), wsp, nb, n, h, w, st)
            if OVERLAP_WGRAD:
                main, side = torch.cuda.current_stream(), _side_stream(dev)
                side.wait_stream(main)
                launch(side.cuda_stream, scratch(nb // 4, dev, slot=8, stream=side.cuda_stream).data_ptr())
                if DEFER_JOIN:
                    _pending.append((dr, y2, x))      # (not dg2 / db2: a second owner makes AccumulateGrad clone a bucket view instead of adopting it)
                    _pending_sides[side.cuda_stream] = side
            else:
                launch(s, scratch(nb // 4, dev, slot=8).data_ptr())
            if dw3 is None:
                dw3 = _wgrad(dy3, 3, t, 24, w3, d3, overlap=OVERLAP_WGRAD)
            if OVERLAP_WGRAD:
                _join_side(dev)
            return (None, dw2, dg2, db2, None, None, None, df.view(fshape), dsb, dw3, dg3, db3, None, None, None, None, None, None)
        dy2 = alloc.empty_like(y2)
        if CEM_BWD_FUSED and CEM_BWD_BN:     # the reduction came out of cem_bwd_mid: fold its partials, then the apply pass
            lib.bn_act_bwd_apply(y2.data_ptr(), 24, dr.data_ptr(), 24, mi2.data_ptr(), g2.data_ptr(), b2.data_ptr(), bnpart.data_ptr(),
                                 lib.cem_bwd_mid_blocks(n, h, w), dy2.data_ptr(), 24, dg2.data_ptr(), db2.data_ptr(), rows, 24, ACT_LEAKY,
                                 frozen, s)
        else:
            _bn_act_bwd(y2, 24, dr, 24, None, 0, 24, mi2, g2, b2, dy2, (dg2, db2, None, None), rows, 24, ACT_LEAKY, frozen, s)
        # conv2: x (3) -> y2 (24)
        d2 = ConvDesc(n, h, w, 3, h, w, 24, 3, 3, 1, 1, ldx, 24)
        dw2 = _wgrad(dy2, 24, x, ldx, w2, d2, overlap=OVERLAP_WGRAD)
        if dw3 is None:
            dw3 = _wgrad(dy3, 3, t, 24, w3, d3, overlap=OVERLAP_WGRAD)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = alloc.empty((n, h, w, 3), dtype=torch.float32, device=dev)
            conv_dgrad(dy2, w2, dx, ConvDesc(n, h, w, 3, h, w, 24, 3, 3, 1, 1, 3, 24), s)
            lib.add(dx.data_ptr(), 3, dout.data_ptr(), ldd, dx.data_ptr(), 3, rows, 3, s)
        if OVERLAP_WGRAD:
            _join_side(dev)
        return (dx, dw2, dg2, db2, None, None, None, df.view(fshape), dsb, dw3, dg3, db3, None, None, None, None, None, None)


def cem_fused(x, w2, bn2, factor, sbias, w3, bn3):
    return _CemFused.apply(x, w2, bn2.weight, bn2.bias, bn2.running_mean, bn2.running_var, bn2.num_batches_tracked, factor, sbias, w3,
                           bn3.weight, bn3.bias, bn3.running_mean, bn3.running_var, bn3.num_batches_tracked, bn2.training, bn2.eps,
                           bn2.momentum)
