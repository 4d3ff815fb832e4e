"""Twin-lane autograd glue: the RGB and IR backbones of the two-stream model as ONE sequence of launches.

The reference walks its two backbones layer by layer as separate module chains (models/yolo_test.py:162-273; rows 0-2 / 3-5,
9-10 / 11-12, ... of the default YAML).  Here the two lanes' activations are one 5-D tensor (N,H,W,2,C) -- lane g owns the channel
block [g*C, (g+1)*C) of every pixel row -- and every layer pair runs as one set of launches over it: the GEMMs (forward,
input-gradient, weight-gradient) carry both lanes' problems in one grid (include/mmidet_hip.h, "twin launches"), everything
per-channel around them (BatchNorm statistics, normalise + activation + residual, their backward) runs once over the 2C channels
with a channel map (mmi_bn_map) that tells it which BatchNorm module a channel belongs to and where its output lives.
A twin tensor may be a strided view (lane stride != C, row stride != 2C): the C3 concat buffer (N,H,W,2,2c_) is written by its
producers in place and read back as views, as in the single-lane form (ops._CatAlias).
"""
import torch
from torch.autograd import Function

from . import alloc, lib, ops
from .lib import ConvDesc
from .ops import _stream, grad_like, scratch, zeroed_scratch

_plan = {}


def layout(t):
    """(row stride, lane stride) in elements of a twin tensor (N,H,W,2,C) whose channel axis is contiguous and whose pixel rows
    are uniformly strided."""
    assert t.dim() == 5 and t.shape[3] == 2 and (t.stride(4) == 1 or t.shape[4] == 1), 'not a twin tensor: %s' % (tuple(t.shape),)
    ld, ls = t.stride(2), t.stride(3)
    n, h, w = t.shape[:3]
    if w == 1:
        ld = t.stride(1) if h > 1 else (t.stride(0) if n > 1 else 2 * t.shape[4])
    assert (h == 1 or t.stride(1) == w * ld) and (n == 1 or t.stride(0) == h * w * ld), 'twin tensor rows are not uniformly strided'
    return ld, ls


def dense(t):
    """The twin tensor as (N,H,W,2,C) contiguous memory (a copy only if it is a strided view)."""
    return t if t.is_contiguous() else t.contiguous()


def _lane_ptrs(t, ls=None, off=0):
    """ctypes pair of the two lanes' base addresses (+ `off` elements)."""
    if ls is None:
        ls = layout(t)[1]
    p = t.data_ptr() + 4 * off
    return lib.ptr_pair(p, p + 4 * ls)


def _pair(a, b):
    return lib.ptr_pair(a.data_ptr() if a is not None else None, b.data_ptr() if b is not None else None)


def _plans(d, key):
    """(fwd workspace, row blocks, dgrad workspace, wgrad workspace) of a twin launch of shape d, cached per shape."""
    v = _plan.get(key)
    if v is None:
        v = _plan[key] = (lib.conv_fwd_workspace_n(d, 2), lib.conv_fwd_row_blocks_n(d, 2))
    return v


def _ws(nbytes, device, s, tag):
    return zeroed_scratch(nbytes + 256, device, s, tag=tag) if nbytes else None


def _aligned(ws):
    """256-byte aligned address inside a workspace tensor allocated with 256 spare bytes."""
    if ws is None:
        raise ValueError('this shape has no twin-launch form (mmi_conv_*_workspace_n returned 0): Conv.twin_ok must route it to the lane form')
    p = ws.data_ptr()
    return (p + 255) & ~255


def conv_bn_fwd2(x, ldx, lsx, wa, wb, y, cout, k, stride, bns, eps, momentum, training, s):
    """y (N,Ho,Wo,2,cout) = conv of both lanes of x; returns mean_invstd (2, 2*cout) of the 2*cout channels.
    bns = ((running_mean, running_var, nbt, nbt2 or None) per lane)."""
    n, h, w = x.shape[:3]
    cin = x.shape[4]
    d = ops._desc((n, h, w, cin), cout, k, stride, ldx, 2 * cout)
    dev = x.device
    mi = alloc.empty(4 * cout, dtype=torch.float32, device=dev)
    if not training:
        for g, wt in enumerate((wa, wb)):
            ops.conv_fwd_raw(x.data_ptr() + 4 * g * lsx, wt.data_ptr(), y.data_ptr() + 4 * g * cout, d, dev, s)
            rm, rv = bns[g][0], bns[g][1]
            tmp = alloc.empty(2 * cout, dtype=torch.float32, device=dev)
            lib.bn_eval_stats(rm.data_ptr(), rv.data_ptr(), cout, eps, tmp.data_ptr(), s)
            mi[g * cout:(g + 1) * cout].copy_(tmp[:cout])
            mi[2 * cout + g * cout:2 * cout + (g + 1) * cout].copy_(tmp[cout:])
        return mi, d
    nb, nrb = _plans(d, ('f',) + ops._desc_key(d))
    ws = _ws(nb, dev, s, 'twin')
    per = (nrb + 64) * 2 * cout
    part = scratch(2 * per, dev, slot=11)
    stats = lib.BnStatsPair()
    for g in range(2):
        rm, rv, nbt, nbt2 = bns[g]
        stats[g] = lib.BnStats(eps, momentum, rm.data_ptr(), rv.data_ptr(), nbt.data_ptr() if nbt is not None else None,
                               nbt2.data_ptr() if nbt2 is not None else None, mi.data_ptr() + 4 * g * cout)
    xp = lib.ptr_pair(x.data_ptr(), x.data_ptr() + 4 * lsx)
    yp = lib.ptr_pair(y.data_ptr(), y.data_ptr() + 4 * cout)
    pp = lib.ptr_pair(part.data_ptr(), part.data_ptr() + 4 * per)
    lib.conv_bn_fwd2(xp, _pair(wa, wb), yp, pp, stats, 2 * cout, _aligned(ws), nb, d, s)
    return mi, d


def dgrad2(dy, cout, wa, wb, dx, d, skip=None, s=None):
    """dx (N,H,W,2,Cin) contiguous = input gradient of both lanes; dy (N,Ho,Wo,2,cout) contiguous.  skip: twin tensor added in the
    GEMM epilogue (1x1 stride-1 layers) or in a pass of its own."""
    cin = dx.shape[4]
    dd = ConvDesc(d.N, d.H, d.W, cin, d.Ho, d.Wo, cout, d.KH, d.KW, d.stride, d.pad, 2 * cin, 2 * cout)
    key = ('d',) + ops._desc_key(dd)
    nb = _plan.get(key)
    if nb is None:
        nb = _plan[key] = lib.conv_dgrad_workspace_n(dd, 2)
    ws = _ws(nb, dy.device, s, 'twin')
    dyp = lib.ptr_pair(dy.data_ptr(), dy.data_ptr() + 4 * cout)
    dxp = lib.ptr_pair(dx.data_ptr(), dx.data_ptr() + 4 * cin)
    fused = False
    sp, lds = None, 0
    if skip is not None:
        lds, lss = layout(skip)
        fused = d.KH == 1 and d.stride == 1 and cin % 4 == 0 and cout % 4 == 0 and lds % 4 == 0 and lss % 4 == 0 and skip.data_ptr() % 16 == 0
        if fused:
            sp = lib.ptr_pair(skip.data_ptr(), skip.data_ptr() + 4 * lss)
    lib.conv_dgrad2(dyp, _pair(wa, wb), dxp, sp, lds, _aligned(ws) if ws is not None else None, nb, dd, s)
    if skip is not None and not fused:
        rows = d.N * d.H * d.W
        lds, lss = layout(skip)
        if lss == cin:     # the two lanes are adjacent in the skip tensor as well: one pass over 2*Cin columns
            lib.add(dx.data_ptr(), 2 * cin, skip.data_ptr(), lds, dx.data_ptr(), 2 * cin, rows, 2 * cin, s)
        else:
            for g in range(2):
                lib.add(dx.data_ptr() + 4 * g * cin, 2 * cin, skip.data_ptr() + 4 * g * lss, lds, dx.data_ptr() + 4 * g * cin, 2 * cin, rows,
                        cin, s)


def dgrad2_bnred(dy, cout, wa, wb, dx, d, y, mi, gammas, betas, act, skip=None, s=None):
    """dgrad2 with both lanes' BatchNorm backward reductions in the epilogue (include/mmidet_hip.h: mmi_conv_dgrad2_bnred).  y
    (N,H,W,2,Cin) dense: the raw conv output of the layer below, mi its (2, 2*Cin) mean | invstd, gammas / betas its two modules'.
    Returns (partials (2, nparts, 2, Cin), nparts), or None where the form does not exist (stride-2 layers, a skip that cannot be
    added in the epilogue, bf16 storage) -- nothing has been launched then."""
    cin = dx.shape[4]
    if dy.dtype != torch.float32 or d.stride != 1 or not (d.KH == 3 or (d.KH == 1 and ops.BNRED_K1)):
        return None
    dd = ConvDesc(d.N, d.H, d.W, cin, d.Ho, d.Wo, cout, d.KH, d.KW, d.stride, d.pad, 2 * cin, 2 * cout)
    sp, lds = None, 0
    if skip is not None:
        lds, lss = layout(skip)
        if not (d.KH == 1 and cin % 4 == 0 and cout % 4 == 0 and lds % 4 == 0 and lss % 4 == 0 and skip.data_ptr() % 16 == 0):
            return None
        sp = lib.ptr_pair(skip.data_ptr(), skip.data_ptr() + 4 * lss)
    key = ('d',) + ops._desc_key(dd)
    nb = _plan.get(key)
    if nb is None:
        nb = _plan[key] = lib.conv_dgrad_workspace_n(dd, 2)
    kb = ('db',) + ops._desc_key(dd)
    nparts = _plan.get(kb)
    if nparts is None:
        nparts = _plan[kb] = lib.conv_dgrad_row_blocks_n(dd, 2)
    ws = _ws(nb, dy.device, s, 'twin')
    parts = alloc.empty((2, nparts, 2, cin), dtype=torch.float32, device=dy.device)
    hooks = lib.BnReduceHookPair()
    for g in range(2):
        hooks[g] = lib.BnReduceHook(y.data_ptr() + 4 * g * cin, 2 * cin, mi.data_ptr() + 4 * g * cin, 2 * cin, gammas[g].data_ptr(),
                                    betas[g].data_ptr(), act, parts[g].data_ptr())
    dyp = lib.ptr_pair(dy.data_ptr(), dy.data_ptr() + 4 * cout)
    dxp = lib.ptr_pair(dx.data_ptr(), dx.data_ptr() + 4 * cin)
    lib.conv_dgrad2_bnred(dyp, _pair(wa, wb), dxp, sp, lds, hooks, _aligned(ws) if ws is not None else None, nb, dd, s)
    return parts, nparts


def bn_apply_map(y, dout, mi, m, parts, dy, rows, c, act, frozen, s):
    """The apply pass of a twin BatchNorm backward whose reduction came out of dgrad2_bnred (parts = its return value)."""
    p, nparts = parts
    ldd, _ = layout(dout)
    lib.bn_act_bwd_apply_map(y.data_ptr(), c, dout.data_ptr(), ldd, None, 0, mi.data_ptr(), m, lib.ptr_pair(p[0].data_ptr(), p[1].data_ptr()),
                             nparts, dy.data_ptr(), c, rows, c, act, frozen, s)


def wgrad2(dy, dy_off, lddy, lsdy, x, ldx, lsx, wa, wb, cout, k, stride, shape, overlap, out=None):
    """(dwa, dwb) = dy_g^T x_g for both lanes in one launch, on the wgrad side stream when overlap (ops._wgrad's conventions:
    deferred join keeps the operands alive).  out = (dwa, dwb): caller-owned outputs of cout rows each (packed parameters)."""
    n, h, w, cin = shape
    d = ops._desc((n, h, w, cin), cout, k, stride, ldx, lddy)
    dwa, dwb = out if out is not None else (grad_like(wa), grad_like(wb))
    key = ('w',) + ops._desc_key(d)
    nb = _plan.get(key)
    if nb is None:
        nb = _plan[key] = lib.conv_wgrad_workspace_n(d, 2)
    tab = ops.wgrad_table(d, wa.device)
    tabp = tab.data_ptr() if tab is not None else None
    dyp = lib.ptr_pair(dy.data_ptr() + 4 * dy_off, dy.data_ptr() + 4 * (dy_off + lsdy))
    xp = lib.ptr_pair(x.data_ptr(), x.data_ptr() + 4 * lsx)
    dwp = _pair(dwa, dwb)

    def launch(ws, st):
        lib.conv_wgrad2(dyp, xp, dwp, None, _aligned(ws) if ws is not None else None, nb, tabp, d, st)
    if overlap:
        main, side = torch.cuda.current_stream(), ops._side_stream(wa.device)
        ws = _ws(nb, wa.device, side.cuda_stream, 'twinw')
        side.wait_stream(main)
        launch(ws, side.cuda_stream)
        if ops.DEFER_JOIN:
            ops._pending.append((dy, x))
            ops._pending_sides[side.cuda_stream] = side
    else:
        ws = _ws(nb, wa.device, _stream(), 'twinw')
        launch(ws, _stream())
    return dwa, dwb


def _bn_map(gammas, betas, blk, period, split, ls0, ls1, dgs=None, dbs=None):
    m = lib.BnMap()
    m.nblk, m.blk, m.period, m.split, m.ls0, m.ls1 = len(gammas), blk, period, split, ls0, ls1
    for i, (g, b) in enumerate(zip(gammas, betas)):
        m.gamma[i], m.beta[i] = g.data_ptr(), b.data_ptr()
        if dgs is not None:
            m.dgamma[i], m.dbeta[i] = dgs[i].data_ptr(), dbs[i].data_ptr()
    return m


class _TwinConvBnAct(Function):
    """act(BN(conv(x))) [+ residual] of BOTH lanes: x (N,H,W,2,Cin) -> (N,Ho,Wo,2,Cout).  One twin GEMM (statistics of both
    BatchNorm modules finished inside it), one normalise/activate pass over the 2*Cout channels.  skip / dest as ops._ConvBnAct
    (dest = ops.Dest holding a twin view (N,Ho,Wo,2,Cout) of a wider buffer, written in place)."""

    @staticmethod
    def forward(ctx, x, wa, wb, ga, ba, gb, bb, rma, rva, nbta, rmb, rvb, nbtb, residual, stride, act, training, eps, momentum, skip,
                dest):
        x_in = x
        ldx, lsx = layout(x)
        wa, wb = ops._ohwi(wa), ops._ohwi(wb)
        cout, k = wa.shape[0], wa.shape[2]
        n, h, w = x.shape[:3]
        s = _stream()
        pad = k // 2
        ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        y = alloc.empty((n, ho, wo, 2, cout), dtype=x.dtype, device=x.device)
        mi, d = conv_bn_fwd2(x, ldx, lsx, wa, wb, y, cout, k, stride, ((rma, rva, nbta, None), (rmb, rvb, nbtb, None)), eps, momentum,
                             training, s)
        out = dest.t if dest is not None else alloc.empty_like(y)
        assert tuple(out.shape) == tuple(y.shape), 'the destination view %s does not fit the output %s' % (tuple(out.shape), tuple(y.shape))
        ldo, lso = layout(out)
        rows = n * ho * wo
        ldr = 0
        if residual is not None:
            residual = dense(residual)
            ldr = 2 * cout
            assert tuple(residual.shape) == tuple(y.shape)
        m = _bn_map((ga, gb), (ba, bb), cout, cout, cout, lso, 0)
        lib.bn_act_fwd_map(y.data_ptr(), 2 * cout, mi.data_ptr(), m, residual.data_ptr() if residual is not None else None, ldr,
                           out.data_ptr(), ldo, None, 0, rows, 2 * cout, act, s)
        ctx.save_for_backward(x, wa, wb, y, mi, ga, ba, gb, bb)
        ctx.cfg = (d, act, training, residual is not None, skip, k, stride)
        ctx.src_in = ops.bn_src_of(x_in, True) if ctx.needs_input_grad[0] else None
        ctx.src_out = None
        if ops.BNRED and dest is None and y.dtype == torch.float32 and any(ctx.needs_input_grad):
            ctx.src_out = out._bnsrc = ops.BnSrc(y, mi, (ga, gb), (ba, bb), act, True)
        return (out, x_in) if skip else out

    @staticmethod
    def backward(ctx, dout, dskip=None):
        x, wa, wb, y, mi, ga, ba, gb, bb = ctx.saved_tensors
        d, act, training, has_res, skip, k, stride = ctx.cfg
        cout = wa.shape[0]
        n, h, w = x.shape[:3]
        cin = x.shape[4]
        rows = d.N * d.Ho * d.Wo
        s = _stream()
        ldd, lsd = layout(dout)
        ldx, lsx = layout(x)
        dy = alloc.empty_like(y)
        dgs = (grad_like(ga), grad_like(gb))
        dbs = (grad_like(ba), grad_like(bb))
        m = _bn_map((ga, gb), (ba, bb), cout, cout, cout, lsd, 0, dgs, dbs)
        parts = ctx.src_out.take(dout) if ctx.src_out is not None and ldd == 2 * cout and lsd == cout else None
        if parts is not None:        # both lanes' reductions came out of the consumer's dgrad epilogue: fold + apply
            bn_apply_map(y, dout, mi, m, parts, dy, rows, 2 * cout, act, 0 if training else 1, s)
        else:
            nbw = ops.bn_bwd_ws(rows, 2 * cout)
            ws = zeroed_scratch(nbw, y.device, s, tag='bn')
            lib.bn_act_bwd_map(y.data_ptr(), 2 * cout, dout.data_ptr(), ldd, None, 0, mi.data_ptr(), m, ws.data_ptr(), nbw, dy.data_ptr(),
                               2 * cout, rows, 2 * cout, act, 0 if training else 1, s)
        need_x = ctx.needs_input_grad[0]
        both = ops.OVERLAP_WGRAD and need_x
        dwa = dwb = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dwa, dwb = wgrad2(dy, 0, 2 * cout, cout, x, ldx, lsx, wa, wb, cout, k, stride, (n, h, w, cin), both)
        dx = None
        if need_x:
            dx = alloc.empty((n, h, w, 2, cin), dtype=x.dtype, device=x.device)
            src, done = ctx.src_in, None
            if src is not None and src.uses == 1 and lsx == cin and ldx == 2 * cin and cin % 4 == 0 and cout % 4 == 0:
                done = dgrad2_bnred(dy, cout, wa, wb, dx, d, src.y, src.mi, src.gammas, src.betas, src.act, dskip if skip else None, s)
                if done is not None:
                    src.wrote(dx, done)
            if done is None:
                dgrad2(dy, cout, wa, wb, dx, d, dskip if skip else None, s)
        if both:
            ops._join_side(x.device)
        return (dx, dwa, dwb, dgs[0], dbs[0], dgs[1], dbs[1], None, None, None, None, None, None, (dout if has_res else None), None,
                None, None, None, None, None, None)


def conv_bn_act2(x, ca, cb, residual=None, skip=False, dest=None):
    """ca / cb: the two lanes' models.common.Conv modules."""
    a, b = ca.bn, cb.bn
    return _TwinConvBnAct.apply(x, ca.conv.weight, cb.conv.weight, a.weight, a.bias, b.weight, b.bias, a.running_mean, a.running_var,
                                a.num_batches_tracked, b.running_mean, b.running_var, b.num_batches_tracked, residual,
                                ca.conv.stride[0], ca._act_id(), a.training, a.eps, a.momentum, skip, dest)


class _TwinDualConvBnAct(Function):
    """C3's cv1 | cv2 (ops._DualConvBnAct) for both lanes: ONE twin GEMM with 2*c_ columns per lane over packed weights, one
    statistics fold, one normalise pass over the 4*c_ channels whose cv1 halves go to the bottleneck chain (N,H,W,2,c_) and whose
    cv2 halves go straight into the lanes' concat buffer (N,H,W,2,2c_) that cv3 reads."""

    @staticmethod
    def forward(ctx, x, w1a, w2a, w1b, w2b, g1a, b1a, g2a, b2a, g1b, b1b, g2b, b2b, rma, rva, nbt1a, nbt2a, rmb, rvb, nbt1b, nbt2b, act,
                training, eps, momentum, cat):
        ldx, lsx = layout(x)
        w1a, w2a, w1b, w2b = (ops._ohwi(t) for t in (w1a, w2a, w1b, w2b))
        c_ = w1a.shape[0]
        n, h, w = x.shape[:3]
        s = _stream()
        y = alloc.empty((n, h, w, 2, 2 * c_), dtype=x.dtype, device=x.device)
        mi, d = conv_bn_fwd2(x, ldx, lsx, w1a, w1b, y, 2 * c_, 1, 1, ((rma, rva, nbt1a, nbt2a), (rmb, rvb, nbt1b, nbt2b)), eps, momentum,
                             training, s)
        a = alloc.empty((n, h, w, 2, c_), dtype=x.dtype, device=x.device)
        b = cat.t[..., c_:]
        ldb, lsb = layout(b)
        rows = n * h * w
        m = _bn_map((g1a, g2a, g1b, g2b), (b1a, b2a, b1b, b2b), c_, 2 * c_, c_, c_, lsb)
        lib.bn_act_fwd_map(y.data_ptr(), 4 * c_, mi.data_ptr(), m, None, 0, a.data_ptr(), 2 * c_, b.data_ptr(), ldb, rows, 4 * c_, act, s)
        ctx.save_for_backward(x, w1a, w2a, w1b, w2b, y, mi, g1a, b1a, g2a, b2a, g1b, b1b, g2b, b2b)
        ctx.cfg = (d, act, training, c_)
        return a, b

    @staticmethod
    def backward(ctx, da, db):
        x, w1a, w2a, w1b, w2b, y, mi, g1a, b1a, g2a, b2a, g1b, b1b, g2b, b2b = ctx.saved_tensors
        d, act, training, c_ = ctx.cfg
        n, h, w = x.shape[:3]
        cin = x.shape[4]
        rows, s = n * h * w, _stream()
        ldx, lsx = layout(x)
        lda, lsa = layout(da)
        ldb, lsb = layout(db)
        dy = alloc.empty_like(y)
        gs, bs = (g1a, g2a, g1b, g2b), (b1a, b2a, b1b, b2b)
        dgs, dbs = tuple(grad_like(t) for t in gs), tuple(grad_like(t) for t in bs)
        m = _bn_map(gs, bs, c_, 2 * c_, c_, lsa, lsb, dgs, dbs)
        nbw = ops.bn_bwd_ws(rows, 4 * c_)
        ws = zeroed_scratch(nbw, y.device, s, tag='bn')
        lib.bn_act_bwd_map(y.data_ptr(), 4 * c_, da.data_ptr(), lda, db.data_ptr(), ldb, mi.data_ptr(), m, ws.data_ptr(), nbw,
                           dy.data_ptr(), 4 * c_, rows, 4 * c_, act, 0 if training else 1, s)
        need_x = ctx.needs_input_grad[0]
        both = ops.OVERLAP_WGRAD and need_x
        if ops.GRAD_SLOTS or not all(ctx.needs_input_grad[1:5]):
            # the four weight gradients: cv1 of both lanes, cv2 of both lanes (each lands in its parameter's bucket view)
            dw1a, dw1b = wgrad2(dy, 0, 4 * c_, 2 * c_, x, ldx, lsx, w1a, w1b, c_, 1, 1, (n, h, w, cin), both)
            dw2a, dw2b = wgrad2(dy, c_, 4 * c_, 2 * c_, x, ldx, lsx, w2a, w2b, c_, 1, 1, (n, h, w, cin), both)
        else:
            # cv1 | cv2 are one (2c_, Cin) matrix per lane (ops.pack_pair), and so is their gradient: ONE twin launch with 2c_
            # output rows; each parameter's .grad is a row block of the lane's buffer
            fa, fb = (alloc.empty(2 * c_ * cin, dtype=w1a.dtype, device=x.device) for _ in range(2))
            wgrad2(dy, 0, 4 * c_, 2 * c_, x, ldx, lsx, w1a, w1b, 2 * c_, 1, 1, (n, h, w, cin), both, out=(fa, fb))
            dw1a, dw2a = (fa[i * c_ * cin:(i + 1) * c_ * cin].as_strided(w1a.shape, w1a.stride()) for i in range(2))
            dw1b, dw2b = (fb[i * c_ * cin:(i + 1) * c_ * cin].as_strided(w1b.shape, w1b.stride()) for i in range(2))
        dx = None
        if need_x:
            dx = alloc.empty((n, h, w, 2, cin), dtype=x.dtype, device=x.device)
            dgrad2(dy, 2 * c_, w1a, w1b, dx, d, None, s)
        if both:
            ops._join_side(x.device)
        return (dx, dw1a, dw2a, dw1b, dw2b, dgs[0], dbs[0], dgs[1], dbs[1], dgs[2], dbs[2], dgs[3], dbs[3]) + (None,) * 13


def dual_conv_bn_act2(x, c3a, c3b, cat):
    """c3a / c3b: the two lanes' C3 modules (cv1 / cv2 packed by ops.pack_pair); cat: ops.Dest holding their concat buffer
    (N,H,W,2,2c_)."""
    a1, a2, b1, b2 = c3a.cv1, c3a.cv2, c3b.cv1, c3b.cv2
    return _TwinDualConvBnAct.apply(
        x, a1.conv.weight, a2.conv.weight, b1.conv.weight, b2.conv.weight,
        a1.bn.weight, a1.bn.bias, a2.bn.weight, a2.bn.bias, b1.bn.weight, b1.bn.bias, b2.bn.weight, b2.bn.bias,
        a1.bn.running_mean, a1.bn.running_var, a1.bn.num_batches_tracked, a2.bn.num_batches_tracked,
        b1.bn.running_mean, b1.bn.running_var, b1.bn.num_batches_tracked, b2.bn.num_batches_tracked,
        a1._act_id(), a1.bn.training, a1.bn.eps, a1.bn.momentum, cat)


class _TwinSppPool(Function):
    """x (N,H,W,2,c) -> (N,H,W,2,4c): per lane cat(x, mp5, mp9, mp13) (ops._SppPool), the lanes' outputs side by side."""

    @staticmethod
    def forward(ctx, x):
        x = dense(x)
        n, h, w, _, c = x.shape
        out = alloc.empty((n, h, w, 2, 4 * c), dtype=x.dtype, device=x.device)
        s = _stream()
        for g in range(2):
            lib.spp_pool_fwd(x.data_ptr() + 4 * g * c, 2 * c, out.data_ptr() + 16 * g * c, 8 * c, n, h, w, c, s)
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        n, h, w, _, c = x.shape
        ldg, lsg = layout(g)
        dx = alloc.empty_like(x)
        s = _stream()
        for q in range(2):
            lib.spp_pool_bwd(x.data_ptr() + 4 * q * c, 2 * c, g.data_ptr() + 4 * q * lsg, ldg, dx.data_ptr() + 4 * q * c, 2 * c, n, h, w, c, s)
        return dx


def spp_pool2(x):
    return _TwinSppPool.apply(x)


class _TwinSpaceToDepth(Function):
    """Focus' slicing (models/common.py:707-709) of both images into one twin tensor: (N,H,W,C) x 2 -> (N,H/2,W/2,2,4C).  The IR
    image carries no gradient; the RGB one does (the CEM has parameters)."""

    @staticmethod
    def forward(ctx, xa, xb):
        xa, xb = xa.contiguous(), xb.contiguous()
        n, h, w, c = xa.shape
        y = alloc.empty((n, h // 2, w // 2, 2, 4 * c), dtype=xa.dtype, device=xa.device)
        s = _stream()
        for g, t in enumerate((xa, xb)):
            lib.space_to_depth_ld(t.data_ptr(), y.data_ptr() + 16 * g * c, n, h, w, c, 8 * c, 0, s)
        ctx.shape = (n, h, w, c)
        return y

    @staticmethod
    def backward(ctx, g):
        n, h, w, c = ctx.shape
        ldg, lsg = layout(g)
        s = _stream()
        outs = []
        for q in range(2):
            if not ctx.needs_input_grad[q]:
                outs.append(None)
                continue
            dx = alloc.empty((n, h, w, c), dtype=g.dtype, device=g.device)
            lib.space_to_depth_ld(g.data_ptr() + 4 * q * lsg, dx.data_ptr(), n, h, w, c, ldg, 1, s)
            outs.append(dx)
        return tuple(outs)


def space_to_depth2(xa, xb):
    return _TwinSpaceToDepth.apply(xa, xb)


class _TwinPoolTokens(Function):
    """fusion_ops._PoolTokens on a twin tensor: AdaptiveAvgPool2d(8,8) of both lanes -> (B,128,C) tokens (rgb first).  skip=True
    also returns the twin tensor itself for its other consumer (the Add2 pair), whose gradient then arrives here and is added
    inside the pool-gradient kernels."""

    @staticmethod
    def forward(ctx, x, skip):
        ctx.set_materialize_grads(False)
        ld, ls = layout(x)
        n, h, w, _, c = x.shape
        tok = alloc.empty((n, 128, c), dtype=x.dtype, device=x.device)
        s = _stream()
        for g in range(2):
            lib.avgpool8_fwd(x.data_ptr() + 4 * g * ls, ld, n, h, w, c, tok.data_ptr() + 4 * g * 64 * c, 128 * c, c, s)
        ctx.shape = (n, h, w, c)
        return (tok, x) if skip else tok

    @staticmethod
    def backward(ctx, g, gx=None):
        n, h, w, c = ctx.shape
        if g is None:
            return gx, None
        g = g.contiguous()
        s = _stream()
        d = alloc.empty((n, h, w, 2, c), dtype=g.dtype, device=g.device)
        lds = lss = 0
        if gx is not None:
            lds, lss = layout(gx)
            if lds % 4 != 0 or lss % 4 != 0 or gx.data_ptr() % 16 != 0:
                gx = gx.contiguous()
                lds, lss = 2 * c, c
        for q in range(2):
            lib.avgpool8_bwd_acc(g.data_ptr() + 4 * q * 64 * c, 128 * c, c, (gx.data_ptr() + 4 * q * lss) if gx is not None else None, lds,
                                 d.data_ptr() + 4 * q * c, 2 * c, n, h, w, c, s)
        return d, None


def pool_tokens2(x, skip=False):
    return _TwinPoolTokens.apply(x, skip)


class _TwinUpsampleAdd(Function):
    """The Add2 pair (models/common.py:924-935 of the reference, index 0 and 1) on a twin tensor: out lane g = x lane g +
    bilinear(tok_g 8x8 -> HxW) (fusion_ops._UpsampleAdd per lane, written side by side)."""

    @staticmethod
    def forward(ctx, x, tok_a, tok_b):
        ld, ls = layout(x)
        n, h, w, _, c = x.shape
        out = alloc.empty((n, h, w, 2, c), dtype=x.dtype, device=x.device)
        s = _stream()
        for g, tok in enumerate((tok_a.contiguous(), tok_b.contiguous())):
            lib.upsample_add_fwd(x.data_ptr() + 4 * g * ls, ld, tok.data_ptr(), 64 * c, c, out.data_ptr() + 4 * g * c, 2 * c, n, h, w, c, s)
        ctx.shape = (n, h, w, c)
        return out

    @staticmethod
    def backward(ctx, g):
        n, h, w, c = ctx.shape
        ldg, lsg = layout(g)
        s = _stream()
        dts = []
        for q in range(2):
            dt = alloc.empty((n, 8, 8, c), dtype=g.dtype, device=g.device)
            lib.upsample_add_bwd(g.data_ptr() + 4 * q * lsg, ldg, dt.data_ptr(), 64 * c, c, n, h, w, c, s)
            dts.append(dt)
        return g, dts[0], dts[1]


def upsample_add2(x, tok_a, tok_b):
    return _TwinUpsampleAdd.apply(x, tok_a, tok_b)


class _TwinAddLanes(Function):
    """The neck's `Add` of the two streams (models/common.py:914-921): lane 0 + lane 1 of a twin tensor -> (N,H,W,C).  The
    gradient of both lanes is the incoming gradient itself: returned as a stride-0 twin view, no copy."""

    @staticmethod
    def forward(ctx, x, dest):
        ld, ls = layout(x)
        n, h, w, _, c = x.shape
        out = ops._dest_view(dest, (n, h, w, c)) if dest is not None else alloc.empty((n, h, w, c), dtype=x.dtype, device=x.device)
        lib.add(x.data_ptr(), ld, x.data_ptr() + 4 * ls, ld, out.data_ptr(), ops.rows_of(out)[1], n * h * w, c, _stream())
        return out

    @staticmethod
    def backward(ctx, g):
        g, ldg = ops.rows_of(g)
        n, h, w, c = g.shape
        return g.as_strided((n, h, w, 2, c), (h * w * ldg, w * ldg, ldg, 0, 1)), None      # both lanes: the same rows, no copy


def add_lanes(x, dest=None):
    return _TwinAddLanes.apply(x, dest)


class _TwinStack(Function):
    """Two single-lane maps (N,H,W,C) -> one twin tensor (a copy): the entry of a twin section whose inputs were produced by
    single-lane layers."""

    @staticmethod
    def forward(ctx, a, b):
        n, h, w, c = a.shape
        out = alloc.empty((n, h, w, 2, c), dtype=a.dtype, device=a.device)
        s = _stream()
        for g, t in enumerate((a, b)):
            t, ld = ops.rows_of(t)
            lib.copy2d(t.data_ptr(), ld, out.data_ptr() + 4 * g * c, 2 * c, n * h * w, c, s)
        return out

    @staticmethod
    def backward(ctx, g):
        return g[..., 0, :], g[..., 1, :]


def stack(a, b):
    return _TwinStack.apply(a, b)


class _TwinLanes(Function):
    """A twin tensor -> its two lanes as (N,H,W,C) views (no copy): the exit of a twin section towards single-lane consumers.  The
    two incoming gradients are copied side by side."""

    @staticmethod
    def forward(ctx, x):
        return x[..., 0, :], x[..., 1, :]

    @staticmethod
    def backward(ctx, ga, gb):
        n, h, w, c = ga.shape
        out = alloc.empty((n, h, w, 2, c), dtype=ga.dtype, device=ga.device)
        s = _stream()
        for q, t in enumerate((ga, gb)):
            t, ld = ops.rows_of(t)
            lib.copy2d(t.data_ptr(), ld, out.data_ptr() + 4 * q * c, 2 * c, n * h * w, c, s)
        return out


def lanes(x):
    return _TwinLanes.apply(x)
