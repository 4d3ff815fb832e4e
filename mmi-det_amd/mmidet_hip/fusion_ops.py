"""autograd glue for the fusion stack (GPT / GPT1_fourier, CBM / IGM statistics) over the C ABI.  Token tensors are
(B,128,C) or (rows,C) contiguous fp32; spatial tensors NHWC."""
import torch
from torch.autograd import Function

from . import alloc, lib, ops
from .lib import (EPI_DROPOUT_RESIDUAL, EPI_GELU, EPI_GELU_GRAD, EPI_NONE, ConvDesc, LinearEpilogue)
from .ops import _nrows, _stream, grad_like, rows_of, scratch, zeroed_scratch

_drop_counter = [0]
_seed_state = {}


def next_seed():
    """Per-call-site salt of a dropout mask: reproducible from torch.manual_seed, distinct per call (and per rank when
    ranks seed differently, as train.py:init_seeds(2 + rank) does).  The per-step randomness comes from the device seed
    word (seed_state), so the salts may be constants baked into a captured graph."""
    _drop_counter[0] += 1
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + _drop_counter[0] * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF


def seed_state(device):
    """The device-resident 64-bit seed word every dropout kernel adds to its salt."""
    t = _seed_state.get(device)
    if t is None:
        t = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device=device)
        _seed_state[device] = t
    return t


def advance_seed(device):
    """Call once per training step (TrainStep does): new masks for every dropout site, also under graph replay."""
    lib.seed_advance(seed_state(device).data_ptr(), _stream())
    _drop_counter[0] = 0


class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = x.contiguous()
        c = x.shape[-1]
        rows = _nrows(x)
        y = alloc.empty_like(x)
        stats = alloc.empty((rows, 2), dtype=x.dtype, device=x.device)
        lib.layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), stats.data_ptr(), rows, c, eps,
                          _stream())
        ctx.save_for_backward(x, gamma, stats)
        ctx.beta = beta
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, stats = ctx.saved_tensors
        dy = dy.contiguous()
        c = x.shape[-1]
        rows = _nrows(x)
        dx = alloc.empty_like(x)
        dg, db = grad_like(gamma), grad_like(ctx.beta)
        part = scratch(lib.layernorm_bwd_parts(rows) * 2 * c, x.device)
        lib.layernorm_bwd(x.data_ptr(), gamma.data_ptr(), stats.data_ptr(), dy.data_ptr(), dx.data_ptr(), part.data_ptr(),
                          dg.data_ptr(), db.data_ptr(), rows, c, _stream())
        return dx, dg, db, None


def layernorm(x, gamma, beta, eps=1e-5):
    return _LayerNorm.apply(x, gamma, beta, eps)


class _Gelu(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = alloc.empty_like(x)
        lib.gelu_fwd(x.data_ptr(), y.data_ptr(), x.numel(), _stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        dy = dy.contiguous()
        dx = alloc.empty_like(x)
        lib.gelu_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), _stream())
        return dx


def gelu(x):
    return _Gelu.apply(x)


class _Sigmoid(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = alloc.empty_like(x)
        lib.sigmoid_fwd(x.data_ptr(), y.data_ptr(), x.numel(), _stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, = ctx.saved_tensors
        dy = dy.contiguous()
        dx = alloc.empty_like(y)
        lib.sigmoid_bwd(y.data_ptr(), dy.data_ptr(), dx.data_ptr(), y.numel(), _stream())
        return dx


def sigmoid(x):
    return _Sigmoid.apply(x)


class _Mul(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        assert a.shape == b.shape
        o = alloc.empty_like(a)
        lib.mul(a.data_ptr(), b.data_ptr(), o.data_ptr(), a.numel(), _stream())
        ctx.save_for_backward(a, b)
        return o

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        da, db = alloc.empty_like(a), alloc.empty_like(b)
        lib.mul(g.data_ptr(), b.data_ptr(), da.data_ptr(), a.numel(), _stream())
        lib.mul(g.data_ptr(), a.data_ptr(), db.data_ptr(), a.numel(), _stream())
        return da, db


def mul(a, b):
    return _Mul.apply(a, b)


class _DropoutAdd(Function):
    """(a [+ b]) * mask/(1-p); b (if given) is broadcast over the batch (positional embedding)."""

    @staticmethod
    def forward(ctx, a, b, p, seed):
        a = a.contiguous()
        out = alloc.empty_like(a)
        bmod = 0
        if b is not None:
            b = b.contiguous()
            bmod = b.numel()
            assert a.numel() % bmod == 0
        lib.dropout(a.data_ptr(), b.data_ptr() if b is not None else None, bmod, out.data_ptr(), a.numel(), p, seed,
                    seed_state(a.device).data_ptr() if p > 0 else None, _stream())
        ctx.cfg = (p, seed, bmod, tuple(b.shape) if b is not None else None)
        return out

    @staticmethod
    def backward(ctx, g):
        p, seed, bmod, bshape = ctx.cfg
        g = g.contiguous()
        if p > 0:
            da = alloc.empty_like(g)
            lib.dropout(g.data_ptr(), None, 0, da.data_ptr(), g.numel(), p, seed, seed_state(g.device).data_ptr(),
                        _stream())
        else:
            da = g
        db = None
        if bshape is not None and ctx.needs_input_grad[1]:
            rows = g.numel() // bmod
            db = alloc.empty(bshape, dtype=g.dtype, device=g.device)
            part = scratch(lib.bn_bwd_parts(rows) * bmod, g.device)
            lib.colsum(da.data_ptr(), bmod, rows, bmod, part.data_ptr(), db.data_ptr(), _stream())
        return da, db, None, None


def dropout_add(a, b=None, p=0.0, training=True):
    p = float(p) if training else 0.0
    if p == 0.0 and b is None:
        return a
    return _DropoutAdd.apply(a, b, p, next_seed() if p > 0 else 0)


class _Attention(Function):
    @staticmethod
    def forward(ctx, q, k, v, heads, p, seed):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        b, t, c = q.shape
        assert t == 128, 'the fusion transformers always see 2*8*8 = 128 tokens'
        dk = c // heads
        out = alloc.empty_like(q)
        probs = alloc.empty((b, heads, t, t), dtype=q.dtype, device=q.device)
        lib.attention_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), probs.data_ptr(), b, heads, dk, c, p,
                          seed, seed_state(q.device).data_ptr() if p > 0 else None, _stream())
        ctx.save_for_backward(q, k, v, probs)
        ctx.cfg = (heads, p, seed)
        return out

    @staticmethod
    def backward(ctx, do):
        q, k, v, probs = ctx.saved_tensors
        heads, p, seed = ctx.cfg
        do = do.contiguous()
        b, t, c = q.shape
        dq, dk_, dv = alloc.empty_like(q), alloc.empty_like(q), alloc.empty_like(q)
        lib.attention_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), probs.data_ptr(), do.data_ptr(), dq.data_ptr(),
                          dk_.data_ptr(), dv.data_ptr(), b, heads, c // heads, c, p, seed,
                          seed_state(q.device).data_ptr() if p > 0 else None, _stream())
        return dq, dk_, dv, None, None, None


def attention(q, k, v, heads, p=0.0, training=True):
    p = float(p) if training else 0.0
    return _Attention.apply(q, k, v, heads, p, next_seed() if p > 0 else 0)


def _linear_fwd(x, w, bias, y, rows, cin, cout, s, kind=EPI_NONE, aux=None, aux_out=None, p=0.0, seed=0, seed_dev=None):
    """y = epilogue(x w^T + bias) over contiguous (rows, cin) -> (rows, cout): mmi_linear_fwd_fused."""
    d = ConvDesc(rows, 1, 1, cin, 1, 1, cout, 1, 1, 1, 0, cin, cout)
    nb = lib.conv_fwd_workspace(d)
    ws = zeroed_scratch(nb, x.device, s) if nb else None
    e = LinearEpilogue(kind, cout, cout, p, aux.data_ptr() if aux is not None else None,
                       aux_out.data_ptr() if aux_out is not None else None, seed, seed_dev)
    lib.linear_fwd_fused(x.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
                         ws.data_ptr() if nb else None, nb, d, e, s)


def _linear_dgrad(dy, w, dx, rows, cin, cout, s, kind=EPI_NONE, aux=None):
    """dx = epilogue(dy w) over contiguous (rows, cout) -> (rows, cin): mmi_linear_dgrad_fused."""
    d = ConvDesc(rows, 1, 1, cin, 1, 1, cout, 1, 1, 1, 0, cin, cout)
    nb = lib.conv_dgrad_workspace(d)
    ws = zeroed_scratch(nb, dy.device, s) if nb else None
    e = LinearEpilogue(kind, cin, cin, 0.0, aux.data_ptr() if aux is not None else None, None, 0, None)
    lib.linear_dgrad_fused(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), ws.data_ptr() if nb else None, nb, d, e, s)


def _linear_wgrad(dy, x, w, bias, rows, cin, cout, need_w, need_b, lddy=None):
    """(dw, dbias) of y = x w^T + bias on the weight-gradient side stream (ops._wgrad); lddy: row stride of dy when it
    is a column block of a wider buffer."""
    lddy = cout if lddy is None else lddy
    if not need_w:
        db = None
        if need_b:
            db = grad_like(bias)
            part = scratch(lib.bn_bwd_parts(rows) * cout, dy.device)
            lib.colsum(dy.data_ptr(), lddy, rows, cout, part.data_ptr(), db.data_ptr(), _stream())
        return None, db
    d = ConvDesc(rows, 1, 1, cin, 1, 1, cout, 1, 1, 1, 0, cin, lddy)
    if bias is not None and need_b:
        return ops._wgrad(dy, lddy, x, cin, w, d, overlap=ops.OVERLAP_WGRAD, want_bias=True, bias=bias)
    return ops._wgrad(dy, lddy, x, cin, w, d, overlap=ops.OVERLAP_WGRAD), None


def _back_to_back(a, b, c):
    """Three equally shaped contiguous tensors that lie one after the other in memory (pack_qkv)."""
    n = a.numel() * a.element_size()
    return (a.shape == b.shape == c.shape and a.is_contiguous() and b.is_contiguous() and c.is_contiguous()
            and b.data_ptr() == a.data_ptr() + n and c.data_ptr() == b.data_ptr() + n)


def pack_qkv(model):
    """Re-seat que_proj / key_proj / val_proj of every SelfAttention (models/common.py:1167-1169 of the reference) on one
    (3d, d) weight and one (3d,) bias buffer: same Parameters, same state_dict keys, same values, but the three projections
    then need no gathering copy per call (_TransformerBlock checks the addresses on every call; a model whose parameters were
    moved afterwards, e.g. by .to(), runs the same kernels on a gathered copy of the three matrices).
    Call before anything caches parameter addresses (optimizer pointer tables, gradient buckets).  Returns the count."""
    n = 0
    for m in model.modules():
        if not all(hasattr(m, a) for a in ('que_proj', 'key_proj', 'val_proj')):
            continue
        lin = (m.que_proj, m.key_proj, m.val_proj)
        if lin[0].bias is None or any(l.weight.shape != lin[0].weight.shape or l.weight.grad is not None for l in lin):
            continue
        for name in ('weight', 'bias'):
            ps = [getattr(l, name) for l in lin]
            if _back_to_back(*ps):
                continue
            flat = torch.cat([p.data.reshape(-1) for p in ps])
            k = ps[0].numel()
            for i, p in enumerate(ps):
                p.data = flat[i * k:(i + 1) * k].view(ps[0].shape)
        n += 1
    return n


def _ln_param_grads(x, stats, dy, gamma, beta, rows, c):
    """dgamma, dbeta: off the dependency chain, so they go where the weight gradients go."""
    dg, db = grad_like(gamma), grad_like(beta)
    nparts = lib.layernorm_bwd_parts(rows)
    if ops.OVERLAP_WGRAD:
        main, side = torch.cuda.current_stream(), ops._side_stream(x.device)
        part = scratch(nparts * 2 * c, x.device, slot=5, stream=side.cuda_stream)
        side.wait_stream(main)
        lib.layernorm_bwd_params(x.data_ptr(), stats.data_ptr(), dy.data_ptr(), part.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                 rows, c, side.cuda_stream)
        if ops.DEFER_JOIN:
            ops._pending.append((x, stats, dy))
            ops._pending_sides[side.cuda_stream] = side
    else:
        part = scratch(nparts * 2 * c, x.device)
        lib.layernorm_bwd_params(x.data_ptr(), stats.data_ptr(), dy.data_ptr(), part.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                 rows, c, _stream())
    return dg, db


class _TransformerBlock(Function):
    """myTransformerBlock (models/common.py:1237-1267 of the reference) as one autograd node:
        x1 = x + drop(out_proj(attention(q, k, v)(ln_input(x))));  x2 = x1 + drop(mlp2(GELU(mlp0(ln_output(x1)))))
    Forward 7 dependent kernels (LayerNorm, the q|k|v projection, attention, out_proj+dropout+residual, LayerNorm, mlp0+GELU,
    mlp2+dropout+residual), backward 9 on the chain; weight, bias and LayerNorm-parameter gradients run on the side stream.
    The token chains are launch-latency bound (2048 rows x 128..1024 channels), so the kernel count is their cost."""

    @staticmethod
    def forward(ctx, x, heads, ps, eps, seeds, *params):
        g1, b1, wq, bq, wk, bk, wv, bv, wo, bo, g2, b2, w1, c1, w2, c2 = params
        x = x.contiguous()
        assert x.dim() == 3 and x.shape[1] == 128, 'the fusion transformers always see 2*8*8 = 128 tokens'
        bsz, t, d = x.shape
        rows, hid, dev, s = bsz * t, w1.shape[0], x.device, _stream()
        sd = seed_state(dev).data_ptr() if max(ps) > 0 else None
        new = lambda *shape: alloc.empty(shape, dtype=x.dtype, device=dev)  # noqa: E731
        ln1y, st1 = alloc.empty_like(x), new(rows, 2)
        lib.layernorm_fwd(x.data_ptr(), g1.data_ptr(), b1.data_ptr(), ln1y.data_ptr(), st1.data_ptr(), rows, d, eps[0], s)
        # q, k, v are ONE projection GEMM over the (3d, d) weight pack_qkv() lays out.  Parameters that do not lie back to back
        # (a model whose parameters were moved after packing, or that never was packed) are gathered into such a matrix here:
        # the same kernels, the same numbers, one small copy per block instead of a second code path.
        packed = _back_to_back(wq, wk, wv) and _back_to_back(bq, bk, bv)
        wqkv, bqkv = (wq, bq) if packed else (torch.cat([wq, wk, wv], 0), torch.cat([bq, bk, bv], 0))
        qkv = new(bsz, t, 3 * d)
        _linear_fwd(ln1y, wqkv, bqkv, qkv, rows, d, 3 * d, s)
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
        o, probs = alloc.empty_like(x), new(bsz, heads, t, t)
        lib.attention_fwd_strided(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), probs.data_ptr(), bsz, heads,
                                  d // heads, 3 * d, d, ps[0], seeds[0], sd if ps[0] > 0 else None, s)
        x1 = alloc.empty_like(x)
        _linear_fwd(o, wo, bo, x1, rows, d, d, s, EPI_DROPOUT_RESIDUAL, aux=x, p=ps[1], seed=seeds[1],
                    seed_dev=sd if ps[1] > 0 else None)
        ln2y, st2 = alloc.empty_like(x), new(rows, 2)
        lib.layernorm_fwd(x1.data_ptr(), g2.data_ptr(), b2.data_ptr(), ln2y.data_ptr(), st2.data_ptr(), rows, d, eps[1], s)
        h, g = new(bsz, t, hid), new(bsz, t, hid)
        _linear_fwd(ln2y, w1, c1, g, rows, d, hid, s, EPI_GELU, aux_out=h)
        x2 = alloc.empty_like(x)
        _linear_fwd(g, w2, c2, x2, rows, hid, d, s, EPI_DROPOUT_RESIDUAL, aux=x1, p=ps[2], seed=seeds[2],
                    seed_dev=sd if ps[2] > 0 else None)
        ctx.save_for_backward(x, st1, ln1y, q, k, v, probs, o, x1, st2, ln2y, h, g, *params)
        ctx.cfg = (heads, ps, seeds, packed)
        return x2

    @staticmethod
    def backward(ctx, dx2):
        x, st1, ln1y, q, k, v, probs, o, x1, st2, ln2y, h, g = ctx.saved_tensors[:13]
        g1, b1, wq, bq, wk, bk, wv, bv, wo, bo, g2, b2, w1, c1, w2, c2 = ctx.saved_tensors[13:]
        heads, ps, seeds, packed = ctx.cfg
        need = ctx.needs_input_grad[5:]
        dx2 = dx2.contiguous()
        bsz, t, d = x.shape
        rows, hid, dev, s = bsz * t, w1.shape[0], x.device, _stream()
        sd = seed_state(dev).data_ptr() if max(ps) > 0 else None
        grads = [None] * 16

        def wgrad(dy, inp, w, b, iw, cin, cout, lddy=None):
            grads[iw], grads[iw + 1] = _linear_wgrad(dy, inp, w, b, rows, cin, cout, need[iw], need[iw + 1], lddy)

        # ---- x2 = x1 + drop(mlp2(GELU(mlp0(ln_output(x1))))) ----
        if ps[2] > 0:
            dy2 = alloc.empty_like(dx2)
            lib.dropout(dx2.data_ptr(), None, 0, dy2.data_ptr(), dx2.numel(), ps[2], seeds[2], sd, s)
        else:
            dy2 = dx2
        wgrad(dy2, g, w2, c2, 14, hid, d)
        dh = alloc.empty_like(h)
        _linear_dgrad(dy2, w2, dh, rows, hid, d, s, EPI_GELU_GRAD, aux=h)
        wgrad(dh, ln2y, w1, c1, 12, d, hid)
        dln2 = alloc.empty_like(x)
        _linear_dgrad(dh, w1, dln2, rows, d, hid, s)
        if need[10] or need[11]:
            grads[10], grads[11] = _ln_param_grads(x1, st2, dln2, g2, b2, rows, d)
        # dx1 = dx2 + LayerNorm'(dln2); dy1 = dx1 through the out_proj dropout mask
        dx1 = alloc.empty_like(x)
        dy1 = alloc.empty_like(x) if ps[1] > 0 else None
        lib.layernorm_bwd_input(x1.data_ptr(), g2.data_ptr(), st2.data_ptr(), dln2.data_ptr(), dx2.data_ptr(), dx1.data_ptr(),
                                dy1.data_ptr() if dy1 is not None else None, ps[1], seeds[1], sd if ps[1] > 0 else None,
                                rows, d, s)
        if dy1 is None:
            dy1 = dx1
        # ---- x1 = x + drop(out_proj(attention(ln_input(x)))) ----
        wgrad(dy1, o, wo, bo, 8, d, d)
        do = alloc.empty_like(x)
        _linear_dgrad(dy1, wo, do, rows, d, d, s)
        ldq = 3 * d
        dqkv = alloc.empty((bsz, t, 3 * d), dtype=x.dtype, device=dev)
        dq, dk_, dv = dqkv[..., :d], dqkv[..., d:2 * d], dqkv[..., 2 * d:]
        lib.attention_bwd_strided(q.data_ptr(), k.data_ptr(), v.data_ptr(), probs.data_ptr(), do.data_ptr(), dq.data_ptr(),
                                  dk_.data_ptr(), dv.data_ptr(), bsz, heads, d // heads, ldq, d, ps[0], seeds[0],
                                  sd if ps[0] > 0 else None, s)
        dln1 = alloc.empty_like(x)
        wqkv = wq if packed else torch.cat([wq, wk, wv], 0)      # (see forward)
        if all(need[2:8]) and not ops.GRAD_SLOTS:
            # the three projections' weight (and bias) gradients as ONE GEMM into one (3d, d) buffer: the parameters lie back
            # to back, their gradients may as well (each .grad is a row block of it); 3d x d output tiles instead of three
            # launches of d x d ones (the K dimension is 2048 tokens: short, so tiles are what fills the chip)
            dwp = alloc.empty((3 * d, d), dtype=wq.dtype, device=dev)
            dbp = alloc.empty(3 * d, dtype=bq.dtype, device=dev)
            dd = ConvDesc(rows, 1, 1, d, 1, 1, 3 * d, 1, 1, 1, 0, d, 3 * d)
            ops._wgrad(dqkv, 3 * d, ln1y, d, wq, dd, overlap=ops.OVERLAP_WGRAD, want_bias=True, out=(dwp, dbp))
            for i in range(3):
                grads[2 + 2 * i], grads[3 + 2 * i] = dwp[i * d:(i + 1) * d], dbp[i * d:(i + 1) * d]
        else:
            for i, (dy, w, b) in enumerate(((dq, wq, bq), (dk_, wk, bk), (dv, wv, bv))):
                wgrad(dy, ln1y, w, b, 2 + 2 * i, d, d, ldq)
        _linear_dgrad(dqkv, wqkv, dln1, rows, d, 3 * d, s)
        if need[0] or need[1]:
            grads[0], grads[1] = _ln_param_grads(x, st1, dln1, g1, b1, rows, d)
        dx = alloc.empty_like(x)
        lib.layernorm_bwd_input(x.data_ptr(), g1.data_ptr(), st1.data_ptr(), dln1.data_ptr(), dx1.data_ptr(), dx.data_ptr(),
                                None, 0.0, 0, None, rows, d, s)
        if ops.OVERLAP_WGRAD:
            ops._join_side(dev)            # (no-op in deferred-join mode: TrainStep joins once after backward)
        return (dx, None, None, None, None, *grads)


def transformer_block(x, heads, ps, eps, params, training=True):
    """ps = (attention, out_proj, mlp) dropout probabilities; params in module order: ln_input (w, b), que/key/val/out_proj
    (w, b each), ln_output (w, b), mlp[0] (w, b), mlp[2] (w, b)."""
    ps = tuple(float(p) if training else 0.0 for p in ps)
    seeds = tuple(next_seed() if p > 0 else 0 for p in ps)
    return _TransformerBlock.apply(x, heads, ps, tuple(float(e) for e in eps), seeds, *params)


class _PoolTokens(Function):
    """AdaptiveAvgPool2d(8,8) of both streams written as one (B,128,C) token tensor (rgb tokens first).
    skip=True also returns the two maps themselves (see ops._ConvBnAct): their other consumer (Add2) takes those aliases, so its
    gradient arrives here and is added inside the pool-gradient kernel -- one pass instead of a pass plus the autograd engine's
    fan-out accumulation (an ATen add over a full map, 8 per step)."""

    @staticmethod
    def forward(ctx, rgb, ir, skip):
        ctx.set_materialize_grads(False)
        ctx.src = rgb.dtype                                  # (bf16 storage: the pooling kernels read fp32 for now)
        rgb_in, ir_in = rgb, ir
        rgb, ld0 = rows_of(ops.raw_cast(rgb, torch.float32))
        ir, ld1 = rows_of(ops.raw_cast(ir, torch.float32))
        n, h, w, c = rgb.shape
        tok = alloc.empty((n, 128, c), dtype=rgb.dtype, device=rgb.device)
        s = _stream()
        lib.avgpool8_fwd(rgb.data_ptr(), ld0, n, h, w, c, tok.data_ptr(), 128 * c, c, s)
        lib.avgpool8_fwd(ir.data_ptr(), ld1, n, h, w, c, tok.data_ptr() + 4 * 64 * c, 128 * c, c, s)
        ctx.shape = (n, h, w, c)
        return (tok, rgb_in, ir_in) if skip else tok

    @staticmethod
    def backward(ctx, g, g_rgb=None, g_ir=None):
        n, h, w, c = ctx.shape
        if g is None:
            return g_rgb, g_ir, None
        g = g.contiguous()
        s = _stream()
        outs = []
        for i, gs in enumerate((g_rgb, g_ir)):
            d = alloc.empty((n, h, w, c), dtype=g.dtype, device=g.device)
            lds = 0
            if gs is not None:
                gs, lds = rows_of(ops.raw_cast(gs, torch.float32))
                if lds % 4 != 0 or gs.data_ptr() % 16 != 0:
                    gs, lds = gs.contiguous(), c
            lib.avgpool8_bwd_acc(g.data_ptr() + 4 * i * 64 * c, 128 * c, c, gs.data_ptr() if gs is not None else None, lds, d.data_ptr(), c,
                                 n, h, w, c, s)
            outs.append(ops.raw_cast(d, ctx.src))
        return outs[0], outs[1], None


def pool_tokens(rgb, ir, skip=False):
    return _PoolTokens.apply(rgb, ir, skip)


class _SplitTokens(Function):
    """(B,128,C) -> two contiguous (B,8,8,C) maps (rgb half, ir half)."""

    @staticmethod
    def forward(ctx, tok):
        tok = tok.contiguous()
        b, t, c = tok.shape
        s = _stream()
        outs = []
        for i in range(2):
            o = alloc.empty((b, 8, 8, c), dtype=tok.dtype, device=tok.device)
            lib.copy2d(tok.data_ptr() + 4 * i * 64 * c, 128 * c, o.data_ptr(), 64 * c, b, 64 * c, s)
            outs.append(o)
        return tuple(outs)

    @staticmethod
    def backward(ctx, g0, g1):
        b, c = g0.shape[0], g0.shape[-1]
        s = _stream()
        dt = alloc.empty((b, 128, c), dtype=g0.dtype, device=g0.device)
        for i, g in enumerate((g0, g1)):
            g = g.contiguous()
            lib.copy2d(g.data_ptr(), 64 * c, dt.data_ptr() + 4 * i * 64 * c, 128 * c, b, 64 * c, s)
        return dt


def split_tokens(tok):
    return _SplitTokens.apply(tok)


class _UpsampleAdd(Function):
    """x + bilinear(tok 8x8 -> HxW): the fusion transformer's output path fused with Add2."""

    @staticmethod
    def forward(ctx, x, tok):
        ctx.src = x.dtype
        x, ldx = rows_of(ops.raw_cast(x, torch.float32))
        tok = tok.contiguous()
        n, h, w, c = x.shape
        out = alloc.empty((n, h, w, c), dtype=x.dtype, device=x.device)
        lib.upsample_add_fwd(x.data_ptr(), ldx, tok.data_ptr(), 64 * c, c, out.data_ptr(), c, n, h, w, c, _stream())
        ctx.shape = (n, h, w, c)
        return ops.raw_cast(out, ctx.src)

    @staticmethod
    def backward(ctx, g):
        n, h, w, c = ctx.shape
        g0 = g
        g, ldg = rows_of(ops.raw_cast(g, torch.float32))
        dtok = alloc.empty((n, 8, 8, c), dtype=g.dtype, device=g.device)
        lib.upsample_add_bwd(g.data_ptr(), ldg, dtok.data_ptr(), 64 * c, c, n, h, w, c, _stream())
        return ops.raw_cast(g0, ctx.src), dtok


def upsample_add(x, tok):
    return _UpsampleAdd.apply(x, tok)


def upsample_only(tok, h, w):
    """bilinear(tok) alone, no gradient (API completeness for callers that want the FFM/GPT maps themselves)."""
    tok = tok.contiguous()
    n, c = tok.shape[0], tok.shape[-1]
    out = alloc.empty((n, h, w, c), dtype=tok.dtype, device=tok.device)
    lib.upsample_add_fwd(None, 0, tok.data_ptr(), 64 * c, c, out.data_ptr(), c, n, h, w, c, _stream())
    return out


def highpass_keep_mask(rows=8, cols=8):
    """Bit (u*8+v) set = unshifted DFT bin (u,v) survives the reference's high-pass box (common.py:43-50), evaluated with
    Python slice semantics exactly as the reference indexes its fftshift-ed spectrum."""
    crow, ccol = rows // 2, cols // 2
    thr = crow + ccol // 4
    keep = [[True] * cols for _ in range(rows)]
    for r in list(range(rows))[crow - thr:crow + thr]:
        for c in list(range(cols))[ccol - thr:ccol + thr]:
            keep[r][c] = False
    mask = 0
    for u in range(rows):
        for v in range(cols):
            su, sv = (u + rows // 2) % rows, (v + cols // 2) % cols      # position after fftshift
            if keep[su][sv]:
                mask |= 1 << (u * 8 + v)
    return mask


def ffm_highpass_mul(pooled):
    """pooled (B,64,C) -> high(pooled).half() * pooled, no gradient (feeds pattenLoss only)."""
    pooled = pooled.contiguous()
    b, _, c = pooled.shape
    out = alloc.empty_like(pooled)
    lib.ffm_highpass(pooled.data_ptr(), out.data_ptr(), b, c, highpass_keep_mask(), _stream())
    return out


def separation_loss(m_rgb, m_ir, m_rgb_hi, m_ir_hi):
    ts = [t.contiguous() for t in (m_rgb, m_ir, m_rgb_hi, m_ir_hi)]
    b = ts[0].shape[0]
    out = alloc.empty((), dtype=torch.float32, device=ts[0].device)
    lib.separation_loss(ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr(), ts[3].data_ptr(), b, out.data_ptr(),
                        _stream())
    return out


def fusion_stats(in_rgb, in_ir, tok):
    """(SSIMloss, Entropy_loss, ContrastiveValue) as a 3-vector, no gradient."""
    a, lda = rows_of(ops.raw_cast(in_rgb, torch.float32))
    b, ldb = rows_of(ops.raw_cast(in_ir, torch.float32))
    tok = tok.contiguous()
    n, h, w, c = a.shape
    ws = scratch(lib.fusion_stats_workspace() // 4 + 4, a.device, slot=2)
    assert ws.data_ptr() % 8 == 0
    out = alloc.empty(3, dtype=torch.float32, device=a.device)
    lib.fusion_stats(a.data_ptr(), lda, b.data_ptr(), ldb, tok.data_ptr(), n, h, w, c, ws.data_ptr(), out.data_ptr(),
                     _stream())
    return out
