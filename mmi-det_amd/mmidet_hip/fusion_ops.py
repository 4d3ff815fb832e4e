"""autograd glue for the fusion stack (GPT / GPT1_fourier, CBM / IGM statistics) over the C ABI.  Token tensors are
(B,128,C) or (rows,C) contiguous fp32; spatial tensors NHWC."""
import torch
from torch.autograd import Function

from . import lib
from .ops import _nrows, _stream, grad_like, rows_of, scratch

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
        y = torch.empty_like(x)
        stats = torch.empty((rows, 2), dtype=x.dtype, device=x.device)
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
        dx = torch.empty_like(x)
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
        y = torch.empty_like(x)
        lib.gelu_fwd(x.data_ptr(), y.data_ptr(), x.numel(), _stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        lib.gelu_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), _stream())
        return dx


def gelu(x):
    return _Gelu.apply(x)


class _Sigmoid(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        lib.sigmoid_fwd(x.data_ptr(), y.data_ptr(), x.numel(), _stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        lib.sigmoid_bwd(y.data_ptr(), dy.data_ptr(), dx.data_ptr(), y.numel(), _stream())
        return dx


def sigmoid(x):
    return _Sigmoid.apply(x)


class _Mul(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        assert a.shape == b.shape
        o = torch.empty_like(a)
        lib.mul(a.data_ptr(), b.data_ptr(), o.data_ptr(), a.numel(), _stream())
        ctx.save_for_backward(a, b)
        return o

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        da, db = torch.empty_like(a), torch.empty_like(b)
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
        out = torch.empty_like(a)
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
            da = torch.empty_like(g)
            lib.dropout(g.data_ptr(), None, 0, da.data_ptr(), g.numel(), p, seed, seed_state(g.device).data_ptr(),
                        _stream())
        else:
            da = g
        db = None
        if bshape is not None and ctx.needs_input_grad[1]:
            rows = g.numel() // bmod
            db = torch.empty(bshape, dtype=g.dtype, device=g.device)
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
        out = torch.empty_like(q)
        probs = torch.empty((b, heads, t, t), dtype=q.dtype, device=q.device)
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
        dq, dk_, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        lib.attention_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), probs.data_ptr(), do.data_ptr(), dq.data_ptr(),
                          dk_.data_ptr(), dv.data_ptr(), b, heads, c // heads, c, p, seed,
                          seed_state(q.device).data_ptr() if p > 0 else None, _stream())
        return dq, dk_, dv, None, None, None


def attention(q, k, v, heads, p=0.0, training=True):
    p = float(p) if training else 0.0
    return _Attention.apply(q, k, v, heads, p, next_seed() if p > 0 else 0)


class _PoolTokens(Function):
    """AdaptiveAvgPool2d(8,8) of both streams written as one (B,128,C) token tensor (rgb tokens first)."""

    @staticmethod
    def forward(ctx, rgb, ir):
        rgb, ld0 = rows_of(rgb)
        ir, ld1 = rows_of(ir)
        n, h, w, c = rgb.shape
        tok = torch.empty((n, 128, c), dtype=rgb.dtype, device=rgb.device)
        s = _stream()
        lib.avgpool8_fwd(rgb.data_ptr(), ld0, n, h, w, c, tok.data_ptr(), 128 * c, c, s)
        lib.avgpool8_fwd(ir.data_ptr(), ld1, n, h, w, c, tok.data_ptr() + 4 * 64 * c, 128 * c, c, s)
        ctx.shape = (n, h, w, c)
        return tok

    @staticmethod
    def backward(ctx, g):
        n, h, w, c = ctx.shape
        g = g.contiguous()
        s = _stream()
        d0 = torch.empty((n, h, w, c), dtype=g.dtype, device=g.device)
        d1 = torch.empty_like(d0)
        lib.avgpool8_bwd(g.data_ptr(), 128 * c, c, d0.data_ptr(), c, n, h, w, c, s)
        lib.avgpool8_bwd(g.data_ptr() + 4 * 64 * c, 128 * c, c, d1.data_ptr(), c, n, h, w, c, s)
        return d0, d1


def pool_tokens(rgb, ir):
    return _PoolTokens.apply(rgb, ir)


class _SplitTokens(Function):
    """(B,128,C) -> two contiguous (B,8,8,C) maps (rgb half, ir half)."""

    @staticmethod
    def forward(ctx, tok):
        tok = tok.contiguous()
        b, t, c = tok.shape
        s = _stream()
        outs = []
        for i in range(2):
            o = torch.empty((b, 8, 8, c), dtype=tok.dtype, device=tok.device)
            lib.copy2d(tok.data_ptr() + 4 * i * 64 * c, 128 * c, o.data_ptr(), 64 * c, b, 64 * c, s)
            outs.append(o)
        return tuple(outs)

    @staticmethod
    def backward(ctx, g0, g1):
        b, c = g0.shape[0], g0.shape[-1]
        s = _stream()
        dt = torch.empty((b, 128, c), dtype=g0.dtype, device=g0.device)
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
        x, ldx = rows_of(x)
        tok = tok.contiguous()
        n, h, w, c = x.shape
        out = torch.empty((n, h, w, c), dtype=x.dtype, device=x.device)
        lib.upsample_add_fwd(x.data_ptr(), ldx, tok.data_ptr(), 64 * c, c, out.data_ptr(), c, n, h, w, c, _stream())
        ctx.shape = (n, h, w, c)
        return out

    @staticmethod
    def backward(ctx, g):
        n, h, w, c = ctx.shape
        g, ldg = rows_of(g)
        dtok = torch.empty((n, 8, 8, c), dtype=g.dtype, device=g.device)
        lib.upsample_add_bwd(g.data_ptr(), ldg, dtok.data_ptr(), 64 * c, c, n, h, w, c, _stream())
        return g, dtok


def upsample_add(x, tok):
    return _UpsampleAdd.apply(x, tok)


def upsample_only(tok, h, w):
    """bilinear(tok) alone, no gradient (API completeness for callers that want the FFM/GPT maps themselves)."""
    tok = tok.contiguous()
    n, c = tok.shape[0], tok.shape[-1]
    out = torch.empty((n, h, w, c), dtype=tok.dtype, device=tok.device)
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
    out = torch.empty_like(pooled)
    lib.ffm_highpass(pooled.data_ptr(), out.data_ptr(), b, c, highpass_keep_mask(), _stream())
    return out


def separation_loss(m_rgb, m_ir, m_rgb_hi, m_ir_hi):
    ts = [t.contiguous() for t in (m_rgb, m_ir, m_rgb_hi, m_ir_hi)]
    b = ts[0].shape[0]
    out = torch.empty((), dtype=torch.float32, device=ts[0].device)
    lib.separation_loss(ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr(), ts[3].data_ptr(), b, out.data_ptr(),
                        _stream())
    return out


def fusion_stats(in_rgb, in_ir, tok):
    """(SSIMloss, Entropy_loss, ContrastiveValue) as a 3-vector, no gradient."""
    a, lda = rows_of(in_rgb)
    b, ldb = rows_of(in_ir)
    tok = tok.contiguous()
    n, h, w, c = a.shape
    ws = scratch(lib.fusion_stats_workspace() // 4 + 4, a.device, slot=2)
    assert ws.data_ptr() % 8 == 0
    out = torch.empty(3, dtype=torch.float32, device=a.device)
    lib.fusion_stats(a.data_ptr(), lda, b.data_ptr(), ldb, tok.data_ptr(), n, h, w, c, ws.data_ptr(), out.data_ptr(),
                     _stream())
    return out
