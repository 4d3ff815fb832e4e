"""GPU parity of the twin-lane launch form (mmidet_hip/twin_ops.py; include/mmidet_hip.h "channel maps and twin launches"): the
RGB and IR backbone copies of a layer run as ONE set of launches over a twin tensor (N,H,W,2,C).  Every case is checked per
lane against a plain fp32 torch CPU evaluation of the reference call site (models/common.py:108-125 Conv, 602-613 Bottleneck,
637-651 C3, 681-693 SPP, 696-709 Focus), forward and every gradient, tolerance 1e-3 relative as BASELINE.json states (measured
1e-6..1e-5); the whole graph is checked against the lane form of the same model."""
import copy

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import tiny_cfg
from test_ops_gpu import close, dev, nchw, nhwc

pytestmark = pytest.mark.gpu


class RefConv(nn.Module):
    """models/common.py:108-125 of the reference in plain torch."""

    def __init__(self, c1, c2, k=1, s=1):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, k // 2, bias=False)
        self.bn = nn.BatchNorm2d(c2, eps=1e-3, momentum=0.03)
        self.act = nn.SiLU()

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))


class RefBottleneck(nn.Module):
    def __init__(self, c1, c2, shortcut=True, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1, self.cv2 = RefConv(c1, c_, 1, 1), RefConv(c_, c2, 3, 1)
        self.add = shortcut and c1 == c2

    def forward(self, x):
        return x + self.cv2(self.cv1(x)) if self.add else self.cv2(self.cv1(x))


class RefC3(nn.Module):
    def __init__(self, c1, c2, n=1, shortcut=True):
        super().__init__()
        c_ = c2 // 2
        self.cv1, self.cv2, self.cv3 = RefConv(c1, c_), RefConv(c1, c_), RefConv(2 * c_, c2)
        self.m = nn.Sequential(*[RefBottleneck(c_, c_, shortcut, e=1.0) for _ in range(n)])

    def forward(self, x):
        return self.cv3(torch.cat((self.m(self.cv1(x)), self.cv2(x)), 1))


class RefSPP(nn.Module):
    def __init__(self, c1, c2):
        super().__init__()
        c_ = c1 // 2
        self.cv1, self.cv2 = RefConv(c1, c_), RefConv(c_ * 4, c2)
        self.m = nn.ModuleList([nn.MaxPool2d(k, 1, k // 2) for k in (5, 9, 13)])

    def forward(self, x):
        x = self.cv1(x)
        return self.cv2(torch.cat([x] + [m(x) for m in self.m], 1))


class RefFocus(nn.Module):
    def __init__(self, c1, c2, k=3):
        super().__init__()
        self.conv = RefConv(c1 * 4, c2, k, 1)

    def forward(self, x):
        return self.conv(torch.cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1))


def randomise(ref, g):
    """Non-trivial BatchNorm parameters and running statistics (the defaults 1 / 0 would hide a mixed-up parameter block)."""
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.copy_(0.5 + torch.rand(m.weight.shape, generator=g))
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
    return ref


def native_of(ref, cls, *args):
    """The package's module with the reference module's parameters and buffers (same state_dict keys)."""
    m = cls(*args)
    m.load_state_dict(ref.state_dict(), strict=True)
    for mod in m.modules():              # (utils/torch_utils.py:149-151 of the reference: what Model's initialisation sets)
        if isinstance(mod, nn.BatchNorm2d):
            mod.eps, mod.momentum = 1e-3, 0.03
    return m.to(dev()).train()


def twin_tensor(xa, xb):
    """Two NCHW CPU tensors -> the twin tensor (N,H,W,2,C) on the device."""
    return torch.stack((nhwc(xa), nhwc(xb)), 3).to(dev())


def compare(refs, nats, outs_ref, out_twin, xs_ref, x_twin, tol=1e-3):
    """Per lane: output, input gradient, every parameter gradient and every buffer."""
    for g in range(2):
        close(nchw(out_twin[..., g, :]), outs_ref[g], tol, 'lane %d output' % g)
        if xs_ref[g].grad is not None:
            close(nchw(x_twin.grad[..., g, :]), xs_ref[g].grad, tol, 'lane %d dx' % g)
        pr, pn = dict(refs[g].named_parameters()), dict(nats[g].named_parameters())
        assert pr.keys() == pn.keys()
        for k in pr:
            assert pn[k].grad is not None, 'lane %d: no gradient for %s' % (g, k)
            close(pn[k].grad, pr[k].grad, tol, 'lane %d d(%s)' % (g, k))
        br, bn_ = dict(refs[g].named_buffers()), dict(nats[g].named_buffers())
        for k in br:
            if br[k].dtype.is_floating_point:
                close(bn_[k], br[k], tol, 'lane %d buffer %s' % (g, k))
            else:
                assert int(bn_[k]) == int(br[k]), 'lane %d buffer %s' % (g, k)


def run_case(make_ref, cls, args, xshape, seed, pack=False, call=None, tol=1e-3):
    from mmidet_hip import ops
    g = torch.Generator().manual_seed(seed)
    refs = [randomise(make_ref(), g).train() for _ in range(2)]
    for r in refs:      # distinct weights per lane
        with torch.no_grad():
            for p in r.parameters():
                if p.dim() == 4:
                    p.copy_(torch.randn(p.shape, generator=g) / (p[0].numel()) ** 0.5)
    nats = [native_of(r, cls, *args) for r in refs]
    if pack:
        for n_ in nats:
            assert ops.pack_pair(n_) == 1
    xs = [torch.randn(xshape, generator=g).requires_grad_() for _ in range(2)]
    outs = [r(x) for r, x in zip(refs, xs)]
    gys = [torch.randn(o.shape, generator=g) for o in outs]
    for o, gy in zip(outs, gys):
        o.backward(gy)
    xt = twin_tensor(xs[0].detach(), xs[1].detach()).requires_grad_()
    assert nats[0].twin_ok(nats[1])
    out = call(nats[0], nats[1], xt) if call is not None else nats[0].twin(nats[1], xt)
    out.backward(twin_tensor(gys[0], gys[1]))
    torch.cuda.synchronize()
    compare(refs, nats, outs, out, xs, xt, tol)


CONV_CASES = [
    # N, H, W, Cin, Cout, k, s
    (2, 20, 24, 64, 64, 3, 1),       # uniform-tap loaders, 3x3
    (2, 20, 24, 64, 128, 3, 2),      # stride 2: dgrad per parity class
    (1, 17, 19, 32, 48, 3, 1),       # ragged tiles, Cout not a multiple of the tile
    (2, 16, 16, 128, 64, 1, 1),      # 1x1
    (4, 40, 40, 128, 128, 3, 1),     # 128x128 tiles
    (2, 32, 32, 12, 32, 3, 1),       # Focus-like: Cin = 12 (cursor loaders), lane offset of 48 bytes
    (2, 12, 12, 20, 36, 1, 1),       # channel counts that are multiples of 4 only
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_twin_conv_matches_the_reference_conv_per_lane(case):
    from models.common import Conv
    N, H, W, Cin, Cout, k, s = case
    run_case(lambda: RefConv(Cin, Cout, k, s), Conv, (Cin, Cout, k, s), (N, Cin, H, W), sum(case))


@pytest.mark.parametrize('slots', [6, 40])
def test_twin_conv_on_the_stream_k_schedule(slots):
    """Both problems on the stream-K schedule (forced onto small shapes): each problem owns half of the grid and its own
    arrival counters and partial-tile slots; run twice so that the second launch finds the counters the first one left."""
    from mmidet_hip import lib
    from models.common import Conv
    lib.set_streamk_slots(slots)
    try:
        for rep in range(2):
            run_case(lambda: RefConv(64, 64, 3, 1), Conv, (64, 64, 3, 1), (2, 64, 20, 24), 77 + rep)
            run_case(lambda: RefConv(128, 256, 1, 1), Conv, (128, 256, 1, 1), (2, 128, 16, 16), 78 + rep)
    finally:
        lib.set_streamk_slots(0)


@pytest.mark.parametrize('shortcut', [True, False])
def test_twin_bottleneck(shortcut):
    from models.common import Bottleneck
    run_case(lambda: RefBottleneck(64, 64, shortcut, e=1.0), Bottleneck, (64, 64, shortcut, 1, 1.0), (2, 64, 20, 20), 5 + shortcut)


@pytest.mark.parametrize('case', [(64, 64, 2, True, 20), (128, 128, 1, False, 12), (32, 64, 3, True, 16)])
def test_twin_c3_matches_the_reference_c3_per_lane(case):
    """cv1 | cv2 of both lanes as ONE GEMM with four BatchNorm parameter blocks, the lanes' concat buffers side by side and written
    in place, the shortcut gradients in the dgrad epilogue."""
    from models.common import C3
    c1, c2, n, shortcut, hw = case
    run_case(lambda: RefC3(c1, c2, n, shortcut), C3, (c1, c2, n, shortcut), (2, c1, hw, hw), sum(map(int, case)), pack=True)


def test_twin_spp():
    from models.common import SPP
    run_case(lambda: RefSPP(64, 64), SPP, (64, 64), (2, 64, 20, 20), 11)


def test_twin_focus_takes_two_images():
    from models.common import Focus
    from mmidet_hip import ops  # noqa: F401
    g = torch.Generator().manual_seed(3)
    refs = [randomise(RefFocus(3, 32, 3), g).train() for _ in range(2)]
    nats = [native_of(r, Focus, 3, 32, 3) for r in refs]
    xs = [torch.rand((2, 3, 32, 48), generator=g).requires_grad_() for _ in range(2)]
    outs = [r(x) for r, x in zip(refs, xs)]
    gys = [torch.randn(o.shape, generator=g) for o in outs]
    for o, gy in zip(outs, gys):
        o.backward(gy)
    xa = nhwc(xs[0].detach()).to(dev()).requires_grad_()
    xb = nhwc(xs[1].detach()).to(dev())                      # the IR image carries no gradient
    out = nats[0].twin(nats[1], xa, xb)
    out.backward(twin_tensor(gys[0], gys[1]))
    torch.cuda.synchronize()
    for q in range(2):
        close(nchw(out[..., q, :]), outs[q], what='lane %d output' % q)
        for (k, pr), (_, pn) in zip(refs[q].named_parameters(), nats[q].named_parameters()):
            close(pn.grad, pr.grad, what='lane %d d(%s)' % (q, k))
    close(nchw(xa.grad), xs[0].grad, what='dx (rgb image)')


def test_twin_lane_glue_ops():
    """pool-to-tokens, the Add2 pair, the neck's Add of the two lanes, stack / lanes: against the single-lane ops they replace."""
    from mmidet_hip import fusion_ops as F2
    from mmidet_hip import ops
    from mmidet_hip import twin_ops as T2
    g = torch.Generator().manual_seed(9)
    d = dev()
    a = torch.randn((2, 16, 24, 32), generator=g).to(d)
    b = torch.randn((2, 16, 24, 32), generator=g).to(d)
    ta, tb = torch.randn((2, 8, 8, 32), generator=g).to(d), torch.randn((2, 8, 8, 32), generator=g).to(d)
    # reference: single-lane ops
    a1, b1, ta1, tb1 = (t.clone().requires_grad_() for t in (a, b, ta, tb))
    tok1 = F2.pool_tokens(a1, b1)
    u1, v1 = F2.upsample_add(a1, ta1), F2.upsample_add(b1, tb1)
    s1 = ops.add(u1, v1)
    (tok1.square().sum() + (s1 * s1).sum() + u1.sum() * 0.5).backward()
    # twin
    a2, b2, ta2, tb2 = (t.clone().requires_grad_() for t in (a, b, ta, tb))
    T = T2.stack(a2, b2)
    tok2, alias = T2.pool_tokens2(T, skip=True)
    U = T2.upsample_add2(alias, ta2, tb2)
    s2 = T2.add_lanes(U)
    ua, _ = T2.lanes(U)
    (tok2.square().sum() + (s2 * s2).sum() + ua.sum() * 0.5).backward()
    torch.cuda.synchronize()
    close(tok2, tok1, 1e-6, 'tokens')
    close(s2, s1, 1e-6, 'lane sum')
    for n_, p, q in (('a', a2, a1), ('b', b2, b1), ('tok a', ta2, ta1), ('tok b', tb2, tb1)):
        close(p.grad, q.grad, 1e-5, 'd(%s)' % n_)


def _model(twin, kind='fourier', width=None):
    from models.yolo_test import Model
    from oracle import portable_init
    from oracle.ref_model import Model as OModel
    cfg = tiny_cfg(kind)
    if width is not None:
        cfg['width_multiple'] = width
        if kind == 'fourier':
            cfg['backbone'][6][3] = [int(128 * width)]
    sd = portable_init.fill_(OModel(cfg).state_dict())
    m = Model(copy.deepcopy(cfg))
    m.load_state_dict(sd, strict=True)
    for mod in m.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    m.twin = twin
    return m.to(dev()).train(), cfg


@pytest.mark.parametrize('kind,width', [('fourier', None), ('fourier', 0.5), ('add', None)])
def test_twin_model_equals_the_lane_form(kind, width):
    """The whole two-stream graph: twin launches against the two-lane form (MMIDET_TWIN=0) of the same weights -- predictions,
    auxiliary losses, loss and EVERY parameter gradient and BatchNorm buffer (the oracle comparison of the twin form itself is
    tests/test_model_gpu.py, which runs with the default, i.e. twin launches)."""
    from oracle import portable_init
    from oracle.ref_loss import scaled_hyp
    from utils.loss import ComputeLoss
    m1, cfg = _model(True, kind, width)
    m2, _ = _model(False, kind, width)
    assert m1._follower_of, 'no twin pairs planned'
    imgs, tg = portable_init.synth_batch(2, 128, cfg['nc'], per_image=4, seed=5)
    x = imgs.to(dev()).float() / 255
    res = []
    for m in (m1, m2):
        m.nc, m.gr, m.hyp = cfg['nc'], 1.0, scaled_hyp(cfg['nc'], 128)
        pred, comb = m(x[:, :3], x[:, 3:])
        loss, items = ComputeLoss(m)(pred, tg.to(dev()), comb.reshape(-1))
        loss.sum().backward()
        res.append((pred, comb, loss))
    torch.cuda.synchronize()
    for pa, pb in zip(res[0][0], res[1][0]):
        close(pa, pb, 1e-4, 'prediction')
    close(res[0][2], res[1][2], 1e-5, 'loss')
    if res[1][1].numel():
        close(res[0][1], res[1][1], 1e-4, 'Combine_loss')
    n_twin = 0
    for (k, pa), (_, pb) in zip(m1.named_parameters(), m2.named_parameters()):
        if pb.grad is None:
            assert pa.grad is None, k
            continue
        if float(pb.grad.abs().max()) < 1e-7:              # analytically zero (the key bias of a softmax attention): rounding noise
            assert float(pa.grad.abs().max()) < 1e-6, k
            continue
        close(pa.grad, pb.grad, 2e-3, 'd(%s)' % k)      # two fp32 summation orders through the whole depth (cf. test_model_gpu)
        n_twin += 1
    assert n_twin > 100
    for (k, ba), (_, bb) in zip(m1.named_buffers(), m2.named_buffers()):
        if ba.dtype.is_floating_point:
            close(ba, bb, 1e-4, 'buffer %s' % k)
        else:
            assert torch.equal(ba, bb), k


@pytest.mark.parametrize('switch', ['MMIDET_CAT_DEST', 'MMIDET_HEAD_VIEW'])
def test_copy_free_concat_and_detect_views_change_nothing(monkeypatch, switch):
    """Round 3: the neck's Concat layers are aliases of buffers their producers wrote into (models/common.py:740-748 of the
    reference copies), and Detect hands the loss its (B,na,ny,nx,no) tensors as strided views of the head convolutions' NHWC
    outputs (models/yolo_test.py:54-55 copies).  Both only remove copies: predictions, loss and every gradient are bit-identical
    to the copying forms (MMIDET_CAT_DEST=0 / MMIDET_HEAD_VIEW=0).  (Bit-identity needs the packed q/k/v projections, which the
    model sets up at its first forward: the three-GEMM fallback of the transformer blocks' backward is not run-to-run
    bit-identical -- DESIGN.md section 7.)"""
    from oracle import portable_init
    from oracle.ref_loss import scaled_hyp
    from utils.loss import ComputeLoss
    res = []
    for off in (False, True):
        if off:
            monkeypatch.setenv(switch, '0')
        m, cfg = _model(True)
        if switch == 'MMIDET_CAT_DEST':
            assert bool(m._cat_plan) == (not off)
        imgs, tg = portable_init.synth_batch(2, 128, cfg['nc'], per_image=4, seed=6)
        x = imgs.to(dev()).float() / 255
        m.nc, m.gr, m.hyp = cfg['nc'], 1.0, scaled_hyp(cfg['nc'], 128)
        pred, comb = m(x[:, :3], x[:, 3:])
        if switch == 'MMIDET_HEAD_VIEW':
            assert pred[0].is_contiguous() == off
        loss, _ = ComputeLoss(m)(pred, tg.to(dev()), comb.reshape(-1))
        loss.sum().backward()
        torch.cuda.synchronize()
        res.append((pred, loss, {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b)
    assert torch.equal(res[0][1], res[1][1])
    assert res[0][2].keys() == res[1][2].keys()
    for k in res[0][2]:
        assert torch.equal(res[0][2][k], res[1][2][k]), k
