"""GPU parity, op level: every HIP op (through the C ABI) against a plain fp32 torch CPU evaluation of the same
reference call site, forward and backward.  Tolerances are written per test; fp32 activations: 1e-3 relative
(BASELINE.json north_star), integer outputs: bit-exact."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

RTOL = 1e-3


def dev():
    assert torch.cuda.is_available(), 'GPU tests need the MI355X'
    return torch.device('cuda:0')


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def close(a, b, tol=RTOL, what=''):
    e = rel_err(a, b)
    if os.environ.get('MMI_ERRLOG'):      # optional: log every measured error (used to rank kernels by accuracy)
        with open(os.environ['MMI_ERRLOG'], 'a') as f:
            f.write('%-60s %-28s %.3e (tol %.0e)\n' % (os.environ.get('PYTEST_CURRENT_TEST', '')[:60], what, e, tol))
    assert e < tol, '%s relative L2 error %.3e >= %.1e' % (what, e, tol)
    # elementwise too, scaled by the tensor's magnitude
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    scale = float(b.abs().max()) + 1e-30
    m = float((a - b).abs().max()) / scale
    assert m < 10 * tol, '%s max abs error / max|ref| = %.3e' % (what, m)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def cl(w):
    return w.contiguous(memory_format=torch.channels_last)


CONV_CASES = [
    # N, H, W, Cin, Cout, k, s
    (2, 20, 24, 64, 64, 3, 1),
    (2, 20, 24, 64, 128, 3, 2),
    (1, 17, 19, 32, 48, 3, 1),     # odd sizes, ragged tiles
    (2, 21, 23, 32, 64, 3, 2),     # odd sizes with stride 2
    (2, 16, 16, 128, 256, 1, 1),
    (3, 8, 8, 256, 128, 1, 1),
    (2, 32, 32, 12, 32, 3, 1),     # Focus-like: K=108 (ragged K slab)
    (1, 24, 24, 3, 24, 3, 1),      # CEM conv2: scalar loader path
    (1, 24, 24, 24, 3, 3, 1),      # CEM conv3
    (2, 10, 10, 256, 33, 1, 1),    # Detect head: Cout=33
    (4, 40, 40, 128, 128, 3, 1),   # 128x128 tile path (M=6400)
    (2, 64, 64, 64, 64, 1, 1),
    (4, 128, 128, 12, 64, 3, 1),   # Focus at a size that takes the 128x64 dgrad tile with N = 12: stacked wave layout (W41)
    (4, 128, 130, 20, 32, 3, 1),   # the same with N = 20 and ragged rows
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd_bwd_with_bias(case):
    from mmidet_hip import ops
    N, H, W, Cin, Cout, k, s = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    yr = F.conv2d(xr, wr, br, s, k // 2)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)

    d = dev()
    xg = nhwc(x).to(d).requires_grad_()
    wg = cl(w).to(d).requires_grad_()
    bg = b.to(d).requires_grad_()
    yg = ops.conv_bias(xg, wg, bg, s)
    assert tuple(yg.shape) == (N, yr.shape[2], yr.shape[3], Cout)
    yg.backward(nhwc(gy).to(d))
    torch.cuda.synchronize()
    close(nchw(yg), yr, what='y')
    close(nchw(xg.grad), xr.grad, what='dx')
    close(wg.grad, wr.grad, what='dw')
    close(bg.grad, br.grad, what='db')


@pytest.mark.parametrize('case', [(2, 20, 24, 64, 64, 3, 1, 1), (2, 20, 24, 64, 128, 3, 2, 1), (2, 16, 16, 128, 64, 1, 1, 1),
                                  (1, 24, 24, 3, 24, 3, 1, 2), (1, 24, 24, 24, 3, 3, 1, 2), (4, 40, 40, 64, 128, 3, 1, 1),
                                  (2, 32, 32, 12, 32, 3, 1, 1)])
@pytest.mark.parametrize('residual', [False, True])
def test_conv_bn_act_train(case, residual):
    """SiLU/LeakyReLU(BN_train(conv(x))) [+ x]: models/common.py:108-125, 602-613, 788-799."""
    from mmidet_hip import ops
    N, H, W, Cin, Cout, k, s, act = case
    if residual and (Cin != Cout or s != 1):
        pytest.skip('residual needs matching shapes')
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    gamma, beta = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.2
    rm, rv = torch.randn(Cout, generator=g) * 0.1, torch.rand(Cout, generator=g) + 0.5
    xr, wr, gr, br = (t.clone().requires_grad_() for t in (x, w, gamma, beta))
    rmr, rvr = rm.clone(), rv.clone()
    z = F.batch_norm(F.conv2d(xr, wr, None, s, k // 2), rmr, rvr, gr, br, True, 0.03, 1e-3)
    outr = F.silu(z) if act == 1 else F.leaky_relu(z, 0.1)
    if residual:
        outr = outr + xr
    gy = torch.randn(outr.shape, generator=g)
    outr.backward(gy)

    d = dev()
    xg = nhwc(x).to(d).requires_grad_()
    wg, gg, bg = cl(w).to(d).requires_grad_(), gamma.to(d).requires_grad_(), beta.to(d).requires_grad_()
    rmg, rvg = rm.to(d), rv.to(d)
    nbt = torch.zeros((), dtype=torch.long, device=d)
    outg = ops.conv_bn_act(xg, wg, gg, bg, rmg, rvg, nbt, stride=s, act=act, residual=xg if residual else None)
    outg.backward(nhwc(gy).to(d))
    torch.cuda.synchronize()
    close(nchw(outg), outr, what='out')
    close(rmg, rmr, what='running_mean')
    close(rvg, rvr, what='running_var')
    assert int(nbt) == 1
    close(nchw(xg.grad), xr.grad, what='dx', tol=2e-3)
    close(wg.grad, wr.grad, what='dw', tol=2e-3)
    close(gg.grad, gr.grad, what='dgamma', tol=2e-3)
    close(bg.grad, br.grad, what='dbeta', tol=2e-3)


def test_conv_bn_act_eval():
    from mmidet_hip import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 32, 12, 12, generator=g)
    w = torch.randn(64, 32, 3, 3, generator=g) / 17
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    rm, rv = torch.randn(64, generator=g) * 0.1, torch.rand(64, generator=g) + 0.5
    ref = F.silu(F.batch_norm(F.conv2d(x, w, None, 1, 1), rm, rv, gamma, beta, False, 0.03, 1e-3))
    d = dev()
    out = ops.conv_bn_act(nhwc(x).to(d), cl(w).to(d), gamma.to(d), beta.to(d), rm.to(d), rv.to(d), None, training=False)
    close(nchw(out), ref, what='eval out')


def test_conv_on_channel_slice_input():
    """Inputs may be channel slices of a wider NHWC buffer (row stride > C): concat-free consumers."""
    from mmidet_hip import ops
    g = torch.Generator().manual_seed(9)
    buf = torch.randn(2, 96, 14, 14, generator=g)
    w = torch.randn(48, 64, 3, 3, generator=g) / 24
    ref = F.conv2d(buf[:, 32:], w, None, 1, 1)
    d = dev()
    xs = nhwc(buf).to(d)[..., 32:]
    out = ops.conv_bias(xs, cl(w).to(d), None, 1)
    close(nchw(out), ref, what='slice conv')


@pytest.mark.parametrize('shape', [(256, 128, 128), (2 * 128, 256, 1024), (130, 64, 24), (16 * 128, 1024, 1024)])
def test_linear(shape):
    """nn.Linear forward/backward: models/common.py:1167-1170, 1254, 1257."""
    from mmidet_hip import ops
    rows, K, Nn = shape
    g = torch.Generator().manual_seed(rows + K)
    x, w, b = torch.randn(rows, K, generator=g), torch.randn(Nn, K, generator=g) / K ** 0.5, torch.randn(Nn, generator=g)
    xr, wr, br = (t.clone().requires_grad_() for t in (x, w, b))
    yr = F.linear(xr, wr, br)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    d = dev()
    xg, wg, bg = (t.to(d).requires_grad_() for t in (x, w, b))
    yg = ops.linear(xg, wg, bg)
    yg.backward(gy.to(d))
    close(yg, yr, what='y')
    close(xg.grad, xr.grad, what='dx')
    close(wg.grad, wr.grad, what='dw')
    close(bg.grad, br.grad, what='db')


def test_elementwise_layout_ops():
    from mmidet_hip import ops
    d = dev()
    g = torch.Generator().manual_seed(3)
    # strided NCHW views -> NHWC (train.py:743-745)
    imgs = torch.rand(2, 6, 32, 48, generator=g)
    x6 = imgs.to(d)
    for sl in (slice(0, 3), slice(3, 6)):
        y = ops.nchw_to_nhwc(x6[:, sl])
        assert torch.equal(y.cpu(), nhwc(imgs[:, sl]))
    # the loader's uint8 (B,6,H,W) batch -> both fp32 NHWC images in one pass: bit-equal to .float()/255 + slices
    u8 = torch.randint(0, 256, (3, 6, 20, 36), dtype=torch.uint8, generator=g)
    u8[0, :, 0, :6] = torch.tensor([0, 1, 127, 128, 254, 255], dtype=torch.uint8)
    rgb, ir = ops.u8_pair_to_nhwc(u8.to(d))
    ref6 = u8.float() / 255.0
    assert torch.equal(rgb.cpu(), nhwc(ref6[:, :3])) and torch.equal(ir.cpu(), nhwc(ref6[:, 3:]))
    assert ops.nchw_to_nhwc(rgb) is rgb        # already in the kernels' layout: Model.forward does not convert again
    # Focus space-to-depth (common.py:708)
    x = torch.randn(2, 3, 16, 24, generator=g, requires_grad=True)
    ref = torch.cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    xg = nhwc(x.detach()).to(d).requires_grad_()
    yg = ops.space_to_depth(xg)
    yg.backward(nhwc(gy).to(d))
    assert torch.equal(nchw(yg).cpu(), ref.detach())
    assert torch.equal(nchw(xg.grad).cpu(), x.grad)
    # add / concat / upsample
    a, b = torch.randn(2, 40, 7, 9, generator=g, requires_grad=True), torch.randn(2, 40, 7, 9, generator=g, requires_grad=True)
    c = torch.randn(2, 24, 7, 9, generator=g, requires_grad=True)
    ref = F.interpolate(torch.cat([a + b, c], 1), scale_factor=2, mode='nearest')
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    ag, bg, cg = (nhwc(t.detach()).to(d).requires_grad_() for t in (a, b, c))
    out = ops.upsample2x(ops.concat([ops.add(ag, bg), cg]))
    out.backward(nhwc(gy).to(d))
    assert torch.equal(nchw(out).cpu(), ref.detach())
    for tg, tr, nm in ((ag, a, 'da'), (bg, b, 'db'), (cg, c, 'dc')):
        close(nchw(tg.grad), tr.grad, what=nm, tol=1e-6)
    # Detect view/permute (yolo_test.py:54-55)
    p = torch.randn(2, 33, 5, 6, generator=g, requires_grad=True)
    ref = p.view(2, 3, 11, 5, 6).permute(0, 1, 3, 4, 2).contiguous()
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    pg = nhwc(p.detach()).to(d).requires_grad_()
    og = ops.head_permute(pg, 3)
    og.backward(gy.to(d))
    assert torch.equal(og.cpu(), ref.detach())
    assert torch.equal(nchw(pg.grad).cpu(), p.grad)


@pytest.mark.parametrize('hw', [(20, 20), (4, 4), (2, 2), (13, 7)])
def test_spp_pool(hw):
    """cat(x, mp5, mp9, mp13): models/common.py:681-693, incl. the backward arg-max routing."""
    from mmidet_hip import ops
    h, w = hw
    g = torch.Generator().manual_seed(h * w)
    x = torch.randn(2, 64, h, w, generator=g, requires_grad=True)
    ref = torch.cat([x] + [F.max_pool2d(x, k, 1, k // 2) for k in (5, 9, 13)], 1)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    d = dev()
    xg = nhwc(x.detach()).to(d).requires_grad_()
    out = ops.spp_pool(xg)
    out.backward(nhwc(gy).to(d))
    assert torch.equal(nchw(out).cpu(), ref.detach())
    close(nchw(xg.grad), x.grad, what='dx', tol=1e-5)


@pytest.mark.parametrize('tag,bs,per', [('b16x32', 16, 32), ('b16x8', 16, 8), ('b1x1', 1, 1), ('b4x0', 4, 0)])
def test_build_targets_bit_exact(tag, bs, per):
    """The integer assignment kernel against the fixture written by the reference itself (utils/loss.py:189-245)."""
    from mmidet_hip import loss_ops
    from oracle import portable_init
    gfile = np.load(os.path.join(GOLDEN, 'build_targets.npz'))
    d = dev()
    _, tg = portable_init.synth_batch(bs, 32, 6, per_image=per, seed=7)
    anchors = torch.tensor([[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]).float()
    anchors = (anchors.view(3, 3, 2) / torch.tensor([8., 16., 32.]).view(3, 1, 1)).to(d)
    grids = [(640 // s, 640 // s) for s in (8, 16, 32)]
    tcls, tbox, idx, anch = loss_ops.build_targets(tg.to(d), anchors, grids, 4.0)
    for i in range(3):
        assert np.array_equal(tcls[i].cpu().numpy(), gfile['%s.tcls%d' % (tag, i)])
        assert np.array_equal(torch.stack(idx[i]).cpu().numpy(), gfile['%s.idx%d' % (tag, i)])
        assert np.array_equal(tbox[i].cpu().numpy(), gfile['%s.tbox%d' % (tag, i)])
        assert np.array_equal(anch[i].cpu().numpy(), gfile['%s.anch%d' % (tag, i)])


@pytest.mark.parametrize('case', [(2, 20, 24, 64, 64, 3, 1), (2, 21, 23, 32, 64, 3, 2), (2, 16, 16, 128, 256, 1, 1), (4, 40, 40, 128, 128, 3, 1),
                                  (1, 33, 17, 96, 64, 3, 1)])
@pytest.mark.parametrize('mode', [0, 2, 3], ids=['fp32', 'bf16x6', 'bf16x9'])
def test_uniform_loaders_are_bit_identical_to_the_general_ones(case, mode):
    """mmi_set_uniform_loaders: the uniform-tap (forward, dgrad) and pixel-table (wgrad) loaders only change how a tile's
    addresses are formed: for 1x1 layers and for every weight gradient the arithmetic and its order are those of the general
    cursor-based loaders -- y, dx, dw, statistics equal bit for bit.  For 3x3 layers the uniform-tap loaders walk the K slabs
    channel-slab major since round 4 (csrc/igemm_kernel.h: MMI_KORD; the general ones tap major): the same products in another
    summation order, so forward and dgrad agree to what two fp32 summation orders differ by -- including zero padding at the
    borders, ragged tiles and the stride-2 parity classes.  All arithmetics that have these loaders."""
    from mmidet_hip import lib, ops
    lib.set_gemm_precision(mode)
    N, H, W, Ci, Co, k, s = case
    d = dev()
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, H, W, Ci, generator=g).to(d)
    w = (torch.randn(Co, k, k, Ci, generator=g) / (k * k * Ci) ** 0.5).to(d)
    desc = ops._desc((N, H, W, Ci), Co, k, s, Ci, Co)
    dy = torch.randn(N, desc.Ho, desc.Wo, Co, generator=g).to(d)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    try:
        for on in (1, 0):
            lib.set_uniform_loaders(on)
            y, dx, dw = torch.empty_like(dy), torch.empty_like(x), torch.empty_like(w)
            part = torch.empty(lib.conv_fwd_row_blocks(desc) * 2 * Co + 64 * 2 * Co, device=d)
            nb = lib.conv_wgrad_workspace(desc)
            ws = torch.zeros(max(nb // 4, 1), device=d)     # (arrival counters at its head: zero-filled once)
            ops.conv_fwd(x, w, None, y, part, desc, st)
            ops.conv_dgrad(dy, w, dx, desc, st)
            lib.conv_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), nb, desc, st)
            torch.cuda.synchronize()
            outs.append((y, dx, dw, part[:lib.conv_fwd_row_blocks(desc) * 2 * Co].clone()))
    finally:
        lib.set_uniform_loaders(1)
        lib.set_gemm_precision(0)
    for a, b, what in zip(outs[0], outs[1], ('y', 'dx', 'dw', 'BN statistics partials')):
        if k == 1 or what == 'dw':
            assert torch.equal(a, b), what
        else:
            close(a, b, tol=3e-6, what=what + ' (channel-slab major vs tap major K order)')
