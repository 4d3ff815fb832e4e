"""GPU parity: Contour Enhancement Module kernels (csrc/cem.hip) against the reference formulation
(models/common.py:751-911): EnhanceConv2d as a real 24->24 conv2d with sobel_weight*sobel_factor, AdaptiveModule3 end
to end (forward, all parameter gradients, BN running stats) at a size that spans several 16x16 tiles and ragged edges."""
import pytest
import torch
import torch.nn.functional as F

from test_ops_gpu import close, dev, nchw, nhwc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('hw', [(40, 56), (17, 19), (16, 16), (5, 3)])
def test_sobel_add_matches_conv2d_bank(hw):
    from mmidet_hip import ops
    from oracle.ref_model import EnhanceConv2d
    h, w = hw
    g = torch.Generator().manual_seed(h * w)
    ref = EnhanceConv2d(24, 24)
    ref.sobel_factor.data = torch.rand(24, 1, 1, 1, generator=g) + 0.5
    ref.bias.data = torch.randn(24, generator=g) * 0.1
    r = torch.randn(2, 24, h, w, generator=g, requires_grad=True)
    t = r + ref(r)
    gt = torch.randn(t.shape, generator=g)
    t.backward(gt)
    d = dev()
    rg = nhwc(r.detach()).to(d).requires_grad_()
    fg = ref.sobel_factor.detach().to(d).requires_grad_()
    bg = ref.bias.detach().to(d).requires_grad_()
    tg = ops.sobel_add(rg, fg, bg)
    tg.backward(nhwc(gt).to(d))
    close(nchw(tg), t, what='t', tol=1e-5)
    close(nchw(rg.grad), r.grad, what='dr', tol=1e-5)
    close(fg.grad, ref.sobel_factor.grad, what='dfactor', tol=1e-4)
    close(bg.grad, ref.bias.grad, what='dbias', tol=1e-5)


@pytest.mark.parametrize('shape', [(2, 48, 40), (1, 33, 17)])
def test_cem_module_end_to_end(shape):
    """AdaptiveModule3: product module (small-channel direct convs + stencil bank) vs the oracle module."""
    from models.common import AdaptiveModule3
    from oracle import portable_init
    from oracle.ref_model import AdaptiveModule3 as OCem
    b, h, w = shape
    o = OCem(3, 3)
    sd = portable_init.fill_({'Enhance.' + k: v for k, v in o.state_dict().items()})
    sd = {k[len('Enhance.'):]: v for k, v in sd.items()}
    o.load_state_dict(sd)
    m = AdaptiveModule3(3, 3)
    for bn in (m.bn2, m.bn3):
        bn.eps, bn.momentum = 1e-3, 0.03
    m.load_state_dict(sd, strict=True)
    m = m.to(dev()).train()
    o.train()
    g = torch.Generator().manual_seed(b * h)
    x = torch.rand(b, 3, h, w, generator=g)
    xo = x.clone().requires_grad_()
    yo = o(xo)
    gy = torch.randn(yo.shape, generator=g)
    yo.backward(gy)
    xg = nhwc(x).to(dev()).requires_grad_()
    assert m.sobel.is_standard_bank()
    yg = m(xg)
    yg.backward(nhwc(gy).to(dev()))
    close(nchw(yg), yo, what='CEM out', tol=1e-4)
    close(nchw(xg.grad), xo.grad, what='CEM dx', tol=1e-3)
    og = dict(o.named_parameters())
    for n, p in m.named_parameters():
        if p.grad is not None:
            close(p.grad, og[n].grad, what='d ' + n, tol=2e-3)
    for k in ('bn2.running_mean', 'bn2.running_var', 'bn3.running_mean', 'bn3.running_var'):
        close(m.state_dict()[k], o.state_dict()[k], what=k, tol=1e-5)


def test_nonstandard_bank_falls_back_to_the_real_conv():
    from models.common import AdaptiveModule3
    m = AdaptiveModule3(3, 3).to(dev())
    assert m.sobel.is_standard_bank()
    with torch.no_grad():
        m.sobel.sobel_weight[0, 0, 0, 0] += 1.0
    assert not m.sobel.is_standard_bank()
    x = torch.rand(1, 8, 8, 3, device=dev())
    assert m(x).shape == x.shape


@pytest.mark.parametrize('shape', [(2, 48, 40), (1, 33, 17), (3, 5, 3), (1, 16, 16)])
@pytest.mark.parametrize('train', [True, False])
def test_fused_forward_matches_the_unfused_chain(shape, train, monkeypatch):
    """csrc/cem.hip::cem_fused_fwd_kernel (x -> conv2 -> BN2 -> LeakyReLU -> stencil bank -> conv3 in one launch, r and t in LDS)
    against the four-kernel chain it replaces, on tiles with ragged edges and on an image smaller than one tile: output, every
    gradient and the BatchNorm running statistics.  Both are fp32 evaluations of the same sums in the same tap order, so the
    tolerance is rounding of the BN affine form only."""
    from mmidet_hip import ops
    from models.common import AdaptiveModule3
    b, h, w = shape
    g = torch.Generator().manual_seed(h * 7 + w)
    x = torch.rand(b, h, w, 3, generator=g).to(dev())
    gy = torch.randn(b, h, w, 3, generator=g).to(dev())
    res = []
    for fused in (False, True):
        monkeypatch.setattr(ops, 'CEM_FUSED', fused)
        torch.manual_seed(5)
        m = AdaptiveModule3(3, 3)
        with torch.no_grad():
            m.sobel.sobel_factor.uniform_(0.5, 1.5)
            m.sobel.bias.normal_(0, 0.1)
            for bn in (m.bn2, m.bn3):
                bn.weight.uniform_(0.5, 1.5)
                bn.bias.normal_(0, 0.1)
                bn.running_mean.normal_(0, 0.1)
                bn.running_var.uniform_(0.5, 1.5)
        m = m.to(dev()).train(train)
        xg = x.clone().requires_grad_()
        y = m(xg)
        y.backward(gy)
        res.append((y.detach(), xg.grad, {n: p.grad for n, p in m.named_parameters() if p.grad is not None},
                    {k: v.clone() for k, v in m.state_dict().items() if 'running' in k or 'tracked' in k}))
    (y0, dx0, g0, s0), (y1, dx1, g1, s1) = res
    close(y1, y0, what='out', tol=2e-6)
    close(dx1, dx0, what='dx', tol=2e-5)
    assert set(g0) == set(g1)
    for n in g0:
        close(g1[n], g0[n], what='d ' + n, tol=5e-5)
    for k in s0:
        close(s1[k].float(), s0[k].float(), what=k, tol=1e-6)


def test_fused_forward_inference_keeps_nothing():
    """Under no_grad in eval mode the fused kernel gets NULL for y2 / t / chansum: 1 read of x, 1 write of y3."""
    from models.common import AdaptiveModule3
    torch.manual_seed(1)
    m = AdaptiveModule3(3, 3).to(dev()).eval()
    x = torch.rand(2, 37, 29, 3, device=dev())
    with torch.no_grad():
        a = m(x)
    b = m(x)
    assert torch.equal(a, b)


@pytest.mark.parametrize('shape', [(2, 48, 40), (1, 33, 17), (3, 5, 3), (1, 16, 16), (2, 130, 70)])
def test_fused_backward_middle_matches_the_two_step_form(shape):
    """mmi_cem_bwd_mid (conv3's input gradient + the stencil bank's backward in one kernel, dt and the D maps in LDS) against
    mmi_conv_dgrad followed by mmi_sobel_add_bwd: dr, dfactor, dbias, on ragged tiles, a sub-tile image and many tiles per workgroup."""
    from mmidet_hip import lib, ops
    from mmidet_hip.ops import ConvDesc
    n, h, w = shape
    d = dev()
    g = torch.Generator().manual_seed(h * 3 + w)
    dy3 = torch.randn(n, h, w, 3, generator=g).to(d)
    w3 = (torch.randn(3, 3, 3, 24, generator=g) * 0.2).to(d)          # OHWI [3][9][24]
    cs = torch.randn(n, h, w, generator=g).to(d)
    f = (torch.rand(24, generator=g) + 0.5).to(d)
    s = torch.cuda.current_stream().cuda_stream
    dt = torch.empty(n, h, w, 24, device=d)
    ops.conv_dgrad(dy3, w3, dt, ConvDesc(n, h, w, 24, h, w, 3, 3, 3, 1, 1, 24, 3), s)
    dr0, df0, db0 = torch.empty_like(dt), torch.empty(24, device=d), torch.empty(24, device=d)
    ws = torch.empty(lib.sobel_add_bwd_workspace(n, h, w, 24) // 4 + 4, device=d)
    lib.sobel_add_bwd(dt.data_ptr(), 24, cs.data_ptr(), f.data_ptr(), dr0.data_ptr(), 24, df0.data_ptr(), db0.data_ptr(), ws.data_ptr(),
                      n, h, w, 24, s)
    dr1, df1, db1 = torch.empty_like(dt), torch.empty(24, device=d), torch.empty(24, device=d)
    ws1 = torch.empty(lib.cem_bwd_mid_workspace(n, h, w) // 4 + 4, device=d)
    lib.cem_bwd_mid(dy3.data_ptr(), w3.data_ptr(), cs.data_ptr(), f.data_ptr(), dr1.data_ptr(), df1.data_ptr(), db1.data_ptr(),
                    ws1.data_ptr(), None, None, None, None, None, n, h, w, s)
    torch.cuda.synchronize()
    close(dr1, dr0, what='dr', tol=2e-6)
    close(df1, df0, what='dfactor', tol=2e-5)
    close(db1, db0, what='dbias', tol=2e-5)
    # the same launch with BatchNorm2's backward reduction riding along: partials -> mmi_bn_act_bwd_apply, against the one-call
    # BatchNorm backward on the same dr
    y2 = torch.randn(n, h, w, 24, generator=g).to(d)
    mi = torch.cat([torch.randn(24, generator=g) * 0.2, torch.rand(24, generator=g) + 0.5]).to(d)
    gam, bet = (torch.rand(24, generator=g) + 0.5).to(d), (torch.randn(24, generator=g) * 0.2).to(d)
    rows = n * h * w
    dy_ref, dg_ref, dbt_ref = torch.empty_like(y2), torch.empty(24, device=d), torch.empty(24, device=d)
    ops._bn_act_bwd(y2, 24, dr0, 24, None, 0, 24, mi, gam, bet, dy_ref, (dg_ref, dbt_ref, None, None), rows, 24, ops.ACT_LEAKY, 0, s)
    nblk = lib.cem_bwd_mid_blocks(n, h, w)
    part = torch.empty(nblk * 48 + 64, device=d)
    dr2 = torch.empty_like(dt)
    lib.cem_bwd_mid(dy3.data_ptr(), w3.data_ptr(), cs.data_ptr(), f.data_ptr(), dr2.data_ptr(), df1.data_ptr(), db1.data_ptr(),
                    ws1.data_ptr(), y2.data_ptr(), mi.data_ptr(), gam.data_ptr(), bet.data_ptr(), part.data_ptr(), n, h, w, s)
    dy_got, dg_got, dbt_got = torch.empty_like(y2), torch.empty(24, device=d), torch.empty(24, device=d)
    lib.bn_act_bwd_apply(y2.data_ptr(), 24, dr2.data_ptr(), 24, mi.data_ptr(), gam.data_ptr(), bet.data_ptr(), part.data_ptr(), nblk,
                         dy_got.data_ptr(), 24, dg_got.data_ptr(), dbt_got.data_ptr(), rows, 24, ops.ACT_LEAKY, 0, s)
    torch.cuda.synchronize()
    assert torch.equal(dr2, dr1)
    close(dg_got, dg_ref, what='dgamma2', tol=2e-5)
    close(dbt_got, dbt_ref, what='dbeta2', tol=2e-5)
    close(dy_got, dy_ref, what='dy2', tol=2e-5)


@pytest.mark.parametrize('shape', [(2, 48, 40), (1, 33, 17), (3, 5, 3), (1, 16, 16), (2, 130, 70)])
def test_two_launch_training_forward_equals_the_recomputing_form(shape):
    """Round 4: conv2 evaluated once.  mmi_cem_conv2_fwd's y2 and mmi_cem_fwd_from_y2's t / chansum / y3 / statistics partials are
    BIT-identical to what mmi_cem_fused_fwd (which recomputes conv2 on every tile's halo region) writes for the same BN2 statistics;
    BN2's batch statistics themselves come from differently grouped partial sums, so they agree to rounding."""
    from mmidet_hip import lib
    n, h, w = shape
    d = dev()
    g = torch.Generator().manual_seed(h * 5 + w)
    x = torch.rand(n, h, w, 3, generator=g).to(d)
    w2 = (torch.randn(24, 3, 3, 3, generator=g) * 0.3).to(d)          # OHWI [24][9][3]
    w3 = (torch.randn(3, 3, 3, 24, generator=g) * 0.2).to(d)          # OHWI [3][9][24]
    f, sb = (torch.rand(24, generator=g) + 0.5).to(d), (torch.randn(24, generator=g) * 0.1).to(d)
    gam, bet = (torch.rand(24, generator=g) + 0.5).to(d), (torch.randn(24, generator=g) * 0.2).to(d)
    s = torch.cuda.current_stream().cuda_stream
    rows, nblk = n * h * w, lib.cem_blocks(n, h, w)
    rm, rv, nbt = torch.zeros(24, device=d), torch.ones(24, device=d), torch.zeros((), dtype=torch.int64, device=d)

    def stats(launch, blocks):
        part = torch.zeros((blocks + 64) * 48, device=d)
        launch(part)
        mi = torch.empty(48, device=d)
        lib.bn_finalize(part.data_ptr(), blocks, rows, 24, 1e-5, 0.1, rm.clone().data_ptr(), rv.clone().data_ptr(), nbt.clone().data_ptr(),
                        mi.data_ptr(), s)
        return mi
    y2b = torch.full((n, h, w, 24), float('nan'), device=d)
    mi_a = stats(lambda p: lib.cem_conv2_stats(x.data_ptr(), 3, w2.data_ptr(), p.data_ptr(), n, h, w, s), nblk)
    mi_b = stats(lambda p: lib.cem_conv2_fwd(x.data_ptr(), 3, w2.data_ptr(), y2b.data_ptr(), p.data_ptr(), n, h, w, s),
                 lib.cem_conv2_fwd_blocks(n, h, w))
    close(mi_b, mi_a, what='BN2 mean | invstd', tol=2e-6)

    def outs():
        return (torch.full((n, h, w, 24), float('nan'), device=d), torch.full((n, h, w), float('nan'), device=d),
                torch.full((n, h, w, 3), float('nan'), device=d), torch.zeros((nblk + 64) * 6, device=d))
    y2a = torch.full((n, h, w, 24), float('nan'), device=d)
    ta, ca, y3a, pa = outs()
    lib.cem_fused_fwd(x.data_ptr(), 3, w2.data_ptr(), mi_a.data_ptr(), gam.data_ptr(), bet.data_ptr(), f.data_ptr(), sb.data_ptr(), w3.data_ptr(),
                      y2a.data_ptr(), ta.data_ptr(), ca.data_ptr(), y3a.data_ptr(), pa.data_ptr(), n, h, w, s)
    tb, cb, y3b, pb = outs()
    lib.cem_fwd_from_y2(y2b.data_ptr(), mi_a.data_ptr(), gam.data_ptr(), bet.data_ptr(), f.data_ptr(), sb.data_ptr(), w3.data_ptr(),
                        tb.data_ptr(), cb.data_ptr(), y3b.data_ptr(), pb.data_ptr(), n, h, w, s)
    torch.cuda.synchronize()
    assert torch.equal(y2b, y2a), 'y2'
    assert torch.equal(tb, ta) and torch.equal(cb, ca), 't / chansum'
    assert torch.equal(y3b, y3a) and torch.equal(pb, pa), 'y3 / its statistics partials'


@pytest.mark.parametrize('shape', [(2, 48, 40), (1, 33, 17), (3, 5, 3), (1, 16, 16), (2, 130, 70)])
@pytest.mark.parametrize('frozen', [0, 1])
def test_conv2_weight_gradient_with_the_batchnorm_backward_in_its_loader(shape, frozen):
    """mmi_cem_conv2_wgrad_bn (dy2 made from (dr, y2) on the way into LDS, never in HBM) against the two-step form: the one-call
    BatchNorm backward writing dy2, then the small-channel weight gradient of (x, dy2).  Ragged tiles included: a tile's pixels
    outside the image must contribute nothing although `dz - mean - xhat * mean` is not zero there."""
    from mmidet_hip import lib, ops
    from mmidet_hip.ops import ConvDesc
    n, h, w = shape
    d = dev()
    g = torch.Generator().manual_seed(h * 11 + w + frozen)
    x = torch.rand(n, h, w, 3, generator=g).to(d)
    dr = torch.randn(n, h, w, 24, generator=g).to(d)
    y2 = torch.randn(n, h, w, 24, generator=g).to(d)
    mi = torch.cat([torch.randn(24, generator=g) * 0.2, torch.rand(24, generator=g) + 0.5]).to(d)
    gam, bet = (torch.rand(24, generator=g) + 0.5).to(d), (torch.randn(24, generator=g) * 0.2).to(d)
    s = torch.cuda.current_stream().cuda_stream
    rows = n * h * w
    dy2, dg0, db0 = torch.empty_like(y2), torch.empty(24, device=d), torch.empty(24, device=d)
    ops._bn_act_bwd(y2, 24, dr, 24, None, 0, 24, mi, gam, bet, dy2, (dg0, db0, None, None), rows, 24, ops.ACT_LEAKY, frozen, s)
    wref = torch.empty(24, 3, 3, 3, device=d).contiguous(memory_format=torch.channels_last)
    dw0 = ops._wgrad(dy2, 24, x, 3, wref, ConvDesc(n, h, w, 3, h, w, 24, 3, 3, 1, 1, 3, 24))
    # the fused form: sums only (dy = NULL), then the weight gradient
    gbuf = torch.empty(64, device=d)
    dg1, db1 = gbuf[1:25], gbuf[35:59]          # (views into a gradient bucket: any 4-byte offset)
    nbw = ops.bn_bwd_ws(rows, 24)
    ws = torch.zeros(nbw, dtype=torch.uint8, device=d)
    lib.bn_act_bwd(y2.data_ptr(), 24, dr.data_ptr(), 24, None, 0, 24, mi.data_ptr(), gam.data_ptr(), bet.data_ptr(), ws.data_ptr(), nbw,
                   None, 24, dg1.data_ptr(), db1.data_ptr(), None, None, rows, 24, ops.ACT_LEAKY, frozen, s)
    nb = lib.cem_conv2_wgrad_bn_workspace(n, h, w)
    wsw = torch.empty(nb // 4, device=d)
    dw1 = torch.full((24, 9, 3), float('nan'), device=d)
    lib.cem_conv2_wgrad_bn(dr.data_ptr(), y2.data_ptr(), x.data_ptr(), 3, mi.data_ptr(), gam.data_ptr(), bet.data_ptr(), dg1.data_ptr(),
                           db1.data_ptr(), frozen, dw1.data_ptr(), wsw.data_ptr(), nb, n, h, w, s)
    torch.cuda.synchronize()
    assert torch.equal(dg1, dg0) and torch.equal(db1, db0), 'the sums are the same launches'
    close(dw1.reshape(-1), dw0.permute(0, 2, 3, 1).reshape(-1), what='dw2', tol=2e-5)
