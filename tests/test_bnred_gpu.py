"""mmi_conv_dgrad_bnred / mmi_conv_dgrad2_bnred (round 4): the input gradient of a convolution with the BatchNorm backward REDUCTION
of the layer below in its epilogue, against the two-step form (mmi_conv_dgrad, then the one-call BatchNorm backward on its output):
dx bit for bit (the GEMM is the same), dgamma / dbeta / dy to summation-order rounding."""
import pytest
import torch

from test_ops_gpu import close, dev

pytestmark = pytest.mark.gpu

# (N, H, W, Cin, Cout, k): ragged tiles, a 1x1, a 64-channel layer, a full-size stream-K shape
SHAPES = [(2, 20, 20, 128, 128, 3), (1, 13, 9, 64, 96, 3), (2, 40, 40, 128, 64, 1), (1, 16, 16, 256, 256, 1), (16, 80, 80, 128, 128, 3)]


def _bn(g, c, d):
    mi = torch.cat([torch.randn(c, generator=g) * 0.2, torch.rand(c, generator=g) + 0.5]).to(d)
    return mi, (torch.rand(c, generator=g) + 0.5).to(d), (torch.randn(c, generator=g) * 0.2).to(d)


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('act', ['silu', 'leaky'])
@pytest.mark.parametrize('skip', [False, True])
def test_dgrad_with_the_batchnorm_reduction_in_its_epilogue(shape, act, skip):
    from mmidet_hip import lib, ops
    from mmidet_hip.ops import ConvDesc
    n, h, w, cin, cout, k = shape
    if skip and k != 1:
        pytest.skip('the epilogue accumulation exists for 1x1 layers')
    d = dev()
    g = torch.Generator().manual_seed(h * 13 + cin + k)
    dy = torch.randn(n, h, w, cout, generator=g).to(d)
    wt = (torch.randn(cout, k, k, cin, generator=g) * 0.05).to(d)
    y = torch.randn(n, h, w, cin, generator=g).to(d)              # the layer below: its raw conv output (BatchNorm's input)
    sk = torch.randn(n, h, w, cin, generator=g).to(d) if skip else None
    mi, gam, bet = _bn(g, cin, d)
    a = ops.ACT_SILU if act == 'silu' else ops.ACT_LEAKY
    s = torch.cuda.current_stream().cuda_stream
    rows = n * h * w
    dd = ConvDesc(n, h, w, cin, h, w, cout, k, k, 1, k // 2, cin, cout)
    # two-step reference
    dx0 = torch.empty(n, h, w, cin, device=d)
    ops.conv_dgrad(dy, wt, dx0, dd, s)
    if skip:
        dx0 += sk
    dyb0, dg0, db0 = torch.empty_like(y), torch.empty(cin, device=d), torch.empty(cin, device=d)
    ops._bn_act_bwd(y, cin, dx0, cin, None, 0, cin, mi, gam, bet, dyb0, (dg0, db0, None, None), rows, cin, a, 0, s)
    # fused
    nparts = lib.conv_dgrad_row_blocks_n(dd, 1)
    part = torch.full((nparts * 2 * cin,), float('nan'), device=d)
    hook = lib.BnReduceHook(y.data_ptr(), cin, mi.data_ptr(), cin, gam.data_ptr(), bet.data_ptr(), a, part.data_ptr())
    dx1 = torch.full((n, h, w, cin), float('nan'), device=d)
    ops.conv_dgrad_bnred(dy, wt, dx1, dd, hook, s, skip=sk, lds=cin)
    dyb1, dg1, db1 = torch.empty_like(y), torch.empty(cin, device=d), torch.empty(cin, device=d)
    lib.bn_act_bwd_apply(y.data_ptr(), cin, dx1.data_ptr(), cin, mi.data_ptr(), gam.data_ptr(), bet.data_ptr(), part.data_ptr(), nparts,
                         dyb1.data_ptr(), cin, dg1.data_ptr(), db1.data_ptr(), rows, cin, a, 0, s)
    torch.cuda.synchronize()
    if skip:
        close(dx1, dx0, what='dx', tol=1e-6)          # (the reference added the skip in a separate pass: same sum, one rounding apart)
    else:
        assert torch.equal(dx1, dx0), 'dx'
    close(dg1, dg0, what='dgamma', tol=2e-5)
    close(db1, db0, what='dbeta', tol=2e-5)
    close(dyb1, dyb0, what='dy of the BatchNorm', tol=2e-5)


@pytest.mark.parametrize('shape', [(2, 20, 20, 128, 128, 3), (2, 40, 40, 64, 64, 1), (16, 80, 80, 128, 128, 3)])
@pytest.mark.parametrize('skip', [False, True])
def test_twin_dgrad_with_both_lanes_reductions(shape, skip, monkeypatch):
    """Twin launch: both lanes' gradients into one (N,H,W,2,Cin) buffer, each lane's partial sums from its own columns, folded and
    applied by mmi_bn_act_bwd_apply_map -- against mmi_conv_dgrad2 followed by mmi_bn_act_bwd_map."""
    from mmidet_hip import lib, ops, twin_ops
    from mmidet_hip.ops import ConvDesc
    n, h, w, cin, cout, k = shape
    if skip and k != 1:
        pytest.skip('the epilogue accumulation exists for 1x1 layers')
    monkeypatch.setattr(ops, 'BNRED_K1', True)          # (the helper declines 1x1 layers unless asked: their epilogue cannot hide the work)
    d = dev()
    g = torch.Generator().manual_seed(h * 17 + cin + k)
    dy = torch.randn(n, h, w, 2, cout, generator=g).to(d)
    wa, wb = [(torch.randn(cout, k, k, cin, generator=g) * 0.05).to(d) for _ in range(2)]
    y = torch.randn(n, h, w, 2, cin, generator=g).to(d)
    sk = torch.randn(n, h, w, 2, cin, generator=g).to(d) if skip else None
    mi = torch.cat([torch.randn(2 * cin, generator=g) * 0.2, torch.rand(2 * cin, generator=g) + 0.5]).to(d)
    gs = [(torch.rand(cin, generator=g) + 0.5).to(d) for _ in range(2)]
    bs = [(torch.randn(cin, generator=g) * 0.2).to(d) for _ in range(2)]
    s = torch.cuda.current_stream().cuda_stream
    rows = n * h * w
    dconv = ops._desc((n, h, w, cin), cout, k, 1, 2 * cin, 2 * cout)

    def bn_map(dgs, dbs):
        return twin_ops._bn_map(gs, bs, cin, cin, cin, cin, 0, dgs, dbs)
    # two-step reference
    dx0 = torch.empty(n, h, w, 2, cin, device=d)
    twin_ops.dgrad2(dy, cout, wa, wb, dx0, dconv, sk, s)
    dg0, db0 = [torch.empty(cin, device=d) for _ in range(2)], [torch.empty(cin, device=d) for _ in range(2)]
    dyb0 = torch.empty_like(y)
    nbw = ops.bn_bwd_ws(rows, 2 * cin)
    ws = torch.zeros(nbw, dtype=torch.uint8, device=d)
    lib.bn_act_bwd_map(y.data_ptr(), 2 * cin, dx0.data_ptr(), 2 * cin, None, 0, mi.data_ptr(), bn_map(dg0, db0), ws.data_ptr(), nbw,
                       dyb0.data_ptr(), 2 * cin, rows, 2 * cin, ops.ACT_SILU, 0, s)
    # fused
    dx1 = torch.full((n, h, w, 2, cin), float('nan'), device=d)
    parts = twin_ops.dgrad2_bnred(dy, cout, wa, wb, dx1, dconv, y, mi, gs, bs, ops.ACT_SILU, sk, s)
    dg1, db1 = [torch.empty(cin, device=d) for _ in range(2)], [torch.empty(cin, device=d) for _ in range(2)]
    dyb1 = torch.empty_like(y)
    twin_ops.bn_apply_map(y, dx1, mi, bn_map(dg1, db1), parts, dyb1, rows, 2 * cin, ops.ACT_SILU, 0, s)
    torch.cuda.synchronize()
    if skip:
        close(dx1, dx0, what='dx', tol=1e-6)
    else:
        assert torch.equal(dx1, dx0), 'dx'
    for i in range(2):
        close(dg1[i], dg0[i], what='dgamma lane %d' % i, tol=2e-5)
        close(db1[i], db0[i], what='dbeta lane %d' % i, tol=2e-5)
    close(dyb1, dyb0, what='dy of the BatchNorm', tol=2e-5)


@pytest.mark.parametrize('kind', ['fourier', 'add'])
def test_training_step_takes_the_fused_reduction_and_matches_the_separate_pass(kind, monkeypatch):
    """Whole model (tiny two-stream graphs, twin and lane layers, Bottleneck shortcuts, C3 concatenations): with the feature on, the
    BatchNorm backward of every layer whose output has exactly one (hook-aware) reader takes its sums from that reader's dgrad
    epilogue -- and every parameter gradient equals the run with the feature off to summation-order rounding.  No epilogue's work
    is thrown away ('rejected' = a consumer computed sums for a gradient that turned out not to be the whole one)."""
    import copy
    from mmidet_hip import ops
    from test_step_gpu import batch, make
    grads = []
    for on in (False, True):
        monkeypatch.setattr(ops, 'BNRED', on)
        ops.BNRED_COUNT.update(taken=0, rejected=0)
        torch.manual_seed(0)
        m, ts, cfg = make(kind)
        imgs, tg = batch(cfg, 77)
        ts._body(imgs, tg)                 # forward, loss, backward (no optimizer step: the gradients are what is compared)
        ops.join_pending()
        torch.cuda.synchronize()
        grads.append({n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
        counts = copy.copy(ops.BNRED_COUNT)
    assert counts['taken'] >= 8 and counts['rejected'] == 0, counts
    assert set(grads[0]) == set(grads[1]) and len(grads[0]) > 50
    for n in grads[0]:
        if float(grads[0][n].abs().max()) < 1e-7:      # (key_proj.bias: its gradient is zero mathematically, rounding noise in fp32)
            continue
        close(grads[1][n], grads[0][n], what=n, tol=5e-4)
    print('BatchNorm reductions taken from a dgrad epilogue:', counts)
