"""GPU parity of the stream-K schedule of the implicit-GEMM convolution (csrc/igemm.hip): small shapes are forced onto
it with a tiny grid (mmi_set_streamk_slots) and checked against torch's CPU conv2d (the same call sites as
test_ops_gpu: models/common.py:114); the BASELINE-size layer is checked through a size-independent property -- the
stream-K result equals the one-tile-per-workgroup result up to fp32 summation order, run to run bit-identical."""
import pytest
import torch
import torch.nn.functional as F

from test_ops_gpu import close, cl, dev, nchw, nhwc

pytestmark = pytest.mark.gpu


@pytest.fixture
def forced_streamk():
    from mmidet_hip import lib
    def force(slots):
        lib.set_streamk_slots(slots)
    yield force
    lib.set_streamk_slots(0)


CASES = [
    # N, H, W, Cin, Cout, k, s, slots
    (2, 20, 24, 64, 64, 3, 1, 5),       # 8 tiles x 18 slabs over 5 workgroups: heads, tails and whole tiles
    (2, 20, 24, 64, 64, 3, 1, 37),      # more workgroups than tiles: tiles split 4-5 ways
    (1, 17, 19, 32, 48, 3, 1, 7),       # ragged M and N tiles
    (4, 40, 40, 128, 128, 3, 1, 24),    # 128x128 tiles
    (2, 16, 16, 128, 256, 1, 1, 9),     # 1x1, two N tiles
    (2, 32, 32, 64, 128, 3, 2, 11),     # stride 2: forward stream-K, dgrad stays on the parity schedule
    (3, 8, 8, 256, 128, 1, 1, 16),
]


@pytest.mark.parametrize('case', CASES)
def test_streamk_conv_matches_conv2d(case, forced_streamk):
    from mmidet_hip import lib, ops
    N, H, W, Cin, Cout, k, s, slots = case
    forced_streamk(slots)
    d0 = ops._desc((N, H, W, Cin), Cout, k, s, Cin, Cout)
    assert lib.conv_fwd_workspace(d0) > 0, 'the case is meant to take the stream-K schedule'
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
    for rep in range(2):          # twice: the second launch runs on the counters the first one left behind
        xg.grad = wg.grad = bg.grad = None
        yg = ops.conv_bias(xg, wg, bg, s)
        yg.backward(nhwc(gy).to(d))
        torch.cuda.synchronize()
        close(nchw(yg), yr, what='y (launch %d)' % rep)
        close(nchw(xg.grad), xr.grad, what='dx (launch %d)' % rep)
        close(wg.grad, wr.grad, what='dw')


def test_streamk_bn_statistics(forced_streamk):
    """The BatchNorm partial sums of the epilogue under stream-K: conv+BN(train)+SiLU against torch."""
    from mmidet_hip import ops
    forced_streamk(13)
    N, H, W, Cin, Cout = 4, 24, 24, 64, 128
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    gamma, beta = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.2
    rm, rv = torch.zeros(Cout), torch.ones(Cout)
    yr = F.silu(F.batch_norm(F.conv2d(x, w, None, 1, 1), rm, rv, gamma, beta, True, 0.03, 1e-3))
    d = dev()
    rmg, rvg = torch.zeros(Cout, device=d), torch.ones(Cout, device=d)
    yg = ops.conv_bn_act(nhwc(x).to(d), cl(w).to(d), gamma.to(d), beta.to(d), rmg, rvg, None, 1, ops.ACT_SILU)
    close(nchw(yg), yr, what='y')
    close(rmg, rm, what='running_mean')
    close(rvg, rv, what='running_var')


@pytest.mark.parametrize('shape', [(16, 80, 80, 128, 128), (16, 20, 20, 512, 512)])
def test_streamk_full_size_equals_data_parallel(shape):
    """BASELINE-size bottleneck convs (800 and 200 tiles on 768 slots): stream-K vs one workgroup per tile."""
    from mmidet_hip import lib, ops
    N, H, W, Cin, Cout = shape
    d = dev()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, H, W, Cin, generator=g).to(d)
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) / (Cin * 9) ** 0.5).to(d)
    dy = torch.randn(N, H, W, Cout, generator=g).to(d)
    desc = ops._desc((N, H, W, Cin), Cout, 3, 1, Cin, Cout)
    outs = {}
    for mode in (0, -1, 0):
        lib.set_streamk_slots(mode)
        try:
            if mode == 0:
                sk_fwd = lib.conv_fwd_workspace(desc)
                assert sk_fwd > 0 and lib.conv_dgrad_workspace(desc) > 0
            else:   # no stream-K slots: the dgrad needs nothing, the forward only its counter block + statistics partials
                assert lib.conv_dgrad_workspace(desc) == 0 and 0 < lib.conv_fwd_workspace(desc) < 1 << 20 < sk_fwd
            nrb = lib.conv_fwd_row_blocks(desc)
            part = torch.zeros(nrb * 2 * Cout, device=d)
            y = torch.empty(N, H, W, Cout, device=d)
            dx = torch.empty(N, H, W, Cin, device=d)
            ops.conv_fwd(x, w, None, y, part, desc, st)
            ops.conv_dgrad(dy, w, dx, desc, st)
            torch.cuda.synchronize()
            stats = part.view(nrb, 2, Cout).double().sum(0)
        finally:
            lib.set_streamk_slots(0)
        outs.setdefault(mode, []).append((y, dx, stats))
    (y1, dx1, s1), (y3, dx3, s3) = outs[0]
    y2, dx2, s2 = outs[-1][0]
    assert torch.equal(y1, y3) and torch.equal(dx1, dx3), 'stream-K must be run-to-run deterministic'
    close(y1, y2, tol=1e-5, what='y stream-K vs data-parallel')
    close(dx1, dx2, tol=1e-5, what='dx stream-K vs data-parallel')
    close(s1, s2, tol=1e-5, what='BN partial sums')
