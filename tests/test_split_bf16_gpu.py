"""GPU parity of the OPT-IN split-bf16 GEMM arithmetics (mmi_set_gemm_precision, csrc/igemm.hip PREC): mode 1 = two bf16
terms per operand, three products (hi*hi + hi*lo + lo*hi); mode 2 = three terms, the six products of total order <= 2
(fp32-level accuracy, also at full depth: test_model_gpu.py); mode 3 = three terms, all nine products (every fp32 product exact); all on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Same checks and
the same tolerances as the default fp32-MFMA path for every op and for the model's forward and loss; the whole-step
parameter gradients get the looser bound they need (see test_train_step_split_bf16_matches_oracle)."""
import pytest
import torch
import torch.nn.functional as F

from test_ops_gpu import CONV_CASES, close, cl, dev, nchw, nhwc, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[1, 2, 3], ids=['bf16x3', 'bf16x6', 'bf16x9'])
def split_bf16(request):
    from mmidet_hip import lib
    lib.set_gemm_precision(request.param)
    lib.mode = request.param
    yield lib
    lib.set_gemm_precision(0)
    lib.set_streamk_slots(0)


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd_bwd_split_bf16(case, split_bf16):
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
    xg, wg, bg = nhwc(x).to(d).requires_grad_(), cl(w).to(d).requires_grad_(), b.to(d).requires_grad_()
    yg = ops.conv_bias(xg, wg, bg, s)
    yg.backward(nhwc(gy).to(d))
    close(nchw(yg), yr, what='y')
    close(nchw(xg.grad), xr.grad, what='dx')
    close(wg.grad, wr.grad, what='dw')
    close(bg.grad, br.grad, what='db')


@pytest.mark.parametrize('slots', [5, 24])
def test_streamk_split_bf16(slots, split_bf16):
    """The stream-K schedule (partial tiles folded by the last contributor) with the split-bf16 inner product."""
    from mmidet_hip import ops
    split_bf16.set_streamk_slots(slots)
    g = torch.Generator().manual_seed(slots)
    x = torch.randn(4, 128, 40, 40, generator=g)
    w = torch.randn(128, 128, 3, 3, generator=g) / (128 * 9) ** 0.5
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    yr = F.conv2d(xr, wr, None, 1, 1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    d = dev()
    xg, wg = nhwc(x).to(d).requires_grad_(), cl(w).to(d).requires_grad_()
    yg = ops.conv_bias(xg, wg, None, 1)
    yg.backward(nhwc(gy).to(d))
    close(nchw(yg), yr, what='y')
    close(nchw(xg.grad), xr.grad, what='dx')
    close(wg.grad, wr.grad, what='dw')


@pytest.mark.parametrize('kind', ['add', 'fourier'])
def test_train_step_split_bf16_matches_oracle(kind, split_bf16):
    from oracle import portable_init
    from oracle.ref_loss import ComputeLoss as OLoss
    from test_model_gpu import build_pair
    from utils.loss import ComputeLoss

    def run(seed):
        m, o, cfg = build_pair(kind, 128)
        imgs, targets = portable_init.synth_batch(2, 128, cfg['nc'], per_image=4, seed=seed)
        x = imgs.float() / 255
        m.train()
        o.train()
        po, co = o(x[:, :3], x[:, 3:])
        lo, io = OLoss(o)(po, targets, co.reshape(-1))
        lo.backward()
        xd = x.to(dev())
        pg, cg = m(xd[:, :3], xd[:, 3:])
        lg, ig = ComputeLoss(m)(pg, targets.to(dev()), cg.reshape(-1))
        lg.backward()
        for i in range(3):
            close(pg[i], po[i], what='pred%d' % i)
        close(lg, lo, what='loss', tol=1e-4)
        close(ig, io, what='items', tol=1e-4)
        og = dict(o.named_parameters())
        return sorted(rel_err(p.grad, og[n].grad) for n, p in m.named_parameters()
                      if og[n].grad is not None and p.grad is not None and float(og[n].grad.norm()) > 1e-5)
    # Gradients are where the split arithmetic shows: a GEMM result is off by ~5e-6, predictions by ~3e-5, but ~150 layers of
    # backward (BatchNorm's mean-subtraction cancels leading digits) amplify that to ~1e-3 per parameter gradient (fp32 MFMA
    # path: 5e-5, tests/diag/b3_diag.py).  That is why the mode is opt-in and not the benchmarked default.
    errs = run(2)
    if split_bf16.mode >= 2:      # three-term splits: the budget of the fp32 path (test_every_parameter_gradient_vs_oracle)
        assert errs[-1] < 2e-3, errs[-1]
        return
    # bf16x3 (inexact products): its rounding level is median < 5e-3 / worst < 2e-2 -- unless a discrete decision of the backward (a
    # LeakyReLU / max-pool tie within the forward's 3e-5) falls the other way, which moves every gradient upstream of it by ~1e-2
    # (test_model_gpu.py::test_yolov5l_640_train_step_matches_oracle describes the same for the full-size graphs).  Which seed that
    # hits depends on the last bits of every kernel upstream (round 4's hardware-exp SiLU moved it from no seed here to seed 2), so:
    # every seed stays under the event level, and at least one of two is at the rounding level.
    runs = [errs, run(3)]
    for e in runs:
        assert e[len(e) // 2] < 3e-2 and e[-1] < 1.5e-1, (e[len(e) // 2], e[-1])
    assert any(e[len(e) // 2] < 5e-3 and e[-1] < 2e-2 for e in runs), [(e[len(e) // 2], e[-1]) for e in runs]


@pytest.mark.parametrize('mode,tol', [(1, 2e-5), (2, 5e-6), (3, 5e-6)], ids=['bf16x3', 'bf16x6', 'bf16x9'])
@pytest.mark.parametrize('shape', [(16, 80, 80, 128, 128, 3, 1), (16, 160, 160, 128, 256, 3, 2), (16, 40, 40, 512, 256, 1, 1)])
def test_full_size_layers_split_bf16_vs_fp32_mfma(shape, mode, tol):
    """BASELINE-size layers: forward, dgrad and wgrad of the split form against the exact fp32-MFMA kernels.  A product
    of the two-term form carries <= 2^-16 relative error and the errors of a K-long sum average out: 2e-5 is a loose bound
    (measured 4.5e-6); the three-term form differs from the fp32 kernels by what two fp32 summation orders differ by
    (measured 3e-7..3.6e-6 on wgrad's 100 K-long sums)."""
    from mmidet_hip import lib, ops
    N, H, W, Cin, Cout, k, s = shape
    d = dev()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, H, W, Cin, generator=g).to(d)
    w = (torch.randn(Cout, k, k, Cin, generator=g) / (Cin * k * k) ** 0.5).to(d)
    desc = ops._desc((N, H, W, Cin), Cout, k, s, Cin, Cout)
    dy = torch.randn(N, desc.Ho, desc.Wo, Cout, generator=g).to(d)
    res = {}
    for md in (0, mode):
        lib.set_gemm_precision(md)
        try:
            y = torch.empty(N, desc.Ho, desc.Wo, Cout, device=d)
            dx = torch.empty_like(x)
            dw = torch.empty_like(w)
            nb = lib.conv_wgrad_workspace(desc)
            ws = torch.zeros(max(nb // 4, 1), device=d)     # (arrival counters at its head: zero-filled once)
            ops.conv_fwd(x, w, None, y, None, desc, st)
            ops.conv_dgrad(dy, w, dx, desc, st)
            lib.conv_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), nb, desc, st)
            torch.cuda.synchronize()
            res[md] = (y, dx, dw)
        finally:
            lib.set_gemm_precision(0)
    for name, a, b in zip(('y', 'dx', 'dw'), res[mode], res[0]):
        close(a, b, tol=tol, what=name + ' split-bf16 vs fp32 MFMA')


@pytest.mark.parametrize('mode,shape', [(1, (4, 128, 128, 12, 64, 3, 1)), (3, (4, 40, 40, 128, 128, 3, 1)),
                                        (1, (16, 40, 40, 512, 256, 1, 1)), (2, (16, 80, 80, 128, 128, 3, 1))],
                         ids=['bf16x3-focus', 'bf16x9-c128', 'bf16x3-1x1', 'bf16x6-c128'])
def test_split_bf16_kernels_repeat_bit_for_bit_wherever_the_buffers_lie(mode, shape):
    """the three split-bf16 forward results that were sparsely wrong ONCE, in a process whose
    allocator layout an earlier failure had shifted.  The kernels take no decision that depends on an address or on timing, so
    the same operands must give the same bits wherever inputs, outputs and workspaces lie and whatever else runs on the chip:
    24 rounds with every buffer at a new address (a growing spacer moves every later block; operands are re-cloned), every other
    round next to a GEMM on a second stream, forward + dgrad + wgrad compared bit for bit with the first round -- and the
    forward with the exact fp32-MFMA kernels' result.  Poison mode (conftest) adds: no element left unwritten, no store outside
    an output."""
    from mmidet_hip import alloc, lib, ops
    N, H, W, Cin, Cout, k, s = shape
    d = dev()
    g = torch.Generator().manual_seed(11)
    x0 = torch.randn(N, H, W, Cin, generator=g).to(d)
    w0 = (torch.randn(Cout, k, k, Cin, generator=g) / (Cin * k * k) ** 0.5).to(d)
    desc = ops._desc((N, H, W, Cin), Cout, k, s, Cin, Cout)
    dy0 = torch.randn(N, desc.Ho, desc.Wo, Cout, generator=g).to(d)
    st = torch.cuda.current_stream().cuda_stream
    side = torch.cuda.Stream()
    a = torch.randn(2048, 2048, device=d)

    def run(x, w, dy):
        y = alloc.empty((N, desc.Ho, desc.Wo, Cout), dtype=torch.float32, device=d)
        dx, dw = alloc.empty_like(x), alloc.empty_like(w)
        nb = lib.conv_wgrad_workspace(desc)
        ws = torch.zeros(max(nb // 4, 1), device=d)
        ops.conv_fwd(x, w, None, y, None, desc, st)
        ops.conv_dgrad(dy, w, dx, desc, st)
        lib.conv_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), nb, desc, st)
        return y, dx, dw

    y_fp32 = run(x0, w0, dy0)[0]
    torch.cuda.synchronize()
    lib.set_gemm_precision(mode)
    try:
        first, spacers = None, []
        for it in range(24):
            spacers.append(torch.empty(((it * 37) % 101 + 1) * 256 * 1024 + 512 * it, dtype=torch.uint8, device=d))
            if it % 3 == 2:
                spacers.pop(0)                                   # holes as well as growth
            x, w, dy = x0.clone(), w0.clone(), dy0.clone()
            if it % 2:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(4):
                        a @ a                                    # something else on the chip while the kernels run
            out = run(x, w, dy)
            torch.cuda.synchronize()
            assert all(bool(torch.isfinite(t).all()) for t in out), 'round %d: an element was left unwritten' % it
            if first is None:
                first = [t.clone() for t in out]
            else:
                for name, t, f in zip(('y', 'dx', 'dw'), out, first):
                    if not torch.equal(t, f):
                        bad = torch.nonzero(t != f)
                        raise AssertionError('round %d: %s differs from round 0 in %d elements, first at %s (%.6g vs %.6g)'
                                             % (it, name, bad.shape[0], bad[0].tolist(), float(t[tuple(bad[0])]), float(f[tuple(bad[0])])))
    finally:
        lib.set_gemm_precision(0)
    close(first[0], y_fp32, tol=2e-5 if mode == 1 else 5e-6, what='y split-bf16 vs fp32 MFMA')
