"""GPU: the round-2 launch fusions against the forms they replace (which stay in the C ABI) and against torch:
BatchNorm statistics finished inside the conv launch (mmi_conv_bn_fwd vs mmi_conv_fwd + mmi_bn_finalize), the one-call
BatchNorm backward with split operands (mmi_bn_act_bwd), the split-K fold inside the wgrad launch, C3 with cv1 | cv2 as one
GEMM and no concat copy against the two-convolution form, the Bottleneck shortcut gradient in the dgrad epilogue, and
run-to-run bit-identity of a whole training step (no float atomics left on the path)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import tiny_cfg
from test_ops_gpu import cl, close, dev, nchw, nhwc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('case', [(2, 20, 24, 64, 64, 3, 1), (4, 40, 40, 64, 128, 3, 1), (2, 21, 23, 32, 64, 3, 2), (16, 80, 80, 64, 128, 1, 1),
                                  (8, 160, 160, 32, 64, 3, 1), (3, 8, 8, 256, 136, 1, 1)])
def test_statistics_folded_in_the_conv_launch(case):
    """mean / invstd / running stats / num_batches_tracked written by the last-arriving workgroups = what the separate fold
    kernel computes from the same partial rows (same fp64 fold; the final 1/sqrt is taken in fp32 here)."""
    from mmidet_hip import lib, ops
    N, H, W, Ci, Co, k, s = case
    d = dev()
    g = torch.Generator().manual_seed(sum(case))
    x = (torch.randn(N, H, W, Ci, generator=g) + 0.3).to(d)
    w = (torch.randn(Co, k, k, Ci, generator=g) / (k * k * Ci) ** 0.5).to(d)
    desc = ops._desc((N, H, W, Ci), Co, k, s, Ci, Co)
    rows = N * desc.Ho * desc.Wo
    st = torch.cuda.current_stream().cuda_stream
    nrb = lib.conv_fwd_row_blocks(desc)
    res = []
    for fused in (False, True):
        y = torch.empty(N, desc.Ho, desc.Wo, Co, device=d)
        part = torch.empty((nrb + 64) * 2 * Co, device=d)
        rm, rv = torch.full((Co,), 0.25, device=d), torch.full((Co,), 2.0, device=d)
        nbt = torch.zeros(2, dtype=torch.long, device=d)
        mi = torch.empty(2 * Co, device=d)
        nb = lib.conv_fwd_workspace(desc)
        ws = torch.zeros(nb, dtype=torch.uint8, device=d)
        for rep in range(2):          # twice: the counters must come back to zero
            if fused:
                bn = lib.BnStats(1e-3, 0.03, rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), nbt[1:].data_ptr(), mi.data_ptr())
                lib.conv_bn_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), part.data_ptr(), bn, ws.data_ptr(), nb, desc, st)
            else:
                lib.conv_fwd(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), part.data_ptr(), ws.data_ptr(), nb, desc, st)
                lib.bn_finalize(part.data_ptr(), nrb, rows, Co, 1e-3, 0.03, rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), mi.data_ptr(), st)
        torch.cuda.synchronize()
        if fused:
            head = ws[:(256 + 64) * 1024].view(torch.int32)
            assert int(head.abs().max()) == 0, 'arrival counters must be zero after the launch'
        res.append((y, mi.clone(), rm, rv, nbt))
    (y0, mi0, rm0, rv0, n0), (y1, mi1, rm1, rv1, n1) = res
    assert torch.equal(y0, y1)
    # (the in-launch fold rounds its per-group sums to fp32 on the way; the separate kernel folds short lists in one go)
    close(mi1[:Co], mi0[:Co], tol=1e-6, what='batch mean')
    close(mi1[Co:], mi0[Co:], tol=1e-6, what='invstd')
    close(rm1, rm0, tol=1e-6, what='running_mean')
    close(rv1, rv0, tol=1e-6, what='running_var')
    assert n0.tolist() == [2, 0] and n1.tolist() == [2, 2]
    yr = y0.double().cpu().reshape(-1, Co)
    close(mi1[:Co], yr.mean(0), tol=1e-5, what='mean vs fp64')
    close(mi1[Co:], 1.0 / torch.sqrt(yr.var(0, unbiased=False) + 1e-3), tol=1e-5, what='invstd vs fp64')


@pytest.mark.parametrize('shape', [(2 * 20 * 24, 64, 64), (16 * 80 * 80, 128, 64), (4 * 40 * 40, 256, 128), (7 * 13, 48, 24), (1 << 18, 64, 32)])
@pytest.mark.parametrize('act', [1, 2])
def test_bn_backward_one_call_with_split_operands(shape, act):
    """mmi_bn_act_bwd (reduce with in-launch dgamma/dbeta + apply), dout and the parameter gradients split over two tensors at
    channel `split`, against torch autograd of act(batch_norm(y))."""
    from mmidet_hip import lib
    rows, C, split = shape
    d = dev()
    g = torch.Generator().manual_seed(rows + C + act)
    y = torch.randn(rows, C, generator=g) * 1.5 + 0.2
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    dout = torch.randn(rows, C, generator=g)
    yr, gr, br = y.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    z = F.batch_norm(yr.t().reshape(1, C, rows), None, None, gr, br, True, 0.03, 1e-3).reshape(C, rows).t()
    (F.silu(z) if act == 1 else F.leaky_relu(z, 0.1)).backward(dout)
    yd = y.to(d)
    mean, var = yd.double().mean(0), yd.double().var(0, unbiased=False)
    mi = torch.cat([mean, 1.0 / torch.sqrt(var + 1e-3)]).float()
    gd, bd = gamma.to(d), beta.to(d)
    # the two halves of dout live in differently strided tensors
    wide0 = torch.zeros(rows, split + 8, device=d)
    wide0[:, :split] = dout[:, :split].to(d)
    wide1 = torch.zeros(rows, 2 * C, device=d)
    wide1[:, C:2 * C - split] = dout[:, split:].to(d)
    d0, d1 = wide0[:, :split], wide1[:, C:2 * C - split]
    st = torch.cuda.current_stream().cuda_stream
    nb = lib.bn_act_bwd_workspace(rows, C)
    ws = torch.zeros(nb, dtype=torch.uint8, device=d)
    for rep in range(2):
        dy = torch.empty(rows, C, device=d)
        dg0, db0 = torch.empty(split, device=d), torch.empty(split, device=d)
        dg1, db1 = torch.empty(C - split, device=d), torch.empty(C - split, device=d)
        lib.bn_act_bwd(yd.data_ptr(), C, d0.data_ptr(), d0.stride(0), d1.data_ptr(), d1.stride(0), split, mi.data_ptr(), gd.data_ptr(),
                       bd.data_ptr(), ws.data_ptr(), nb, dy.data_ptr(), C, dg0.data_ptr(), db0.data_ptr(), dg1.data_ptr(), db1.data_ptr(),
                       rows, C, act, 0, st)
        torch.cuda.synchronize()
        assert int(ws[:64 * 1024].view(torch.int32).abs().max()) == 0
        if act == 2 and rows > 100000:
            # LeakyReLU has a kink at 0: over 16.7 M elements a few sit within rounding distance of it and take the other
            # slope under this test's fp64 batch statistics; each is off by 0.9 |dout| (measured 2.9e-4 of the L2 norm)
            from test_ops_gpu import rel_err
            assert rel_err(dy, yr.grad) < 1e-3
            assert int(((dy.cpu() - yr.grad).abs() > 1e-3).sum()) < 20
        else:
            close(dy, yr.grad, tol=2e-4, what='dy')
        ptol = 1e-3 if (act == 2 and rows > 100000) else 2e-4       # (the same kink flips land in the column sums)
        close(torch.cat([dg0, dg1]), gr.grad, tol=ptol, what='dgamma')
        close(torch.cat([db0, db1]), br.grad, tol=ptol, what='dbeta')


@pytest.mark.parametrize('case', [(2, 20, 24, 64, 64, 3, 1), (16, 20, 20, 512, 512, 3, 1), (2, 16, 16, 128, 256, 1, 1), (1, 33, 17, 96, 64, 3, 1),
                                  (64, 1, 1, 256, 1024, 1, 1)])
def test_wgrad_fold_inside_the_launch(case, monkeypatch):
    """dw (and the bias gradient) folded by the last-arriving workgroup of each tile = torch's conv weight gradient, twice
    in a row bit-identical, counters back at zero."""
    from mmidet_hip import lib, ops
    N, H, W, Ci, Co, k, s = case
    d = dev()
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    desc = ops._desc((N, H, W, Ci), Co, k, s, Ci, Co)
    dy = torch.randn(N, Co, desc.Ho, desc.Wo, generator=g)
    wr = w.clone().requires_grad_()
    F.conv2d(x, wr, None, s, k // 2).backward(dy)
    xg, dyg = nhwc(x).to(d), nhwc(dy).to(d)
    st = torch.cuda.current_stream().cuda_stream
    nb = lib.conv_wgrad_workspace(desc)
    ws = torch.zeros(max(nb, 16), dtype=torch.uint8, device=d)
    outs = []
    for rep in range(2):
        dw = torch.empty(Co, k, k, Ci, device=d)
        db = torch.empty(Co, device=d)
        lib.conv_wgrad(dyg.data_ptr(), xg.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, desc, st)
        torch.cuda.synchronize()
        if nb:
            assert int(ws[:16384].view(torch.int32).abs().max()) == 0
        outs.append((dw, db))
    close(outs[0][0].permute(0, 3, 1, 2), wr.grad, tol=2e-4, what='dw')
    close(outs[0][1], dy.sum((0, 2, 3)), tol=2e-4, what='dbias')
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize('case', [(2, 20, 24, 64, 64, 3, 1), (2, 21, 23, 32, 64, 3, 2), (16, 40, 40, 256, 256, 3, 1), (1, 33, 17, 96, 64, 3, 1),
                                  (3, 9, 50, 64, 128, 3, 2)])
def test_wgrad_precomputed_pixel_table_is_bit_identical(case):
    """mmi_conv_wgrad_tab with the layer's precomputed pixel table against the in-kernel table builder (table = NULL): the
    same entries, so dw must be equal bit for bit -- borders, ragged last slab, stride 2 and split-K included."""
    from mmidet_hip import lib, ops
    N, H, W, Ci, Co, k, s = case
    d = dev()
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, H, W, Ci, generator=g).to(d)
    desc = ops._desc((N, H, W, Ci), Co, k, s, Ci, Co)
    dy = torch.randn(N, desc.Ho, desc.Wo, Co, generator=g).to(d)
    st = torch.cuda.current_stream().cuda_stream
    nb = lib.conv_wgrad_workspace(desc)
    ws = torch.zeros(max(nb, 16), dtype=torch.uint8, device=d)
    tb = lib.conv_wgrad_table_bytes(desc)
    assert tb > 0
    tab = torch.empty(tb, dtype=torch.uint8, device=d)
    lib.conv_wgrad_table_build(tab.data_ptr(), desc, st)
    outs = []
    for t in (None, tab.data_ptr()):
        dw = torch.empty(Co, k, k, Ci, device=d)
        lib.conv_wgrad_tab(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), nb, t, desc, st)
        torch.cuda.synchronize()
        outs.append(dw)
    assert torch.equal(outs[0], outs[1])
    # 1x1 stride-1 layers take no table
    assert lib.conv_wgrad_table_bytes(ops._desc((N, H, W, Ci), Co, 1, 1, Ci, Co)) == 0


def _c3(c1, c2, n, shortcut, seed):
    from models.common import C3
    torch.manual_seed(seed)
    m = C3(c1, c2, n, shortcut)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eps, mod.momentum = 1e-3, 0.03
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.normal_(0, 0.2)
    return m


@pytest.mark.parametrize('cfg', [(64, 64, 1, True), (128, 128, 3, True), (256, 128, 2, False)])
def test_c3_merged_concat_free_matches_two_conv_form(cfg):
    """C3 with cv1 | cv2 packed (one GEMM, one BN pass, outputs written into the concat buffer, Bottleneck shortcut gradients in
    the dgrad epilogue) against the same module unpacked (two convolutions, torch-style concat copy): outputs, input gradient,
    every parameter gradient and every BatchNorm buffer."""
    import copy
    from mmidet_hip import ops
    c1, c2, n, sc = cfg
    d = dev()
    a = _c3(c1, c2, n, sc, 5).to(d).train()
    b = copy.deepcopy(a)
    assert ops.pack_pair(b) == 1 and b.packed() and not a.packed()
    for k, v in a.state_dict().items():
        assert torch.equal(v, b.state_dict()[k]), k
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 24, 20, c1, generator=g).to(d)
    gy = torch.randn(4, 24, 20, c2, generator=g).to(d)
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    ya, yb = a(xa), b(xb)
    ya.backward(gy)
    yb.backward(gy)
    torch.cuda.synchronize()
    close(yb, ya, tol=1e-5, what='out')
    close(xb.grad, xa.grad, tol=1e-4, what='dx')
    for (k, p), q in zip(a.named_parameters(), b.parameters()):
        close(q.grad, p.grad, tol=2e-4, what='grad ' + k)
    for (k, u), v in zip(a.named_buffers(), b.buffers()):
        if u.dtype.is_floating_point:
            close(v, u, tol=1e-5, what='buffer ' + k)
        else:
            assert torch.equal(u, v), k


def test_training_step_is_bit_identical_run_to_run():
    """Two fresh trainings of the tiny FFM graph, three optimizer steps each (dropout off): losses, a weight, a BN buffer and an
    EMA weight equal bit for bit.  (Round 1 had fp32 atomics in the SPP backward and the loss-gradient scatter.)"""
    from test_step_gpu import batch, make
    runs = []
    for rep in range(2):
        m, ts, cfg = make('fourier')
        losses = []
        for it in range(3):
            loss, items = ts.step(*batch(cfg, 90 + it))
            losses.append((loss.clone(), items.clone()))
        torch.cuda.synchronize()
        runs.append((losses, m.model[1].conv.weight.detach().clone(), m.model[1].bn.running_var.clone(),
                     m.model[-1].m[0].bias.detach().clone(), ts.ema.ema.model[2].cv3.conv.weight.clone()))
    for (l0, i0), (l1, i1) in zip(runs[0][0], runs[1][0]):
        assert torch.equal(l0, l1) and torch.equal(i0, i1)
    for u, v in zip(runs[0][1:], runs[1][1:]):
        assert torch.equal(u, v)
