"""GPU parity, path level: the MI355X-native Model + ComputeLoss (through the C ABI) against the oracle on the same
hash-initialised weights and inputs, and against the fixtures written by the reference itself (tests/golden).
fp32 tolerance: 1e-3 relative on activations / loss (BASELINE.json north_star); anchor indices bit-exact."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, tiny_cfg
from test_ops_gpu import close, dev, rel_err

pytestmark = pytest.mark.gpu


def build_pair(kind, size):
    from models.yolo_test import Model
    from oracle import portable_init
    from oracle.ref_loss import scaled_hyp
    from oracle.ref_model import Model as OModel
    cfg = tiny_cfg(kind)
    o = OModel(cfg, dropout=0.0)
    sd = portable_init.fill_(o.state_dict())
    o.load_state_dict(sd)
    m = Model(tiny_cfg(kind))
    m.load_state_dict(sd, strict=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    for mm in (o, m):
        mm.nc, mm.gr, mm.hyp = cfg['nc'], 1.0, scaled_hyp(cfg['nc'], size)
    return m.to(dev()), o, cfg


@pytest.mark.parametrize('kind', ['add', 'fourier'])
def test_train_step_matches_reference_fixture_and_oracle(kind):
    from oracle import portable_init
    from oracle.ref_loss import ComputeLoss as OLoss
    from utils.loss import ComputeLoss
    g = np.load(os.path.join(GOLDEN, 'model_%s_train.npz' % kind))
    m, o, cfg = build_pair(kind, 128)
    imgs, targets = portable_init.synth_batch(2, 128, cfg['nc'], per_image=4, seed=1)
    d = dev()
    x = imgs.to(d).float() / 255                                   # train.py:743
    m.train()
    pred, comb = m(x[:, :3], x[:, 3:])                             # strided NCHW views, as train.py:744-745
    lf = ComputeLoss(m)
    loss, items = lf(pred, targets.to(d), comb.reshape(-1))
    loss.backward()
    torch.cuda.synchronize()
    # --- against the reference's own outputs (fixture) ---
    for i in range(3):
        close(pred[i], torch.from_numpy(g['pred%d' % i]), what='pred%d vs reference' % i)
    assert tuple(loss.shape) == tuple(g['loss'].shape)
    close(loss, torch.from_numpy(g['loss']), what='loss vs reference')
    close(items, torch.from_numpy(g['items']), what='loss items vs reference')
    if kind == 'fourier':
        close(comb, torch.from_numpy(g['combine']), what='Combine_loss', tol=1e-4)
        close(m.SSIMloss, torch.from_numpy(g['SSIMloss']), what='SSIMloss', tol=1e-4)
        close(m.PTLoss, torch.from_numpy(g['PTLoss']), what='PTLoss')
        close(m.Entropy_loss, torch.from_numpy(g['Entropy_loss']), what='Entropy', tol=5e-3)
        close(m.ContrastiveValue, torch.from_numpy(g['ContrastiveValue']), what='Contrastive', tol=1e-5)
    else:
        assert comb.numel() == 0
    grads = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
    assert list(grads.keys()) == list(g['grad_names'])
    mine = torch.tensor([float(v.double().norm()) for v in grads.values()], dtype=torch.float64)
    ref = torch.from_numpy(g['grad_norms'])
    bad = ((mine - ref).abs() > 5e-3 * ref + 1e-7).nonzero().flatten().tolist()
    assert not bad, [(list(grads.keys())[i], float(mine[i]), float(ref[i])) for i in bad[:8]]
    close(m.Enhance.conv2.weight.grad, torch.from_numpy(g['grad_Enhance_conv2']), what='dW CEM conv2', tol=5e-3)
    close(m.model[-1].m[0].bias.grad, torch.from_numpy(g['grad_det0_bias']), what='db Detect0', tol=2e-3)
    tcls, tbox, idx, anch = lf.build_targets(pred, targets.to(d))
    for i in range(3):                                             # integer path: bit-exact
        assert np.array_equal(tcls[i].cpu().numpy(), g['tcls%d' % i])
        assert np.array_equal(torch.stack(idx[i]).cpu().numpy(), g['idx%d' % i])
        assert np.array_equal(tbox[i].cpu().numpy(), g['tbox%d' % i])
        assert np.array_equal(anch[i].cpu().numpy(), g['anch%d' % i])
    sd = m.state_dict()
    for k in ('Enhance.bn2.running_mean', 'Enhance.bn2.running_var', 'model.1.bn.running_mean', 'model.1.bn.running_var'):
        close(sd[k], torch.from_numpy(g['after.' + k]), what=k)
    assert int(sd['model.1.bn.num_batches_tracked']) == 1


@pytest.mark.parametrize('kind', ['add', 'fourier'])
def test_every_parameter_gradient_vs_oracle(kind):
    """Elementwise gradient parity for every parameter.  Uses batch seed 2: with seed 1 (the fixture batch) one SPP
    max-pool window of the IR stream holds two values 1 ulp apart, the arg-max flips between any two fp32 evaluation
    orders and moves ~3e-3 of that stream's gradient (tests/diag/diag_grads.py shows it; fp32-vs-fp64 CPU runs flip too on
    other seeds).  Norm-level parity on the fixture batch is asserted in the test above."""
    from oracle import portable_init
    from oracle.ref_loss import ComputeLoss as OLoss
    from utils.loss import ComputeLoss
    m, o, cfg = build_pair(kind, 128)
    imgs, targets = portable_init.synth_batch(2, 128, cfg['nc'], per_image=4, seed=2)
    d = dev()
    x = imgs.to(d).float() / 255
    m.train()
    pred, comb = m(x[:, :3], x[:, 3:])
    loss, _ = ComputeLoss(m)(pred, targets.to(d), comb.reshape(-1))
    loss.backward()
    o.train()
    x_cpu = imgs.float() / 255
    po, co = o(x_cpu[:, :3], x_cpu[:, 3:])
    lo, _ = OLoss(o)(po, targets, co.reshape(-1))
    lo.backward()
    close(loss, lo, what='loss', tol=1e-4)
    # key_proj.bias has an analytically ZERO gradient (softmax is invariant to a per-query constant); so has any
    # per-channel constant in front of a 1x1 Conv + train-mode BN (e.g. model.29.ln_f.bias): skip numerically-nil grads
    og = dict(o.named_parameters())
    errs = [(rel_err(p.grad, og[n].grad), n) for n, p in m.named_parameters()
            if p.grad is not None and 'key_proj.bias' not in n and float(og[n].grad.norm()) > 1e-5]
    assert len(errs) > 100
    worst = max(errs)
    assert worst[0] < 5e-3, worst
    med = np.median([e for e, _ in errs])
    assert med < 5e-4, med


@pytest.mark.parametrize('kind', ['add', 'fourier'])
def test_eval_forward_matches_reference_fixture(kind):
    from oracle import portable_init
    g = np.load(os.path.join(GOLDEN, 'model_%s_eval.npz' % kind))
    m, _, cfg = build_pair(kind, 128)
    imgs, _ = portable_init.synth_batch(2, 128, cfg['nc'], per_image=4, seed=1)
    x = imgs.to(dev()).float() / 255
    m.eval()
    with torch.no_grad():
        (z, pred), comb = m(x[:, :3], x[:, 3:])
    close(z, torch.from_numpy(g['z']), what='inference output')
    for i in range(3):
        close(pred[i], torch.from_numpy(g['pred%d' % i]), what='pred%d' % i)


@pytest.mark.parametrize('bs,per,size,nc', [(4, 8, 256, 6), (16, 32, 640, 6), (2, 0, 128, 9), (1, 1, 64, 1)])
def test_detect_loss_kernel_vs_oracle(bs, per, size, nc):
    """ComputeLoss value, items and d loss / d pred on random head outputs (incl. duplicate cells, nt=0, nc=1)."""
    from oracle import portable_init
    from oracle.ref_loss import ComputeLoss as OLoss, scaled_hyp
    from utils.loss import ComputeLoss

    class Det:
        pass

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1))
    mo, mg = M(), M().to(dev())
    a = torch.tensor([[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]).float().view(3, 3, 2)
    a = a / torch.tensor([8., 16., 32.]).view(3, 1, 1)
    for mm, dv in ((mo, 'cpu'), (mg, dev())):
        det = Det()
        det.nl, det.na, det.nc, det.anchors, det.stride = 3, 3, nc, a.to(dv), torch.tensor([8., 16., 32.])
        mm.model = [det]
        mm.hyp, mm.gr = scaled_hyp(nc, size), 1.0
    _, tg = portable_init.synth_batch(bs, 32, nc, per_image=per, seed=11)
    g = torch.Generator().manual_seed(bs * per + size)
    p = [torch.randn(bs, 3, size // s, size // s, nc + 5, generator=g) for s in (8, 16, 32)]
    comb = torch.tensor([0.37])
    pr = [t.clone().requires_grad_() for t in p]
    # duplicate (b,a,gj,gi) cells: torch's non-accumulating index_put_ (loss.py:135) is UNDEFINED for duplicates and
    # racy on a threaded CPU; single-threaded it is "last record wins", which is the order the HIP kernel fixes
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        lo, io = OLoss(mo)(pr, tg, comb)
    finally:
        torch.set_num_threads(nthreads)
    lo.sum().backward()
    pg = [t.to(dev()).requires_grad_() for t in p]
    lg, ig = ComputeLoss(mg)(pg, tg.to(dev()), comb.to(dev()))
    (lg.sum() * 1.0).backward()
    close(lg, lo, what='loss', tol=1e-5)
    close(ig, io, what='items', tol=1e-5)
    for i in range(3):
        close(pg[i].grad, pr[i].grad, what='dpred%d' % i, tol=1e-4)
    # empty CombineLoss -> (1,1) shaped loss (reference quirk), Flag=False -> detection loss only
    l2, _ = ComputeLoss(mg)([t.detach() for t in pg], tg.to(dev()), torch.zeros(0, device=dev()))
    assert tuple(l2.shape) == (1, 1)
    l3, i3 = ComputeLoss(mg)([t.detach() for t in pg], tg.to(dev()), comb.to(dev()), Flag=False)
    close(l3, io[3:4] * bs, what='Flag=False', tol=1e-5)


@pytest.mark.parametrize('gemm', [0, 2, 3], ids=['fp32_mfma', 'bf16x6', 'bf16x9'])
def test_yolov5l_640_train_step_matches_oracle(gemm):
    """The BASELINE model at its real layer shapes (yolov5l two-stream-fourier, 640x640, batch 2): forward, loss and
    gradients against the oracle.  This is where the stream-K schedule, the 128x128 tiles, the parity-class dgrad and the
    split-K plans of the full-size layers are exercised end to end (the tiny fixtures never reach them).
    gemm = 2, 3: the same check at the same tolerances with the opt-in three-term split-bf16 arithmetics.

    Gradient tolerances.  Predictions and loss agree to 2e-5 / 1e-6 for every seed.  Parameter gradients of two fp32
    evaluations that differ only in summation order agree to 7e-4 .. 1.5e-3 (median over all tensors: rounding noise amplified
    ~1e4 times by the depth) -- unless a discrete decision of the backward (a max-pool arg-max, an activation sign within one
    ulp of the tie) falls the other way, which moves every gradient upstream of it by ~1e-2.  Which arithmetic that happens to
    depends on the seed: of four seeds it hit the fp32 path once (median 3e-3, worst 1.1e-2) and the split forms once
    (4e-3, 2.3e-2) (profiles/r01_gemm_modes_full_size_gradients.txt).  So the bound is the event level, not the rounding level."""
    import yaml
    from models.yolo_test import Model
    from oracle import portable_init
    from oracle.ref_loss import ComputeLoss as OLoss, scaled_hyp
    from oracle.ref_model import Model as OModel
    from utils.loss import ComputeLoss
    from mmidet_hip import lib
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, '..', 'mmi-det_amd', 'models', 'transformer',
                           'yolov5l_fusion_transformer_M3FD_fuse3_fourier.yaml')) as f:
        cfg = yaml.safe_load(f)
    cfg['nc'] = 6
    o = OModel(cfg, dropout=0.0)
    sd = portable_init.fill_(o.state_dict())
    o.load_state_dict(sd)
    m = Model(cfg)
    m.load_state_dict(sd, strict=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    for mm in (o, m):
        mm.nc, mm.gr, mm.hyp = 6, 1.0, scaled_hyp(6, 640)
    m = m.to(dev()).train()
    o.train()
    imgs, targets = portable_init.synth_batch(2, 640, 6, per_image=8, seed=3)
    x = imgs.float() / 255
    po, co = o(x[:, :3], x[:, 3:])
    lo, io = OLoss(o)(po, targets, co.reshape(-1))
    lo.backward()
    xd = x.to(dev())
    lib.set_gemm_precision(gemm)
    try:
        pg, cg = m(xd[:, :3], xd[:, 3:])
        lg, ig = ComputeLoss(m)(pg, targets.to(dev()), cg.reshape(-1))
        lg.backward()
        torch.cuda.synchronize()
    finally:
        lib.set_gemm_precision(0)
    for i in range(3):
        close(pg[i], po[i], what='pred%d' % i)
    close(lg, lo, what='loss', tol=1e-4)
    close(ig, io, what='loss items', tol=1e-4)
    close(cg, co, what='Combine_loss', tol=1e-4)
    og = dict(o.named_parameters())
    checked = 0
    errs = []
    for n, p in m.named_parameters():
        if not any(k in n for k in ('model.1.conv', 'model.2.m.0.cv2.conv', 'model.10.m.4.cv2.conv', 'model.17.m.8.cv1.conv',
                                    'model.23.conv', 'model.25.cv3.conv', 'model.29.trans_blocks.3.mlp.0.weight',
                                    'model.13.trans_blocks.0.sa.que_proj.weight', 'model.6.conv2.weight', 'Enhance.conv3',
                                    'model.49.m.1', 'model.38.m.0.cv2.bn')):
            continue
        r = og[n].grad
        if r is None or float(r.norm()) < 1e-9:
            continue
        errs.append((rel_err(p.grad, r), n))
        checked += 1
    errs.sort()
    assert errs[len(errs) // 2][0] < 8e-3 and errs[-1][0] < 4e-2, errs[-4:]
    assert checked >= 12


@pytest.mark.parametrize('bs', [1, 3])
def test_odd_batch_sizes_match_oracle(bs):
    """B=1 (the Contrast Bridge has no neighbour pair: its value is NaN in the reference too, and it feeds nothing) and an
    odd batch: forward, loss and a gradient against the oracle."""
    from oracle import portable_init
    from oracle.ref_loss import ComputeLoss as OLoss
    from utils.loss import ComputeLoss
    m, o, cfg = build_pair('fourier', 128)
    imgs, targets = portable_init.synth_batch(bs, 128, cfg['nc'], per_image=4, seed=5)
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
    assert torch.isfinite(lg).all()
    for i in range(3):
        close(pg[i], po[i], what='pred%d' % i)
    close(lg, lo, what='loss', tol=1e-4)
    close(ig, io, what='items', tol=1e-4)
    w, wo = m.model[1].conv.weight, o.model[1].conv.weight
    close(w.grad, wo.grad, what='dW(model.1.conv)', tol=2e-3)


def test_rectangular_image_matches_oracle():
    """Non-square input (96 x 160, both multiples of 32 as general.py:140-146 enforces): adaptive 8x8 pooling windows that
    are neither uniform nor equal along the two axes, rectangular grids in the loss."""
    from oracle.ref_loss import ComputeLoss as OLoss
    from oracle.portable_init import _u01
    from utils.loss import ComputeLoss
    import numpy as np
    m, o, cfg = build_pair('fourier', 160)
    bs, h, w = 2, 96, 160
    x = torch.from_numpy(_u01('rect', bs * 6 * h * w).astype(np.float32)).reshape(bs, 6, h, w)
    targets = torch.tensor([[0, 1, .3, .4, .2, .3], [0, 3, .7, .6, .1, .2], [1, 2, .5, .5, .4, .3], [1, 0, .2, .8, .15, .1]])
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
        assert tuple(pg[i].shape) == tuple(po[i].shape)
        close(pg[i], po[i], what='pred%d' % i)
    close(lg, lo, what='loss', tol=1e-4)
    close(cg, co, what='Combine_loss', tol=1e-4)
    close(m.model[1].conv.weight.grad, o.model[1].conv.weight.grad, what='dW(model.1.conv)', tol=2e-3)
    close(m.model[6].conv1.weight.grad, o.model[6].conv1.weight.grad, what='dW(FFM conv1)', tol=2e-3)
