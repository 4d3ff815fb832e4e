"""CPU: the oracle (oracle/ref_model.py, oracle/ref_loss.py) against the fixtures produced by the real reference
(oracle/gen_golden.py).  This is what pins the oracle; the GPU parity tests then compare the HIP path with it."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, tiny_cfg
from oracle import portable_init
from oracle.ref_loss import ComputeLoss, bbox_ciou, scaled_hyp
from oracle.ref_model import Model, extract_frequency2, separation_loss


def build_oracle(kind, size=128):
    cfg = tiny_cfg(kind)
    m = Model(cfg, dropout=0.0)
    m.load_state_dict(portable_init.fill_(m.state_dict()))
    m.nc, m.gr, m.hyp = cfg['nc'], 1.0, scaled_hyp(cfg['nc'], size)
    return m, cfg


@pytest.mark.parametrize('kind', ['fourier', 'add'])
def test_state_dict_keys_and_param_count(kind):
    g = np.load(os.path.join(GOLDEN, 'model_%s_train.npz' % kind))
    m, _ = build_oracle(kind)
    assert list(m.state_dict().keys()) == list(g['sd_keys'])
    assert sum(p.numel() for p in m.parameters()) == int(g['n_params'])


@pytest.mark.parametrize('kind', ['fourier', 'add'])
def test_train_step_matches_reference(kind):
    g = np.load(os.path.join(GOLDEN, 'model_%s_train.npz' % kind))
    m, cfg = build_oracle(kind)
    imgs, targets = portable_init.synth_batch(2, 128, cfg['nc'], per_image=4, seed=1)
    x = imgs.float() / 255
    m.train()
    pred, comb = m(x[:, :3], x[:, 3:])
    for i in range(3):
        np.testing.assert_allclose(pred[i].detach().numpy(), g['pred%d' % i], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(comb.numpy(), g['combine'], rtol=1e-5, atol=1e-6)
    for k in ('ContrastiveValue', 'SSIMloss', 'PTLoss', 'Entropy_loss'):
        np.testing.assert_allclose(torch.as_tensor(getattr(m, k)).detach().float().numpy(), g[k], rtol=1e-4, atol=1e-6)
    lf = ComputeLoss(m)
    loss, items = lf(pred, targets, comb.reshape(-1))
    assert tuple(loss.shape) == tuple(g['loss'].shape)          # (1,) with FFM, (1,1) without (reference quirk)
    np.testing.assert_allclose(loss.detach().numpy(), g['loss'], rtol=1e-5)
    np.testing.assert_allclose(items.numpy(), g['items'], rtol=1e-5)
    loss.backward()
    grads = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
    assert list(grads.keys()) == list(g['grad_names'])
    mine = np.array([float(v.double().norm()) for v in grads.values()])
    np.testing.assert_allclose(mine, g['grad_norms'], rtol=2e-3, atol=1e-7)
    np.testing.assert_allclose(m.Enhance.conv2.weight.grad.numpy(), g['grad_Enhance_conv2'], rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(m.model[-1].m[0].bias.grad.numpy(), g['grad_det0_bias'], rtol=1e-4, atol=1e-7)
    tcls, tbox, idx, anch = lf.build_targets(pred, targets)
    for i in range(3):
        assert np.array_equal(tcls[i].numpy(), g['tcls%d' % i])
        assert np.array_equal(torch.stack(idx[i]).numpy(), g['idx%d' % i])
        assert np.array_equal(tbox[i].numpy(), g['tbox%d' % i])
        assert np.array_equal(anch[i].numpy(), g['anch%d' % i])
    sd = m.state_dict()
    for k in ('Enhance.bn2.running_mean', 'Enhance.bn2.running_var', 'model.1.bn.running_mean', 'model.1.bn.running_var'):
        np.testing.assert_allclose(sd[k].numpy(), g['after.' + k], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('kind', ['fourier', 'add'])
def test_eval_forward_matches_reference(kind):
    g = np.load(os.path.join(GOLDEN, 'model_%s_eval.npz' % kind))
    m, cfg = build_oracle(kind)
    imgs, _ = portable_init.synth_batch(2, 128, cfg['nc'], per_image=4, seed=1)
    x = imgs.float() / 255
    m.eval()
    with torch.no_grad():
        (z, pred), comb = m(x[:, :3], x[:, 3:])
    np.testing.assert_allclose(z.numpy(), g['z'], rtol=1e-4, atol=1e-4)
    for i in range(3):
        np.testing.assert_allclose(pred[i].numpy(), g['pred%d' % i], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(comb.numpy(), g['combine'], rtol=1e-5, atol=1e-6)


def test_op_pins():
    g = np.load(os.path.join(GOLDEN, 'ops.npz'))
    u = portable_init._u01
    n = 512
    b1 = torch.from_numpy(u('ciou:b1', n * 4).reshape(n, 4)) * torch.tensor([3., 3., 4., 4.]) + torch.tensor([-.5, -.5, .05, .05])
    b2 = torch.from_numpy(u('ciou:b2', n * 4).reshape(n, 4)) * torch.tensor([1., 1., 6., 6.]) + torch.tensor([0., 0., .05, .05])
    assert np.array_equal(bbox_ciou(b1.T, b2).numpy(), g['ciou'])
    img = torch.from_numpy(u('freq:x', 2 * 5 * 8 * 8).reshape(2, 5, 8, 8)) * 4 - 1
    lo, hi = extract_frequency2(img)
    assert np.array_equal(lo.float().numpy(), g['freq_lo'])
    assert np.array_equal(hi.float().numpy(), g['freq_hi'])
    M = torch.from_numpy(u('sep:M', 36 * 64).reshape(36, 64))
    np.testing.assert_allclose(separation_loss(M).numpy(), g['sep'], rtol=1e-6)


@pytest.mark.parametrize('tag,bs,per', [('b16x32', 16, 32), ('b16x8', 16, 8), ('b1x1', 1, 1), ('b4x0', 4, 0)])
def test_build_targets_full_size_bit_exact(tag, bs, per):
    g = np.load(os.path.join(GOLDEN, 'build_targets.npz'))
    m, _ = build_oracle('fourier', 640)
    m.nc, m.hyp = 6, scaled_hyp(6, 640)
    lf = ComputeLoss(m)
    _, tg = portable_init.synth_batch(bs, 32, 6, per_image=per, seed=7)
    p = [torch.zeros(bs, 3, 640 // s, 640 // s, 11) for s in (8, 16, 32)]
    tcls, tbox, idx, anch = lf.build_targets(p, tg)
    for i in range(3):
        assert np.array_equal(tcls[i].numpy(), g['%s.tcls%d' % (tag, i)])
        assert np.array_equal(torch.stack(idx[i]).numpy(), g['%s.idx%d' % (tag, i)])
        assert np.array_equal(tbox[i].numpy(), g['%s.tbox%d' % (tag, i)])
        assert np.array_equal(anch[i].numpy(), g['%s.anch%d' % (tag, i)])


# ---- evaluation path: Model.fuse() and non_max_suppression -------------------------------------------
@pytest.mark.parametrize('kind', ['fourier', 'add'])
def test_oracle_fuse_matches_reference(kind):
    """oracle fuse_conv_and_bn / Model.fuse() against the reference's own (utils/torch_utils.py:181-201)."""
    from oracle import portable_init
    from oracle.ref_model import Model
    g = np.load(os.path.join(GOLDEN, 'eval_path.npz'))
    cfg = tiny_cfg(kind)
    m = Model(cfg)
    m.load_state_dict(portable_init.fill_(m.state_dict()))
    imgs, _ = portable_init.synth_batch(2, 128, cfg['nc'], per_image=4, seed=1)
    x = imgs.float() / 255
    m.eval().fuse()
    with torch.no_grad():
        (z, _), _ = m(x[:, :3], x[:, 3:])
    np.testing.assert_allclose(m.model[1].conv.weight.numpy(), g['%s.fused_w' % kind], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(m.model[1].conv.bias.numpy(), g['%s.fused_b' % kind], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(z.numpy(), g['%s.z_fused' % kind], rtol=1e-4, atol=1e-4)


def test_oracle_nms_matches_reference_wrapper():
    """oracle.ref_nms.non_max_suppression against the reference's function (general.py:486-580) run around the same
    greedy NMS: every case bit-equal (same torch ops in the same order)."""
    from oracle import ref_nms
    g = np.load(os.path.join(GOLDEN, 'eval_path.npz'))
    for name, seed, rows, kw in ref_nms.NMS_CASES:
        pred = ref_nms.synth_predictions(seed, rows=rows)
        out = ref_nms.non_max_suppression(pred, **kw)
        for i, t in enumerate(out):
            ref = g['nms.%s.%d' % (name, i)]
            assert tuple(t.shape) == ref.shape, (name, i, t.shape, ref.shape)
            assert np.array_equal(t.numpy(), ref), (name, i)
    assert g['nms.none.0'].shape[0] == 0 and g['nms.test_py.0'].shape[0] == 300
