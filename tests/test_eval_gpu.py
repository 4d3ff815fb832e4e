"""GPU parity of the evaluation path through the C ABI: Detect decode (models/yolo_test.py:57-68),
Model.fuse() + fused forward (yolo_test.py:304-312, torch_utils.py:181-201, common.py:124-125) and
non_max_suppression (utils/general.py:486-580) against the oracle and the reference-generated fixtures
tests/golden/eval_path.npz."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, tiny_cfg
from test_ops_gpu import close, cl, dev, nchw, nhwc

pytestmark = pytest.mark.gpu


def build(kind):
    from models.yolo_test import Model
    from oracle import portable_init
    from oracle.ref_model import Model as OModel
    cfg = tiny_cfg(kind)
    o = OModel(cfg)
    sd = portable_init.fill_(o.state_dict())
    o.load_state_dict(sd)
    m = Model(tiny_cfg(kind))
    m.load_state_dict(sd, strict=True)
    imgs, _ = portable_init.synth_batch(2, 128, cfg['nc'], per_image=4, seed=1)
    return m.to(dev()).eval(), o.eval(), imgs.float() / 255


@pytest.mark.parametrize('kind', ['fourier', 'add'])
def test_eval_decode_and_fused_model(kind):
    m, o, x = build(kind)
    g = np.load(os.path.join(GOLDEN, 'model_%s_eval.npz' % kind))
    gf = np.load(os.path.join(GOLDEN, 'eval_path.npz'))
    xd = x.to(dev())
    with torch.no_grad():
        (z, pred), _ = m(xd[:, :3], xd[:, 3:])
        close(z, torch.from_numpy(g['z']), tol=1e-4, what='decoded z vs reference')
        for i, p in enumerate(pred):
            close(p, torch.from_numpy(g['pred%d' % i]), tol=1e-4, what='raw head %d' % i)
        m.fuse()
        assert not any(hasattr(mod, 'bn') for mod in m.model.modules() if type(mod).__name__ == 'Conv')
        close(m.model[1].conv.weight, torch.from_numpy(gf['%s.fused_w' % kind]), tol=1e-6, what='folded weight')
        close(m.model[1].conv.bias, torch.from_numpy(gf['%s.fused_b' % kind]), tol=1e-6, what='folded bias')
        (zf, _), _ = m(xd[:, :3], xd[:, 3:])
    close(zf, torch.from_numpy(gf['%s.z_fused' % kind]), tol=1e-4, what='fused z vs reference fuse()')
    close(zf, z, tol=1e-4, what='fused vs unfused')


@pytest.mark.parametrize('act,res', [(1, False), (1, True), (2, False), (0, True)])
def test_conv_bias_act_epilogue(act, res):
    """act(conv(x)+b) [+ residual] in one kernel, incl. a stream-K sized launch."""
    import torch.nn.functional as F
    from mmidet_hip import lib, ops
    g = torch.Generator().manual_seed(10 + act)
    x = torch.randn(2, 64, 24, 20, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) / 24
    b = torch.randn(64, generator=g)
    y = F.conv2d(x, w, b, 1, 1)
    y = F.silu(y) if act == 1 else (F.leaky_relu(y, 0.1) if act == 2 else y)
    if res:
        y = y + x
    d = dev()
    for slots in (0, 7):
        lib.set_streamk_slots(slots)
        try:
            yg = ops.conv_bias_act(nhwc(x).to(d), cl(w).to(d), b.to(d), 1, act, nhwc(x).to(d) if res else None)
        finally:
            lib.set_streamk_slots(0)
        close(nchw(yg), y, tol=1e-4, what='y (streamk slots %d)' % slots)


def test_nms_matches_oracle_and_reference_fixture():
    from oracle import ref_nms
    from utils.general import non_max_suppression
    gf = np.load(os.path.join(GOLDEN, 'eval_path.npz'))
    for name, seed, rows, kw in ref_nms.NMS_CASES:
        pred = ref_nms.synth_predictions(seed, rows=rows)
        out = non_max_suppression(pred.to(dev()), **kw)
        assert len(out) == pred.shape[0]
        for i, t in enumerate(out):
            ref = gf['nms.%s.%d' % (name, i)]
            assert tuple(t.shape) == ref.shape, (name, i, tuple(t.shape), ref.shape)
            # box / conf values are copies or one fp32 product of the inputs: bit-equal, in the same order
            assert np.array_equal(t.cpu().numpy(), ref), (name, i)


def test_nms_eighty_classes_with_filter():
    """nc = 80 (the reference's COCO default): no 64-class limit, class filter as a device byte array."""
    from oracle import ref_nms
    from utils.general import non_max_suppression
    pred = ref_nms.synth_predictions(21, bs=2, rows=200, nc=80)
    for kw in (dict(), dict(classes=[3, 70, 79]), dict(multi_label=True, conf_thres=0.5)):
        out = non_max_suppression(pred.to(dev()), **kw)
        ref = ref_nms.non_max_suppression(pred, **kw)
        for a, b in zip(out, ref):
            assert torch.equal(a.cpu(), b), kw


def test_nms_edge_cases():
    from oracle import ref_nms
    from utils.general import non_max_suppression
    d = dev()
    # identical boxes and scores: ties broken by row order, one survivor per class
    p = torch.zeros(1, 8, 7)
    p[0, :, :4] = torch.tensor([100., 100., 50., 50.])
    p[0, :, 4] = 0.9
    p[0, :, 5] = 0.8
    p[0, 4:, 5], p[0, 4:, 6] = 0.0, 0.8
    out = non_max_suppression(p.to(d))[0].cpu()
    ref = ref_nms.non_max_suppression(p)[0]
    assert torch.equal(out, ref) and out.shape[0] == 2
    # a single row, nothing to suppress; and an image with no row above the threshold
    p = torch.rand(2, 1, 7)
    p[0, 0, 4:] = 0.9
    p[1, 0, 4] = 0.01
    out = non_max_suppression(p.to(d))
    assert out[0].shape == (1, 6) and out[1].shape == (0, 6)
    # full-size row count of a 640x640 image (25200 rows), test.py settings: the max_det cap and ordering
    pred = ref_nms.synth_predictions(9, bs=2, rows=25200, nc=6)
    pred[..., 4] *= (torch.arange(25200) % 50 == 0).float()          # ~500 live rows per image keeps the oracle quick
    out = non_max_suppression(pred.to(d), conf_thres=0.001, iou_thres=0.6, multi_label=True)
    ref = ref_nms.non_max_suppression(pred, conf_thres=0.001, iou_thres=0.6, multi_label=True)
    for a, b in zip(out, ref):
        assert torch.equal(a.cpu(), b)
        assert bool((a[1:, 4] <= a[:-1, 4]).all())                    # sorted by confidence


def test_nms_a_priori_labels_and_candidate_cap():
    """general.py:519-526 (labels) and :501,557-559 (max_nms = 30 000): test.py's settings on a full 640x640 row count with
    every row live -- 25 200 rows x 6 classes = ~150 000 candidate pairs, of which the 30 000 most confident take part."""
    from oracle import ref_nms
    from utils.general import non_max_suppression
    d = dev()
    pred = ref_nms.synth_predictions(31, bs=2, rows=300, nc=4)
    labels = [torch.tensor([[1, 320., 320., 100., 80.], [3, 100., 500., 60., 60.]]), torch.zeros((0, 5))]
    out = non_max_suppression(pred.to(d), labels=labels)
    ref = ref_nms.non_max_suppression(pred, labels=labels)
    for a, b in zip(out, ref):
        assert torch.equal(a.cpu(), b)
    assert float(out[0][0, 4]) == 1.0                                  # the a-priori boxes lead with confidence 1
    pred = ref_nms.synth_predictions(32, bs=2, rows=25200, nc=6)
    pred[1, :, 4] *= (torch.arange(25200) % 40 == 0).float()          # image 1 stays far below the cap
    kw = dict(conf_thres=0.001, iou_thres=0.6, multi_label=True)
    out = non_max_suppression(pred.to(d), **kw)
    ref = ref_nms.non_max_suppression(pred, **kw)
    for a, b in zip(out, ref):
        assert a.shape == b.shape and torch.equal(a.cpu(), b)


def test_checkpoint_to_detections(tmp_path):
    """detect_twostream.py:33,89-93 end to end: whole-object checkpoint -> attempt_load (fp32, fused, eval) -> forward ->
    non_max_suppression; the detections equal those of the model that was saved (unfused) up to fp32 rounding."""
    from copy import deepcopy
    from models.experimental import attempt_load
    from utils.general import non_max_suppression
    m, _, x = build('fourier')
    f = str(tmp_path / 'best.pt')
    torch.save({'model': deepcopy(m).cpu(), 'ema': None}, f)
    loaded = attempt_load(f, map_location=dev())
    xd = x.to(dev())
    with torch.no_grad():
        z0 = m(xd[:, :3], xd[:, 3:])[0][0]
        z1 = loaded(xd[:, :3], xd[:, 3:])[0][0]
    close(z1, z0, tol=1e-4, what='decoded predictions after save / load / fuse')
    d0 = non_max_suppression(z0, 0.2, 0.45)
    d1 = non_max_suppression(z1, 0.2, 0.45)
    assert [t.shape for t in d0] == [t.shape for t in d1]
    for a, b in zip(d0, d1):
        if a.numel():
            close(a[:, :5], b[:, :5], tol=1e-3, what='boxes + confidences')
            assert torch.equal(a[:, 5], b[:, 5])
