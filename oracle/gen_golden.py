"""Golden-vector generator (test infrastructure; runs ONLY in the build container, where /root/reference
exists).  Imports the real reference with inert stubs for the packages its import chain names but its
arithmetic never touches, drives it with hash-generated weights/inputs
(oracle/portable_init.py) and writes small .npz fixtures to tests/golden/.  Only data is written: inputs
are regenerated from the hash, outputs are stored.

    python oracle/gen_golden.py            # regenerate all fixtures
    python oracle/gen_golden.py eval       # only tests/golden/eval_path.npz (Model.fuse() and NMS)
"""
import contextlib
import importlib.machinery
import io
import os
import sys
from copy import deepcopy
from unittest import mock

import numpy as np
import torch
import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
OUT = os.path.join(REPO, 'tests', 'golden')
sys.path.insert(0, REPO)
from oracle import portable_init  # noqa: E402
from oracle.ref_loss import scaled_hyp  # noqa: E402


def import_reference():
    sys.dont_write_bytecode = True
    for name in ('cv2', 'torchvision', 'torchvision.ops', 'torchvision.models', 'seaborn', 'thop', 'torchsummary'):
        m = mock.MagicMock()
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__path__ = []
        sys.modules[name] = m
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        from models.yolo_test import Model
        from utils.loss import ComputeLoss
        from utils.general import bbox_iou
        from models.common import extract_frequency2, Seperation_loss
    return Model, ComputeLoss, bbox_iou, extract_frequency2, Seperation_loss


def tiny_cfg(kind):
    """The tiny graphs of the fixtures (also rebuilt by tests/ from the YAMLs committed under mmi-det_amd/)."""
    if kind == 'fourier':
        with open(os.path.join(REF, 'models/transformer/yolov5l_fusion_transformer_M3FD_fuse3_fourier.yaml')) as f:
            d = yaml.safe_load(f)
        d['depth_multiple'], d['width_multiple'] = 0.33, 0.25
        d['backbone'][6][3] = [32]          # GPT1_fourier channel is not width-scaled by the reference (B3)
    else:
        with open(os.path.join(REF, 'models/transformer/yolov5s_fusion_add_vedai.yaml')) as f:
            d = yaml.safe_load(f)
        d['width_multiple'] = 0.25
    return d


def run_model_case(Model, ComputeLoss, kind, bs, size, train):
    quiet = io.StringIO()
    with contextlib.redirect_stdout(quiet):
        cfg = tiny_cfg(kind)
        model = Model(deepcopy(cfg))
    sd = model.state_dict()
    portable_init.fill_(sd)
    model.load_state_dict(sd)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    nc = cfg['nc']
    model.nc, model.gr = nc, 1.0
    model.hyp = scaled_hyp(nc, size)
    imgs, targets = portable_init.synth_batch(bs, size, nc, per_image=4, seed=1)
    x = imgs.float() / 255
    rgb, ir = x[:, :3], x[:, 3:]
    out = {}
    layer_stats = []

    def hook(mod, inp, o):
        ts = o if isinstance(o, (list, tuple)) else [o]
        t = ts[0]
        layer_stats.append([float(t.float().mean()), float(t.float().abs().mean())])
    hooks = [m.register_forward_hook(hook) for m in list(model.model)[:-1]]
    model.train(train)
    with contextlib.redirect_stdout(quiet):
        if train:
            pred, comb = model(rgb, ir)
            loss_fn = ComputeLoss(model)
            loss, items = loss_fn(pred, targets, comb.reshape(-1))
            loss.backward()
            tcls, tbox, indices, anch = loss_fn.build_targets(pred, targets)
        else:
            with torch.no_grad():
                (z, pred), comb = model(rgb, ir)
            out['z'] = z.numpy()
    for h in hooks:
        h.remove()
    out['layer_stats'] = np.array(layer_stats, np.float32)
    for i, p in enumerate(pred):
        out['pred%d' % i] = p.detach().numpy()
    out['combine'] = comb.detach().numpy()
    for k in ('ContrastiveValue', 'SSIMloss', 'PTLoss', 'Entropy_loss'):
        out[k] = torch.as_tensor(getattr(model, k)).detach().float().numpy()
    if train:
        out['loss'] = loss.detach().numpy()
        out['items'] = items.numpy()
        names, gnorm = [], []
        for n, p in model.named_parameters():
            if p.grad is not None:
                names.append(n)
                gnorm.append(float(p.grad.double().norm()))
        out['grad_names'] = np.array(names)
        out['grad_norms'] = np.array(gnorm, np.float64)
        out['grad_Enhance_conv2'] = model.Enhance.conv2.weight.grad.numpy()
        out['grad_det0_bias'] = model.model[-1].m[0].bias.grad.numpy()
        for i in range(3):
            out['tcls%d' % i] = tcls[i].numpy()
            out['tbox%d' % i] = tbox[i].numpy()
            out['anch%d' % i] = anch[i].numpy()
            out['idx%d' % i] = torch.stack(indices[i]).numpy()
        rs = model.state_dict()
        for k in ('Enhance.bn2.running_mean', 'Enhance.bn2.running_var', 'model.1.bn.running_mean',
                  'model.1.bn.running_var'):
            out['after.' + k] = rs[k].numpy()
    n_params = sum(p.numel() for p in model.parameters())
    out['n_params'] = np.array(n_params)
    out['sd_keys'] = np.array(list(model.state_dict().keys()))
    return out


def gen_eval_path(Model):
    """tests/golden/eval_path.npz: (1) the reference's Model.fuse() (yolo_test.py:304-312) + eval forward on the tiny
    graphs; (2) the reference's non_max_suppression (general.py:486-580) run around oracle.ref_nms.greedy_nms, which
    stands in for the torchvision.ops.nms call the image cannot provide (see oracle/ref_nms.py: that part is unpinned)."""
    from oracle import ref_nms
    o = {}
    quiet = io.StringIO()
    for kind in ('fourier', 'add'):
        with contextlib.redirect_stdout(quiet):
            cfg = tiny_cfg(kind)
            model = Model(deepcopy(cfg))
        sd = model.state_dict()
        portable_init.fill_(sd)
        model.load_state_dict(sd)
        imgs, _ = portable_init.synth_batch(2, 128, cfg['nc'], per_image=4, seed=1)
        x = imgs.float() / 255
        model.eval()
        with contextlib.redirect_stdout(quiet), torch.no_grad():
            model.fuse()
            (z, _), _ = model(x[:, :3], x[:, 3:])
        o['%s.z_fused' % kind] = z.numpy()
        o['%s.fused_w' % kind] = model.model[1].conv.weight.detach().numpy()
        o['%s.fused_b' % kind] = model.model[1].conv.bias.detach().numpy()
    import utils.general as G
    sys.modules['torchvision'].ops.nms = ref_nms.greedy_nms
    G.torchvision.ops.nms = ref_nms.greedy_nms
    for name, seed, rows, kw in ref_nms.NMS_CASES:
        pred = ref_nms.synth_predictions(seed, rows=rows)
        out = G.non_max_suppression(pred.clone(), **kw)
        for i, t in enumerate(out):
            o['nms.%s.%d' % (name, i)] = t.numpy()
    fn = os.path.join(OUT, 'eval_path.npz')
    np.savez_compressed(fn, **o)
    print('wrote', fn, os.path.getsize(fn))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    Model, ComputeLoss, bbox_iou, extract_frequency2, Seperation_loss = import_reference()
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == 'eval':      # only the §8 f-3 fixtures
        gen_eval_path(Model)
        return

    for kind, bs, size in (('fourier', 2, 128), ('add', 2, 128)):
        for train in (True, False):
            o = run_model_case(Model, ComputeLoss, kind, bs, size, train)
            fn = os.path.join(OUT, 'model_%s_%s.npz' % (kind, 'train' if train else 'eval'))
            np.savez_compressed(fn, **o)
            print('wrote', fn, os.path.getsize(fn))

    # ---- op-level pins ------------------------------------------------------------------------------
    u = portable_init._u01
    o = {}
    # CIoU (general.py:403-447) on hash boxes
    n = 512
    b1 = torch.from_numpy(u('ciou:b1', n * 4).reshape(n, 4)) * torch.tensor([3., 3., 4., 4.]) + torch.tensor([-.5, -.5, .05, .05])
    b2 = torch.from_numpy(u('ciou:b2', n * 4).reshape(n, 4)) * torch.tensor([1., 1., 6., 6.]) + torch.tensor([0., 0., .05, .05])
    o['ciou'] = bbox_iou(b1.T, b2, x1y1x2y2=False, CIoU=True).numpy()
    # extract_frequency2 (common.py:37-69) on an 8x8 plane stack
    img = torch.from_numpy(u('freq:x', 2 * 5 * 8 * 8).reshape(2, 5, 8, 8)) * 4 - 1
    lo, hi = extract_frequency2(img)
    o['freq_lo'] = lo.real.float().numpy() if lo.is_complex() else lo.float().numpy()
    o['freq_hi'] = hi.real.float().numpy() if hi.is_complex() else hi.float().numpy()
    # Seperation_loss (common.py:128-139)
    M = torch.from_numpy(u('sep:M', 36 * 64).reshape(36, 64))
    o['sep'] = Seperation_loss(M).numpy()
    np.savez_compressed(os.path.join(OUT, 'ops.npz'), **o)

    # ---- build_targets at full-size grids (loss.py:189-245): B=16, 32 objects/image, 640x640, nc=6 -------
    class _Det:
        pass

    class _M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1))
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = tiny_cfg('fourier')
        ref = Model(deepcopy(cfg))
    ref.nc, ref.gr, ref.hyp = 6, 1.0, scaled_hyp(6, 640)
    lf = ComputeLoss(ref)
    o = {}
    for tag, bs, per in (('b16x32', 16, 32), ('b16x8', 16, 8), ('b1x1', 1, 1), ('b4x0', 4, 0)):
        _, tg = portable_init.synth_batch(bs, 32, 6, per_image=per, seed=7)
        p = [torch.zeros(bs, 3, 640 // s, 640 // s, 11) for s in (8, 16, 32)]
        tcls, tbox, indices, anch = lf.build_targets(p, tg)
        for i in range(3):
            o['%s.tcls%d' % (tag, i)] = tcls[i].numpy()
            o['%s.tbox%d' % (tag, i)] = tbox[i].numpy()
            o['%s.anch%d' % (tag, i)] = anch[i].numpy()
            o['%s.idx%d' % (tag, i)] = torch.stack(indices[i]).numpy() if len(indices[i][0]) or True else None
    np.savez_compressed(os.path.join(OUT, 'build_targets.npz'), **o)
    gen_eval_path(Model)
    print('done')


if __name__ == '__main__':
    main()
