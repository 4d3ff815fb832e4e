"""GPU: the reference's own training loop around the native model.

train.py:568-589 (three SGD groups, nesterov), 680 (`model.half().float()`), 683-686 (DistributedDataParallel wrapper), 706
(GradScaler), 765-773 (warm-up writes into param_groups), 784-804 (autocast forward, scaled backward, scaler.step / update,
zero_grad, ModelEMA.update) -- restated here in this repo's words and driven with synthetic batches -- must train the native
`models.yolo_test.Model` exactly as this package's fused step (mmidet_hip.train_step.TrainStep) does, and as the CPU oracle
does in plain fp32.  What that checks at the boundary: `amp.autocast` around the forward is harmless for the custom autograd
functions (they compute in fp32 whatever the context says); GradScaler's power-of-two loss scale passes through every backward
kernel exactly; torch.optim.SGD, ModelEMA and DDP see ordinary Parameters / .grad tensors (lazy C3 packing, twin launches and
the wgrad side streams included)."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import tiny_cfg
from test_ops_gpu import close, dev

pytestmark = pytest.mark.gpu

NBS = 64


def _models(kind='fourier'):
    from models.yolo_test import Model
    from oracle import portable_init
    from oracle.ref_model import Model as OModel
    cfg = tiny_cfg(kind)
    o = OModel(cfg, dropout=0.0)
    sd = portable_init.fill_(o.state_dict())
    o.load_state_dict(sd)

    def native():
        m = Model(copy.deepcopy(cfg))
        m.load_state_dict(sd, strict=True)
        for mod in m.modules():
            if isinstance(mod, nn.Dropout):
                mod.p = 0.0
        return m.to(dev()).train()
    return cfg, o.train(), native


def _batches(cfg, n, size=128, bs=2):
    from oracle import portable_init
    return [portable_init.synth_batch(bs, size, cfg['nc'], per_image=4, seed=40 + i) for i in range(n)]


def _groups(model):
    """train.py:572-579."""
    pg0, pg1, pg2 = [], [], []
    for _, v in model.named_modules():
        if hasattr(v, 'bias') and isinstance(v.bias, nn.Parameter):
            pg2.append(v.bias)
        if isinstance(v, nn.BatchNorm2d):
            pg0.append(v.weight)
        elif hasattr(v, 'weight') and isinstance(v.weight, nn.Parameter):
            pg1.append(v.weight)
    return pg0, pg1, pg2


def _schedule(ni, nw, hyp, j):
    """train.py:765-773 at epoch 0 (lf(0) = 1): (lr, momentum) of group j at integrated batch ni."""
    xi = [0, nw]
    lr = float(np.interp(ni, xi, [hyp['warmup_bias_lr'] if j == 2 else 0.0, hyp['lr0']]))
    mom = float(np.interp(ni, xi, [hyp['warmup_momentum'], hyp['momentum']]))
    return lr, mom


def reference_style_loop(model, batches, cfg, size, device, amp=True, wrap_ddp=False, half_float=True, ni0=300, nw=1000):
    """The reference's per-batch loop (see the module docstring for the line numbers).  Returns (losses, model, ema)."""
    from mmidet_hip.train_step import HYP_SCRATCH          # data/hyp.scratch.yaml:6-22 of the reference
    from oracle.ref_loss import scaled_hyp
    from utils.torch_utils import ModelEMA
    if device.type == 'cuda':
        from utils.loss import ComputeLoss
    else:
        from oracle.ref_loss import ComputeLoss
    hyp0 = dict(HYP_SCRATCH)
    bs = batches[0][0].shape[0]
    world = 1
    accumulate = max(round(NBS / (bs * world)), 1)
    wd = hyp0['weight_decay'] * bs * world * accumulate / NBS                                   # train.py:568-570
    pg0, pg1, pg2 = _groups(model)
    opt = torch.optim.SGD(pg0, lr=hyp0['lr0'], momentum=hyp0['momentum'], nesterov=True)      # train.py:585
    opt.add_param_group({'params': pg1, 'weight_decay': wd})
    opt.add_param_group({'params': pg2})
    for g in opt.param_groups:
        g['initial_lr'] = g['lr']
    ema = ModelEMA(model)                                                                      # train.py:609
    if half_float:
        model.half().float()                                                                   # train.py:680
    net = model
    if wrap_ddp:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[device.index or 0], output_device=device.index or 0)
    net.nc, net.hyp, net.gr = cfg['nc'], scaled_hyp(cfg['nc'], size), 1.0                     # train.py:689-695
    scaler = torch.amp.GradScaler('cuda', enabled=amp and device.type == 'cuda')               # train.py:706
    compute_loss = ComputeLoss(net)
    losses = []
    opt.zero_grad()
    for i, (imgs_u8, targets) in enumerate(batches):
        ni = ni0 + i
        imgs = imgs_u8.to(device, non_blocking=True).float() / 255.0                            # train.py:743-745
        rgb, ir = imgs[:, :3, :, :], imgs[:, 3:, :, :]
        for j, g in enumerate(opt.param_groups):                                               # train.py:765-773
            g['lr'], g['momentum'] = _schedule(ni, nw, hyp0, j)
        with torch.amp.autocast('cuda', enabled=amp and device.type == 'cuda'):                # train.py:784-791
            pred, comb = net(rgb, ir)
            loss, items = compute_loss(pred, targets.to(device), comb.reshape(-1))
            if wrap_ddp:
                loss = loss * world
        scaler.scale(loss).backward()                                                          # train.py:796
        scaler.step(opt)                                                                       # train.py:799-804 (accumulate: every batch here)
        scaler.update()
        opt.zero_grad()
        ema.update(net)
        losses.append(loss.detach().float().cpu().reshape(-1))
    return losses, model, ema


def fused_loop(model, batches, cfg, size, ni0=300, nw=1000):
    """The same schedule through this package's step (one fused SGD+EMA launch, wgrad side streams joined once)."""
    from mmidet_hip.train_step import HYP_SCRATCH, TrainStep
    with torch.no_grad():            # train.py:680 -- what reference_style_loop does through model.half().float()
        for t in list(model.parameters()) + list(model.buffers()):
            if t.dtype == torch.float32:
                t.copy_(t.half().float())
    ts = TrainStep(model, cfg['nc'], size, batches[0][0].shape[0], accumulate=1)
    losses = []
    for i, (imgs_u8, targets) in enumerate(batches):
        for j, g in enumerate(ts.optimizer.param_groups):
            g['lr'], g['momentum'] = _schedule(ni0 + i, nw, HYP_SCRATCH, j)
        loss, _ = ts.step(imgs_u8.to(dev()), targets.to(dev()))
        losses.append(loss.detach().float().cpu().reshape(-1))
    return losses, model, ts.ema


def _compare(tag, la, lb, ma, mb, ea, eb, tol_loss, tol_w):
    for i, (a, b) in enumerate(zip(la, lb)):
        close(a, b, tol_loss, '%s: loss of step %d' % (tag, i))
    sa, sb = ma.state_dict(), mb.state_dict()
    for k in sa:
        if sa[k].dtype.is_floating_point and sa[k].numel():
            close(sa[k], sb[k], tol_w, '%s: %s' % (tag, k))
    ka, kb = ea.ema.state_dict(), eb.ema.state_dict()
    for k in ka:
        if ka[k].dtype.is_floating_point and ka[k].numel():
            close(ka[k], kb[k], tol_w, '%s: ema %s' % (tag, k))


@pytest.mark.parametrize('kind', ['fourier', 'add'])
def test_reference_loop_trains_the_native_model_like_the_fused_step_and_the_oracle(kind):
    cfg, oracle, native = _models(kind)
    batches = _batches(cfg, 3)
    torch.manual_seed(0)
    l_ref, m_ref, e_ref = reference_style_loop(native(), batches, cfg, 128, dev())
    l_fus, m_fus, e_fus = fused_loop(native(), batches, cfg, 128)
    torch.cuda.synchronize()
    # same kernels, same numbers: the loss scale is a power of two, torch.optim.SGD and the fused launch do the same arithmetic
    # (losses to 1e-5; parameters to 2e-4 of their norm: torch.optim.SGD and the fused launch round the same update in a different
    #  order, and BatchNorm biases that start at zero are only as large as three updates -- tests/test_step_gpu.py measures the
    #  optimiser arithmetic itself element by element)
    _compare('autocast + GradScaler + torch.optim.SGD vs the fused step', l_ref, l_fus, m_ref, m_fus, e_ref, e_fus, 1e-5, 2e-4)
    l_cpu, m_cpu, e_cpu = reference_style_loop(oracle, batches, cfg, 128, torch.device('cpu'), amp=False)
    _compare('native loop vs the CPU oracle', l_ref, l_cpu, m_ref, m_cpu, e_ref, e_cpu, 1e-3, 1e-3)
    assert any(p.grad is not None for p in m_ref.parameters()), 'parameters outside every group keep their gradient (pos_emb, sobel_factor)'


def test_reference_loop_under_distributed_data_parallel_world1():
    """`DDP(model, device_ids=[local_rank])` is what the unchanged train.py builds at train.py:683-686: torch's reducer hooks the
    native model's parameters; gradients produced by the HIP kernels (wgrad side streams, twin launches) reach its buckets."""
    import os
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29541')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev())
    try:
        cfg, _, native = _models('fourier')
        batches = _batches(cfg, 3)
        l_ddp, m_ddp, e_ddp = reference_style_loop(native(), batches, cfg, 128, dev(), wrap_ddp=True)
        l_one, m_one, e_one = reference_style_loop(native(), batches, cfg, 128, dev())
        torch.cuda.synchronize()
        _compare('DDP wrapper vs the bare model', l_ddp, l_one, m_ddp, m_one, e_ddp, e_one, 1e-5, 2e-4)
    finally:
        dist.destroy_process_group()
