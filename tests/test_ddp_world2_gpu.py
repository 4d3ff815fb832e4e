"""GPU, world size 2 on ONE card (two processes share cuda:0; rendezvous and collectives over gloo, which carries device tensors
through the host): `TrainStep`'s own data-parallel sequencing -- prepare -> forward -> backward (weight gradients written into
the reducer's buckets by the HIP kernels, grad-ready hooks launching bucket collectives during backward) -> join of the wgrad
streams -> finish -> fused optimizer + EMA -> zero -- for three steps on different batches per rank (train.py:683-686, 741-804 of
the reference under DistributedDataParallel).  RCCL refuses two ranks on one device, so this is the only way to run N > 1 on a
one-GPU box; the transport differs from production (gloo instead of RCCL), everything around it is the production path.

Checked: (1) both ranks hold the same weights afterwards; (2) they equal, to fp32 rounding, a SINGLE process that accumulates the
two ranks' batches (TrainStep(accumulate=2)): the reference scales the loss by world_size and DDP averages, so a world-2 step is
the sum of the two per-rank gradients -- exactly what two accumulated micro-batches give (scaling by 2 is exact in fp32)."""
import os
import sys
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

from conftest import PKG, REPO

pytestmark = pytest.mark.gpu
STEPS = 3


def _build(world, accumulate):
    sys.path[:0] = [PKG, REPO, os.path.join(REPO, 'tests')]
    from test_step_gpu import make
    # (four ranks share the one GPU and the box's host cores: the fusion-off graph keeps that case to a minute)
    m, ts, cfg = make('fourier' if world <= 2 else 'add', world_size=world, accumulate=accumulate)
    return m, ts, cfg


def _build_single(world):
    """The accumulating single process of the four-rank case: the same (fusion-off) graph the ranks train."""
    sys.path[:0] = [PKG, REPO, os.path.join(REPO, 'tests')]
    from test_step_gpu import make
    return make('add', world_size=1, accumulate=world)


def _batch(cfg, it, rank):
    from test_step_gpu import batch
    return batch(cfg, 300 + 8 * it + rank)


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        m, ts, cfg = _build(world, 1)
        from mmidet_hip.ddp import GradReducer
        red = GradReducer(list(m.parameters()), bucket_mb=4 if world <= 2 else 1, comm='torch')        # several buckets: collectives start during backward
        assert len(red.buckets) > 1 and red.direct
        red.broadcast_parameters(m)
        red.broadcast_parameters(ts.ema.ema)
        ts.reducer = red
        losses = []
        for it in range(STEPS):
            loss, _ = ts.step(*_batch(cfg, it, rank))
            losses.append(float(loss.detach().sum()))
        torch.cuda.synchronize()
        sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items() if v.dtype.is_floating_point and 'running_' not in k}
        torch.save({'sd': sd, 'losses': losses}, os.path.join(out, 'rank%d.pt' % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_train_step_world2_on_one_gpu_matches_accumulated_single_process(world):
    """world = 4: bucket ordering, hooks and the cross-rank sequencing with more ranks than any pair test sees (a one-GPU box
    admits at most 6 processes on the card, this one included, so 8 ranks cannot be rehearsed here)."""
    port = 29700 + os.getpid() % 200
    ctx = mp.get_context('spawn')
    with tempfile.TemporaryDirectory() as out:
        procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(600)
            assert p.exitcode == 0, 'rank process failed (exit code %r)' % (p.exitcode,)
        got = [torch.load(os.path.join(out, 'rank%d.pt' % r)) for r in range(world)]
    for r in range(1, world):
        for k, v in got[0]['sd'].items():
            assert torch.equal(v, got[r]['sd'][k]), 'ranks 0 and %d disagree on %s after %d steps' % (r, k, STEPS)
    # single process, the ranks' batches as accumulated micro-batches per optimizer step
    m, ts, cfg = _build(1, world) if world <= 2 else _build_single(world)
    ref_losses = [[] for _ in range(world)]
    for it in range(STEPS):
        for r in range(world):
            loss, _ = ts.step(*_batch(cfg, it, r))
            ref_losses[r].append(float(loss.detach().sum()))
    torch.cuda.synchronize()
    from test_ops_gpu import close
    for r in range(world):                       # (the DDP ranks report the loss scaled by world_size, train.py:790)
        for a, b in zip(got[r]['losses'], ref_losses[r]):
            assert abs(a - world * b) <= (1e-5 if world == 2 else 1e-4) * abs(a), (r, a, b)   # (four addends: the all-reduce's order is not the accumulation's)
    ref = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    for k, v in got[0]['sd'].items():
        if v.numel():
            close(v, ref[k], tol=1e-5 if world == 2 else 2e-4, what=k)
