"""CPU, world_size 2 over gloo: the bucketed gradient reducer (mmidet_hip/ddp.py) gives every rank the mean of the
per-rank gradients, with parameter .grad tensors living inside the flat buckets (incl. channels_last conv weights)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, REPO


def _worker(rank, world, port, ret):
    sys.path.insert(0, PKG)
    sys.path.insert(0, REPO)
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from mmidet_hip.ddp import GradReducer
    torch.manual_seed(rank)                       # different init per rank: broadcast must fix it
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.Conv2d(8, 4, 1),
                              torch.nn.Flatten(), torch.nn.Linear(4 * 6 * 6, 5))
    net[0].weight.data = net[0].weight.data.contiguous(memory_format=torch.channels_last)
    red = GradReducer(list(net.parameters()), bucket_mb=0.0005)     # tiny buckets -> several collectives
    assert len(red.buckets) > 2
    red.broadcast_parameters(net)
    ref = [p.detach().clone() for p in net.parameters()]
    grads = []
    for it in range(2):                            # two steps: zero() must reset the flat buffers
        torch.manual_seed(100 + rank + 10 * it)
        x = torch.randn(4, 3, 6, 6)
        red.prepare()
        net(x).square().mean().backward()
        red.finish()
        grads.append([p.grad.clone() for p in net.parameters()])
        assert net[0].weight.grad.stride() == net[0].weight.stride()
        red.zero()
        assert all(float(p.grad.abs().max()) == 0 for p in net.parameters())
    # reference: plain autograd on each rank's data, averaged by all_reduce
    for it in range(2):
        torch.manual_seed(100 + rank + 10 * it)
        x = torch.randn(4, 3, 6, 6)
        for p in net.parameters():
            p.grad = None
        net(x).square().mean().backward()          # hooks still fire; they only count
        for g, p in zip(grads[it], net.parameters()):
            want = p.grad.clone()
            dist.all_reduce(want)
            want /= world
            assert torch.allclose(g, want, rtol=1e-5, atol=1e-7)
    ok = all(torch.equal(a, b) for a, b in zip(ref, [p.detach() for p in net.parameters()]))
    gathered = [None] * world
    dist.all_gather_object(gathered, [r.sum().item() for r in ref])
    assert gathered[0] == gathered[1], 'parameters differ across ranks after broadcast'
    ret[rank] = ok
    dist.destroy_process_group()


def test_grad_reducer_world2_gloo():
    world = 2
    port = 29500 + os.getpid() % 2000
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret[r] for r in range(world))
