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


def _worker_direct(rank, world, port, ret):
    """TrainStep-shaped logic at world size 2: the DIRECT gradient form of the GPU path (ddp.GradReducer docstring) with its
    ops.GRAD_SLOTS / SLOT_HANDED_OUT bookkeeping, the weight-gradient "kernels" modelled by an autograd.Function that asks
    ops.grad_like() for its output and performs the write LATER, on a fake side stream (a queue flushed only when somebody
    waits for it) -- so a collective that is not ordered behind the side stream reduces poison."""
    sys.path.insert(0, PKG)
    sys.path.insert(0, REPO)
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from mmidet_hip import ops
    from mmidet_hip.ddp import GradReducer

    side = []                                       # the fake wgrad stream: closures that write a gradient slot

    class Lin(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            return x @ w.t()

        @staticmethod
        def backward(ctx, dy):
            x, w = ctx.saved_tensors
            dw = ops.grad_like(w)                   # the bucket view under the reducer, fresh memory otherwise
            # the "kernel" holds the raw address only (a second reference to `dw` itself would make autograd clone it)
            raw = torch.empty(0).set_(dw.untyped_storage(), dw.storage_offset(), dw.shape, dw.stride())
            side.append(lambda: raw.copy_(dy.t() @ x))
            if w.data_ptr() not in ops.GRAD_SLOTS:  # accumulation form: autograd reads dw right away, so the layer joins
                while side:                         # its side stream before returning (ops._join_side, DEFER_JOIN off)
                    side.pop(0)()
            return dy @ w, dw

    class Reducer(GradReducer):
        def _order_behind_writers_host(self, b):    # = comm.wait_stream(side) of the GPU path
            while side:
                side.pop(0)()

    torch.manual_seed(0)
    ws = [torch.nn.Parameter(torch.randn(16, 16) * 0.3) for _ in range(4)]
    bias = torch.nn.Parameter(torch.zeros(16))      # a small vector: produced by autograd itself, copied in by the hook

    def net(x):
        for w in ws:
            x = torch.tanh(Lin.apply(x, w))
        return x + bias

    def data(it):
        g = torch.Generator().manual_seed(1000 + 10 * it + rank)
        return torch.randn(8, 16, generator=g)

    def reference(its):
        """mean over ranks of the sum over `its` of the local gradients, by plain autograd"""
        out = []
        for p in ws + [bias]:
            p.grad = None
        for it in its:
            x = data(it)
            y = x
            for w in ws:
                y = torch.tanh(y @ w.t())
            (y + bias).square().mean().backward()
        for p in ws + [bias]:
            g = p.grad.clone()
            dist.all_reduce(g)
            out.append(g / world)
            p.grad = None
        return out

    want1, want2 = reference([0]), reference([1, 2])
    red = Reducer(ws + [bias], bucket_mb=0.0015, direct=True)       # 16x16 floats = 1 KB: several buckets
    assert red.direct and len(red.buckets) >= 2 and all(p.grad is None for p in ws)
    assert set(ops.GRAD_SLOTS) >= {w.data_ptr() for w in ws}
    # ---- one direct-mode step on poisoned buckets
    for b in red.buckets:
        b.flat.fill_(float('nan'))
    red.prepare()
    net(data(0)).square().mean().backward()
    while side:                                     # = ops.join_pending()
        side.pop(0)()
    red.finish()
    assert ops.SLOT_HANDED_OUT == {w.data_ptr() for w in ws}
    for p, g in zip(ws + [bias], want1):
        assert p.grad.data_ptr() == red._slot[p].data_ptr(), 'gradient must live in its bucket view'
        assert torch.allclose(p.grad, g, rtol=1e-5, atol=1e-7)
    # ---- a second backward without zero(): refused, not doubled
    try:
        red.prepare()
        raise AssertionError('stale .grad accepted in direct mode')
    except RuntimeError as e:
        assert 'direct mode cannot accumulate' in str(e)
    # ---- leftovers of a replayed graph (keep_grads) are dropped, then the step is right again
    red.zero(keep_grads=True)
    assert all(p.grad is not None for p in ws)
    red.prepare()
    assert all(p.grad is None for p in ws)
    net(data(0)).square().mean().backward()
    while side:
        side.pop(0)()
    red.finish()
    for p, g in zip(ws + [bias], want1):
        assert torch.allclose(p.grad, g, rtol=1e-5, atol=1e-7)
    red.zero()
    # ---- gradient accumulation over two backward passes: what TrainStep does when accumulate > 1
    red.set_direct(False)
    assert not ops.GRAD_SLOTS and all(p.grad is not None and float(p.grad.abs().max()) == 0 for p in ws)
    for it in (1, 2):
        red.prepare()
        net(data(it)).square().mean().backward()
        while side:
            side.pop(0)()
        red.finish()
    for p, g in zip(ws + [bias], want2):
        assert p.grad.data_ptr() == red._slot[p].data_ptr()
        assert torch.allclose(p.grad, g, rtol=1e-5, atol=1e-7), (p.grad - g).abs().max()
    ret[rank] = True
    dist.destroy_process_group()


def test_grad_reducer_direct_mode_world2_gloo():
    world = 2
    port = 31500 + os.getpid() % 2000
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker_direct, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert all(ret[r] for r in range(world))
