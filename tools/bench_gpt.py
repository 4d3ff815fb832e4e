"""The fusion transformers (GPT1_fourier at P3, GPT at P4/P5) stand-alone at the bench shapes: forward and forward+backward
wall time per module, (a) eager with the host far ahead (back-to-back launches), (b) the same work replayed as a
hipGraph (no host in the loop), (c) the GEMM share (summed stand-alone times of its Linear layers are in
profiles/r01_conv_microbench_v5.txt).  These sections run with nothing else on the device (both lanes have joined), so
their wall time is step time."""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, 'mmi-det_amd')]
import bench  # noqa: E402
from mmidet_hip import fusion_ops as F2, ops  # noqa: E402
from models.common import GPT  # noqa: E402
from models.yolo_test import Model  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
GRAPH = len(sys.argv) > 2 and sys.argv[2] == 'graph'
cfg = bench.load_cfg('l_fourier')
dev = torch.device('cuda:0')
model = Model(cfg).to(dev).train()
if os.environ.get('BENCH_GPT_PACK', '1') != '0':
    F2.pack_qkv(model)              # what TrainStep does: q/k/v projections as one GEMM
for m in model.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.1
gpts = [(n, m) for n, m in model.named_modules() if isinstance(m, GPT)]


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def queued(fn, n=3):
    """fn's device time with its launches already queued behind a spin kernel (the host is not in the way, as in the
    steady-state step where it runs about half a step ahead)."""
    fn()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    torch.cuda._sleep(1000000)
    e[1].record()
    torch.cuda.synchronize()
    per_ms = 1000000 / e[0].elapsed_time(e[1])
    best = 1e9
    for _ in range(n):
        torch.cuda._sleep(int(per_ms * 40))
        e[1].record()
        fn()
        e[2].record()
        torch.cuda.synchronize()
        best = min(best, e[1].elapsed_time(e[2]))
    return best


print('%-28s %6s %9s | %9s %9s | %9s %9s %9s' % ('module', 'd', 'HxW', 'fwd eager', 'fwd queued', 'f+b eager', '1 stream', 'f+b queued'))
for name, g in gpts:
    d = g.n_embd
    hw = {256: 80, 512: 40, 1024: 20}.get(d, 40)
    rgb = torch.randn(bs, hw, hw, d, device=dev, requires_grad=True)
    ir = torch.randn(bs, hw, hw, d, device=dev, requires_grad=True)

    def out():
        o = g([rgb, ir])
        o = o[0] if isinstance(o, tuple) else o
        return o.maps[0], o.maps[1]

    def fwd():
        with torch.no_grad():
            out()

    go = None

    def fb():
        a, b = out()
        torch.autograd.backward([a, b], [go[0], go[1]])
        ops.join_pending()
        for p in g.parameters():
            p.grad = None
        rgb.grad = ir.grad = None

    a, b = out()
    go = (torch.randn_like(a), torch.randn_like(b))
    res = []
    for fn in (fwd, fb):
        eager = timed(fn)
        if fn is fb:
            ops.OVERLAP_WGRAD = False
            res.append(timed(fn))
            ops.OVERLAP_WGRAD = True
        try:
            if not GRAPH:
                raise RuntimeError('skipped')
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                fn()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            ops.OVERLAP_WGRAD = False     # nested stream forks do not survive hipStreamEndCapture on this ROCm
            with torch.cuda.graph(gr):
                fn()
            ops.OVERLAP_WGRAD = True
            graph = timed(gr.replay)
        except Exception as ex:  # noqa: BLE001
            if GRAPH:
                print('graph capture failed:', str(ex)[:200])
            graph = queued(fn)
        res = res[:-1] + [eager, res[-1], graph] if fn is fb else res + [eager, graph]
    print('%-28s %6d %4dx%-4d | %7.3f ms %7.3f ms | %7.3f ms %7.3f ms %7.3f ms' % (name, d, hw, hw, res[0], res[1], res[2], res[3], res[4]), flush=True)
