"""bench.py - paired RGB+IR img/s of one full training step of the two-stream MMI-Det hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload l_fourier|s_add|s_fourier] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = /255 + RGB/IR split -> forward (twin CSPDarknet + CEM/FFM/GPT fusion + PANet + Detect) -> ComputeLoss ->
backward -> [bucketed RCCL gradient all-reduce on a side stream] -> SGD(nesterov) step -> EMA update, on a synthetic
uint8 (B,6,640,640) batch already resident in HBM.  Default workload = BASELINE.json configs[2]:
yolov5l two-stream-fourier, bs=16/GPU, 640x640, nc=6, dropout p=0.1, fp32 MFMA.  Weak scaling: B per GPU is fixed.

Prints ONE JSON line on rank 0 (metric/value/... + "roofline" for the dominant kernel family, the fp32-MFMA implicit
GEMM, timed live with HIP events on its launch stream + "cpu_baseline": the oracle timed on the host cores).
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, 'mmi-det_amd'))
sys.path.insert(0, REPO)

CFG_DIR = os.path.join(REPO, 'mmi-det_amd', 'models', 'transformer')
WORKLOADS = {
    # name: (yaml, depth, width, ffm_channels, nc, default batch, fwd GFLOP per paired image)
    'l_fourier': ('yolov5l_fusion_transformer_M3FD_fuse3_fourier.yaml', 1.0, 1.0, 128, 6, 16, 230.2),
    's_fourier': ('yolov5l_fusion_transformer_M3FD_fuse3_fourier.yaml', 0.33, 0.50, 64, 6, 16, 41.6),
    's_add': ('yolov5s_fusion_add_vedai.yaml', 0.33, 0.50, None, 9, 8, 32.5),
    # BASELINE.json configs[4]: yolov5x (depth 1.33, width 1.25, FFM 160), 1280x1280, bs=8/GPU -- the HBM-heavy regime
    'x_1280': ('yolov5l_fusion_transformer_M3FD_fuse3_fourier.yaml', 1.33, 1.25, 160, 6, 8, 1519.8),
}
IMAGE_SIZE = {'x_1280': 1280}
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD


def load_cfg(name):
    import yaml
    fn, gd, gw, ffm, nc, _, _ = WORKLOADS[name]
    with open(os.path.join(CFG_DIR, fn)) as f:
        d = yaml.safe_load(f)
    d['depth_multiple'], d['width_multiple'], d['nc'] = gd, gw, nc
    if ffm is not None:
        d['backbone'][6][3] = [ffm]     # the FFM width is not scaled by width_multiple in the reference (quirk B3)
    return d


def synth(bs, size, nc, device, seed):
    """SURVEY.md §8d synthetic batch: uint8 (B,6,S,S); 8 targets per image."""
    g = torch.Generator().manual_seed(seed)
    imgs = torch.randint(0, 256, (bs, 6, size, size), dtype=torch.uint8, generator=g)
    t = 8
    tg = torch.zeros(bs * t, 6)
    tg[:, 0] = torch.arange(bs).repeat_interleave(t)
    tg[:, 1] = torch.randint(0, nc, (bs * t,), generator=g).float()
    tg[:, 2:4] = 0.1 + 0.8 * torch.rand(bs * t, 2, generator=g)
    tg[:, 4:6] = 0.02 + 0.30 * torch.rand(bs * t, 2, generator=g)
    return imgs.to(device), tg.to(device)


class ConvTimer:
    """HIP events around every implicit-GEMM launch (conv fwd / dgrad / wgrad incl. its split-K reduce), recorded on the
    stream the kernel is launched on (the last argument of every entry point; wgrad runs on a side stream next to
    dgrad).  Events are asynchronous: nothing is serialised, the timed region stays the timed region.  Because launches on
    the two streams overlap, the family's busy time is the UNION of the per-launch [start, end] intervals (all measured
    against one reference event), not their sum."""

    def __init__(self):
        self.recs = []
        self.shapes = []
        self.on = False
        self.ref = None
        self._streams = {}

    def _stream(self, handle):
        st = self._streams.get(handle)
        if st is None:
            cur = torch.cuda.current_stream()
            st = cur if cur.cuda_stream == handle else torch.cuda.ExternalStream(handle)
            self._streams[handle] = st
        return st

    def start(self):
        self.ref = torch.cuda.Event(enable_timing=True)
        self.ref.record()
        self.on = True

    def install(self):
        from mmidet_hip import lib
        timer = self

        def wrap(name, flops_of, label=None):
            fn = getattr(lib, name)
            attr, name = name, label or name

            def timed(*a):
                if not timer.on:
                    return fn(*a)
                st = timer._stream(a[-1])
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(st)
                r = fn(*a)
                e.record(st)
                timer.recs.append((name, flops_of(a), s, e))
                dd = [x for x in a if hasattr(x, 'Cout')][0]
                timer.shapes.append((dd.N, dd.H, dd.W, dd.Cin, dd.Cout, dd.KH, dd.stride))
                return r
            setattr(lib, attr, timed)

        def fl(d):
            return 2.0 * d.N * d.Ho * d.Wo * d.Cout * d.Cin * d.KH * d.KW
        def desc_of(a):
            return [x for x in a if hasattr(x, 'Cout')][0]
        wrap('conv_fwd', lambda a: fl(desc_of(a)))
        wrap('conv_bn_fwd', lambda a: fl(desc_of(a)), 'conv_fwd')        # (training Conv: statistics folded in the same launch)
        wrap('conv_dgrad', lambda a: fl(desc_of(a)))
        wrap('conv_wgrad', lambda a: fl(desc_of(a)))
        wrap('conv_wgrad_tab', lambda a: fl(desc_of(a)), 'conv_wgrad')
        # twin launches (both backbones' copies of a layer in one launch: mmidet_hip/twin_ops.py): two problems of shape d
        wrap('conv_bn_fwd2', lambda a: 2 * fl(desc_of(a)), 'conv_fwd')
        wrap('conv_dgrad2', lambda a: 2 * fl(desc_of(a)), 'conv_dgrad')
        wrap('conv_wgrad2', lambda a: 2 * fl(desc_of(a)), 'conv_wgrad')
        # the transformer blocks' Linear layers with fused epilogues (same kernels, counted with the convolutions)
        wrap('linear_fwd_fused', lambda a: fl(desc_of(a)), 'conv_fwd')
        wrap('linear_dgrad_fused', lambda a: fl(desc_of(a)), 'conv_dgrad')

    def by_shape(self):
        """[(entry point, shape) -> launches, ms, TFLOP/s] sorted by time: where the GEMM time goes."""
        agg = {}
        for (name, f, s, e), shp in zip(self.recs, self.shapes):
            v = agg.setdefault((name, shp), [0, 0.0, 0.0])
            v[0] += 1
            v[1] += s.elapsed_time(e)
            v[2] += f
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
        return ['%-11s N%-5d %4dx%-4d %5d->%-5d k%d s%d  x%-4d %8.2f ms  %6.1f TF' %
                (k[0], *k[1], v[0], v[1], v[2] / (v[1] * 1e-3) / 1e12) for k, v in rows]

    def summary(self):
        """-> (total flop, union-of-intervals busy ms, {entry point: [flop, sum of launch ms, launches]})"""
        tot_f = 0.0
        per = {}
        iv = []
        for name, f, s, e in self.recs:
            t0, t1 = self.ref.elapsed_time(s), self.ref.elapsed_time(e)
            iv.append((t0, t1))
            tot_f += f
            p = per.setdefault(name, [0.0, 0.0, 0])
            p[0] += f
            p[1] += t1 - t0
            p[2] += 1
        iv.sort()
        busy, cur0, cur1 = 0.0, None, None
        for t0, t1 in iv:
            if cur1 is None or t0 > cur1:
                if cur1 is not None:
                    busy += cur1 - cur0
                cur0, cur1 = t0, t1
            else:
                cur1 = max(cur1, t1)
        if cur1 is not None:
            busy += cur1 - cur0
        return tot_f, busy, per


def hbm_ops(model, bs, size, dev):
    """The memory-bound fusion ops of the path (SURVEY.md §8a rows 3, 7, 8, 10, 12, 13 and the BatchNorm passes) timed stand-alone
    with HIP events on the current stream at this workload's shapes: achieved GB/s of ALGORITHMIC traffic (every input read
    once, every output written once) against the 8 TB/s HBM peak of MI355X_MICROARCH.md."""
    from mmidet_hip import fusion_ops as F2, ops
    c2 = model.model[2].cv3.conv.weight.shape[0] if hasattr(model.model[2], 'cv3') else 128      # P2 width (128 at yolov5l)
    h2 = size // 4
    out = {}

    def timed(fn, n=10):
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e-3

    def row(name, nbytes, fn):
        t = timed(fn)
        out[name] = {'algorithmic_MB': round(nbytes / 1e6, 1), 'us': round(t * 1e6, 1), 'GB/s': round(nbytes / t / 1e9, 0),
                     'frac_of_8TB/s': round(nbytes / t / 8e12, 3)}
    with torch.no_grad():
        a = torch.randn(bs, h2, h2, c2, device=dev)
        b = torch.randn(bs, h2, h2, c2, device=dev)
        nb = a.numel() * 4
        tok = F2.pool_tokens(a, b)
        row('avgpool8 P2 (both streams -> tokens)', 2 * nb + tok.numel() * 4, lambda: F2.pool_tokens(a, b))
        t1, _ = F2.split_tokens(tok)
        row('bilinear 8x8 upsample + Add2 P2', 2 * nb + t1.numel() * 4, lambda: F2.upsample_add(a, t1))
        row('CBM + IGM statistics P2', 2 * nb, lambda: F2.fusion_stats(a, b, tok))
        a3, b3 = a[:, :h2 // 2].contiguous(), b[:, :h2 // 2].contiguous()
        row('Add (rgb + ir) half a T2', 3 * nb // 2, lambda: ops.add(a3, b3))
        x5 = torch.randn(bs, size // 32, size // 32, 4 * c2, device=dev)
        row('SPP 5/9/13 pools -> concat buffer', x5.numel() * 4 * 5, lambda: ops.spp_pool(x5))
        img = torch.rand(bs, size, size, 3, device=dev)
        was = model.Enhance.training
        model.Enhance.eval()                     # (eval: the timing must not move the BatchNorm running statistics)
        row('CEM forward (3 reads + 1 write of x; 24-ch intermediates are the excess)', 4 * img.numel() * 4, lambda: model.Enhance(img))
        model.Enhance.train(was)
    if was and any(p.requires_grad for p in model.Enhance.parameters()):
        # the TRAINING forward writes what the backward reads (y2, t: 24 channels each; the channel-sum map) besides y3 and the output, and
        # moves the BatchNorm running statistics: restored afterwards
        keep = {k: v.clone() for k, v in model.Enhance.state_dict().items() if 'running' in k or 'tracked' in k}
        px = img.numel() // 3
        # (algorithmic: x read; y2, t, chansum, y3 written; BatchNorm3 + residual pass: y3, x read, out written -- the second launch's
        #  re-read of y2 is the implementation's, not the algorithm's)
        row('CEM training forward (y2, t, chansum kept for the backward; y3, out written)', px * 4 * (3 + 24 + 24 + 1 + 3 + 3 + 3 + 3),
            lambda: model.Enhance(img))
        model.Enhance.load_state_dict({**model.Enhance.state_dict(), **keep})
    with torch.no_grad():
        mi = torch.cat([torch.zeros(c2, device=dev), torch.ones(c2, device=dev)])
        g1, b1 = torch.ones(c2, device=dev), torch.zeros(c2, device=dev)
        o = torch.empty_like(a)
        from mmidet_hip import lib
        st = torch.cuda.current_stream().cuda_stream
        rows = a.numel() // c2
        row('BatchNorm normalise + SiLU (T2-size)', 2 * nb, lambda: lib.bn_act_fwd(a.data_ptr(), c2, mi.data_ptr(), g1.data_ptr(), b1.data_ptr(),
                                                                                     None, 0, o.data_ptr(), c2, rows, c2, 1, st))
        nbw = lib.bn_act_bwd_workspace(rows, c2)
        ws = torch.zeros(nbw, dtype=torch.uint8, device=dev)
        dg, db = torch.empty(c2, device=dev), torch.empty(c2, device=dev)
        row('BatchNorm backward, both passes (T2-size; 4 reads + 1 write)', 5 * nb, lambda: lib.bn_act_bwd(
            a.data_ptr(), c2, b.data_ptr(), c2, None, 0, c2, mi.data_ptr(), g1.data_ptr(), b1.data_ptr(), ws.data_ptr(), nbw, o.data_ptr(), c2,
            dg.data_ptr(), db.data_ptr(), None, None, rows, c2, 1, 0, st))
    return out


def host_cores():
    """CPU share of this process: affinity, capped by the cgroup quota (a 1-GPU box grants 16 cores)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(workload, seconds_budget=25.0):
    """The oracle (pure-torch CPU restatement, proven equal to the reference on the golden fixtures) timed on this box's
    host cores on a bounded sample of the same workload: same graph, same step definition, B=2."""
    from oracle.ref_loss import ComputeLoss as OLoss, scaled_hyp
    from oracle.ref_model import Model as OModel
    cfg = load_cfg(workload)
    nc = cfg['nc']
    cores = host_cores()
    torch.set_num_threads(cores)
    m = OModel(cfg)
    m.nc, m.gr, m.hyp = nc, 1.0, scaled_hyp(nc, 640)
    lf = OLoss(m)
    opt = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=0.01, momentum=0.937, nesterov=True)
    bs = 2
    imgs, tg = synth(bs, 640, nc, 'cpu', 1)
    m.train()

    def one():
        x = imgs.float() / 255
        pred, comb = m(x[:, :3], x[:, 3:])
        loss, _ = lf(pred, tg, comb.reshape(-1))
        loss.sum().backward()
        opt.step()
        opt.zero_grad()
    one()
    print('[bench] cpu_baseline: warm-up step done on %d threads' % cores, file=sys.stderr, flush=True)
    t0 = time.time()
    n = 0
    while True:
        one()
        n += 1
        print('[bench] cpu_baseline: step %d, %.1f s' % (n, time.time() - t0), file=sys.stderr, flush=True)
        if time.time() - t0 > seconds_budget or n >= 5:
            break
    dt = (time.time() - t0) / n
    return {'value': round(bs / dt, 4), 'unit': 'paired img/s', 'cores': cores, 'kind': 'port', 'batch': bs,
            'sample': '%d timed steps of the %s graph at B=%d 640x640 (fwd+loss+bwd+SGD, fp32, torch CPU oracle), %.1f s/step'
                      % (n, workload, bs, dt)}


def _cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.strip().split(','):
        if not part:
            continue
        a, _, b = part.partition('-')
        out.extend(range(int(a), int(b or a) + 1))
    return out


def plan_rank_cores(local_rank, local_world, allowed, sysfs='/sys'):
    """Cores for one rank of `local_world` ranks on this node, out of the `allowed` ones.  Ranks whose GPU reports a NUMA node
    (sysfs: the AMD display / accelerator PCI functions in bus order = HIP's device order) share that node's cores among the ranks
    of the node, each an equal contiguous block; without that information the allowed cores are cut into equal contiguous blocks
    in rank order.  Reads sysfs text files only -- nothing here touches the GPU."""
    allowed = sorted(allowed)
    per = len(allowed) // max(local_world, 1)
    fallback = allowed[local_rank * per:(local_rank + 1) * per] if per >= 1 else []
    try:
        gpus = []
        base = os.path.join(sysfs, 'bus', 'pci', 'devices')
        for dev in sorted(os.listdir(base)):
            def rd(name, dev=dev):
                with open(os.path.join(base, dev, name)) as f:
                    return f.read().strip()
            try:
                if rd('vendor') != '0x1002' or rd('class')[:4] not in ('0x03', '0x12'):   # display controller / processing accelerator
                    continue
                gpus.append(int(rd('numa_node')))
            except (OSError, ValueError):
                continue
        vis = os.environ.get('HIP_VISIBLE_DEVICES') or os.environ.get('CUDA_VISIBLE_DEVICES') or os.environ.get('ROCR_VISIBLE_DEVICES')
        if vis:                                           # a visible-device list renumbers the GPUs: follow it when it is numeric
            try:
                gpus = [gpus[int(v)] for v in vis.split(',') if v.strip() != '']
            except (ValueError, IndexError):
                return fallback
        if len(gpus) < local_world or gpus[local_rank] < 0:
            return fallback
        node = gpus[local_rank]
        with open(os.path.join(sysfs, 'devices', 'system', 'node', 'node%d' % node, 'cpulist')) as f:
            cores = [c for c in _cpulist(f.read()) if c in set(allowed)]
        mates = [r for r in range(local_world) if gpus[r] == node]       # the ranks that share this node's cores
        share = len(cores) // len(mates)
        if share < 1:
            return fallback
        k = mates.index(local_rank)
        return cores[k * share:(k + 1) * share]
    except (OSError, ValueError, IndexError):
        return fallback


def pin_rank_to_cores(local_rank, local_world, world):
    """-> 'first-last (n cores)' description of the set this rank was pinned to, or None when nothing was changed."""
    want = os.environ.get('MMIDET_PIN', '1' if world > 1 else '0')
    if want == '0' or not hasattr(os, 'sched_setaffinity'):
        return None
    try:
        mine = plan_rank_cores(local_rank, local_world, os.sched_getaffinity(0))
        if not mine:
            return None
        os.sched_setaffinity(0, mine)
        return '%d-%d (%d cores)' % (mine[0], mine[-1], len(mine))
    except OSError:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='l_fourier', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=None, help='paired images per GPU')
    ap.add_argument('--dropout', type=float, default=0.1)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--dump-gemm', default=None, help='write the per-shape GEMM time table of the roofline pass here')
    ap.add_argument('--ddp', action='store_true', help='run the data-parallel code path even on one GPU (world size 1)')
    ap.add_argument('--gemm', default='fp32', choices=['fp32', 'bf16x9', 'bf16x6', 'bf16x3', 'exact_auto'],
                    help='arithmetic of the conv/linear GEMMs in the TIMED region.  fp32 (default, the headline): exact fp32 '
                         'products on v_mfma_f32_32x32x2_f32.  bf16x6: every operand split into three bf16 terms, six bf16 MFMA '
                         'products of total order <= 2, fp32 accumulation (GEMM 3e-7..1e-6 vs the fp32 kernels).  bf16x9: all nine '
                         'products (each product exact).  Both: full-depth gradients as the fp32 path over four seeds. '
                         'bf16x3: two terms, three products (GEMM 4.5e-6, full-depth gradients ~1e-2)')
    ap.add_argument('--no-split-probe', action='store_true', help='skip the extra split-bf16 measurement after the timed region')
    ap.add_argument('--storage', default='f32', choices=['f32', 'bf16'],
                    help='activation storage.  f32 (default, the headline: fp32 parity).  bf16: OPT-IN AMP-like mode (SURVEY.md §8 f-4) -- '
                         'bf16 maps in HBM, one bf16 MFMA product per element pair, fp32 accumulation / weights / statistics / loss')
    ap.add_argument('--h2d', action='store_true', help='also time the step fed from HOST memory through the pinned, double-buffered '
                                                       'feeder (mmidet_hip.feed): the PCIe-inclusive rate, reported beside `value`')
    ap.add_argument('--mode', default='auto', choices=['auto', 'graph', 'eager'], help='launch mode (see main)')
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of the captured whole-step hipGraph')
    ap.add_argument('--spawn', action='store_true', help='start the rank(s) through torch.distributed.run even for one GPU: rehearses '
                                                         'the launcher path `--gpus N` takes for N > 1 (tests/test_bench_contract_gpu.py)')
    args = ap.parse_args()
    if args.storage == 'bf16':          # (the roofline pass and the split-bf16 probe describe the fp32 kernels)
        args.no_roofline = args.no_split_probe = True

    if (args.gpus > 1 or args.spawn) and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: start one rank per GPU ourselves.  This parent never touches the GPU
        # (nothing above initialises HIP); the ranks are fresh child processes and rank 0's JSON line is relayed as is.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != '--spawn']
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        env.setdefault('OMP_NUM_THREADS', '2')
        raise SystemExit(subprocess.call(cmd, env=env))

    t_start = time.perf_counter()
    # The contract is ONE JSON line on stdout.  RCCL prints a version banner to fd 1 from native code, so the real stdout
    # is parked on a private descriptor and fd 1 points at stderr until the result is written.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    assert world == args.gpus, '--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)' % (args.gpus, world)
    # One core set per rank, taken BEFORE anything touches the GPU (so that the HIP runtime's own threads inherit it): N Python
    # threads each enqueue ~3000 launches per step, and the unpinned ones migrate across the box and onto each other's cores.
    # The ranks of this node split the cores this process may use into equal contiguous blocks, rank i gets block i.
    # MMIDET_PIN=0 leaves the affinity alone; MMIDET_PIN=1 pins a single rank too (to its whole set: a no-op, reported).
    cpu_set = pin_rank_to_cores(local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', world)), world)
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (there is no CPU fallback for the product path)'
    # MMIDET_COMM=gloo: REHEARSAL transport (tests/test_bench_contract_gpu.py): every rank on GPU 0, rendezvous and the bucket
    # collectives over gloo (which carries device tensors through the host; RCCL refuses two ranks on one device) -- the whole
    # N > 1 code path of this file and of TrainStep / GradReducer on a one-GPU box.  Its numbers mean nothing.
    rehearsal = os.environ.get('MMIDET_COMM', 'torch') == 'gloo'
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    ddp = world > 1 or args.ddp
    # 'torch' (default): torch.distributed / ProcessGroupNCCL = RCCL.  'native': the library's own communicator (mmi_comm_*: RCCL
    # called directly on the reducer's HIP stream, collectives capturable into the step graph, no watchdog thread).  Both run the
    # same bucket logic and, re-measured interleaved at world size 1 in round 3, the same step time (native 123.2 / 123.3 ms, torch
    # 123.0 / 123.4 ms: profiles/r03_ab_comm_world1.txt; round 2's "+3 ms" was the reducer's copy path, fixed since).  Neither has
    # run at N > 1 on this pool (one-GPU boxes); the default stays the transport every other PyTorch job on a node exercises,
    # MMIDET_COMM=native selects the other one.
    comm_kind = os.environ.get('MMIDET_COMM', 'torch')
    if ddp:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        if comm_kind == 'gloo':
            dist.init_process_group('gloo', rank=rank, world_size=world)
        elif comm_kind == 'native':
            # control plane (barriers, the max over ranks of a few floats, the 128-byte RCCL id) on gloo; the gradients go
            # through the library's own communicator on the reducer's HIP stream: no ProcessGroupNCCL, no watchdog thread
            from mmidet_hip.ddp import init_native_comm
            dist.init_process_group('gloo', rank=rank, world_size=world)
            try:
                init_native_comm(rank, world)
            except Exception as e:      # (RCCL could not be bound or initialised: the same on every rank -> fall back together)
                print('[bench] native RCCL transport unavailable (%r); falling back to torch.distributed/nccl' % (e,), file=sys.stderr, flush=True)
                dist.destroy_process_group()
                comm_kind = 'torch'
                dist.init_process_group('nccl', rank=rank, world_size=world)
        else:
            # (no device_id=: binding the group to the device at init makes every later step ~5 ms slower on this
            # torch/RCCL; torch.cuda.set_device above already pins the rank to its GPU)
            dist.init_process_group('nccl', rank=rank, world_size=world)
    # tools/ab_comm3.sh (world 1, no reducer, the plain single-GPU step): which part of a transport's SET-UP changes the step time
    # -- 'gloo' = a gloo process group exists, 'rccl' = the library's communicator exists, 'nccl' = a ProcessGroupNCCL that has
    # run one collective exists; combinations with '+'
    bisect = os.environ.get('MMIDET_COMM_BISECT', '') if not ddp else ''
    if bisect:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        parts = bisect.split('+')
        if 'gloo' in parts:
            dist.init_process_group('gloo', rank=0, world_size=1)
        elif 'nccl' in parts:
            dist.init_process_group('nccl', rank=0, world_size=1)
            dist.barrier()
        if 'rccl' in parts:
            from mmidet_hip.ddp import init_native_comm
            init_native_comm(0, 1)
    ctl_dev = dev if (ddp and comm_kind not in ('native', 'gloo')) else torch.device('cpu')   # where control-plane tensors live

    from mmidet_hip.train_step import TrainStep
    from models.yolo_test import Model
    torch.manual_seed(2 + rank)                                           # train.py: init_seeds(2 + rank)
    cfg = load_cfg(args.workload)
    nc = cfg['nc']
    bs = args.batch or WORKLOADS[args.workload][5]
    model = Model(cfg).to(dev)
    model.storage = args.storage
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = args.dropout
    model.train()
    # Launch mode.  "eager": every kernel enqueued by the host each step (~120-135 ms of host time per step, so the GPU
    # stays the bottleneck as long as a core is free), wgrad on a side stream next to dgrad.  "graph": hipGraph replay --
    # one GPU: the whole step is one graph; data parallel: forward+backward is the graph, then the bucket all-reduces
    # (RCCL) and the one-launch optimizer follow on the same stream.  On this ROCm a replay costs about as much host time
    # as the eager enqueue and leaves larger gaps between dependent kernels, so neither mode wins everywhere: "auto"
    # (default) times three warm-up steps of each and keeps the faster one for the timed region.
    mode = 'eager' if args.no_graph else args.mode
    use_graph = False
    size = IMAGE_SIZE.get(args.workload, 640)
    ts = TrainStep(model, nc, size, bs, world_size=world, accumulate=1, graph=False)
    if ddp:
        from mmidet_hip.ddp import GradReducer
        red = GradReducer(list(model.parameters()), comm='torch') if comm_kind == 'gloo' else GradReducer(list(model.parameters()))
        red.broadcast_parameters(model)
        if ts.ema is not None:
            red.broadcast_parameters(ts.ema.ema)      # (the EMA copy was taken from the rank-local initialisation)
        ts.reducer = red
    imgs, tg = synth(bs, size, nc, dev, 100 + rank)

    timer = ConvTimer()
    if not args.no_roofline:
        timer.install()          # inert until timer.start(); events cannot be recorded inside a graph replay

    def barrier():
        if ddp:
            dist.barrier()
        torch.cuda.synchronize()

    def note(msg):
        if rank == 0:
            print('[bench %.1fs] %s' % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    from mmidet_hip import lib as _lib
    # bf16 storage: the GEMMs whose operands stay fp32 in HBM (token-side Linear layers, Focus, Detect heads) take the same
    # single-product bf16 arithmetic (mode 5 = "bf16x1")
    _lib.set_gemm_precision(5 if args.storage == 'bf16' else {'fp32': 0, 'bf16x3': 1, 'bf16x6': 2, 'bf16x9': 3, 'exact_auto': 6}[args.gemm])
    note('model on device, %d params; warmup' % sum(p.numel() for p in model.parameters()))
    for i in range(args.warmup):
        ts.step(imgs, tg)
        torch.cuda.synchronize()
        note('warmup step %d done, %.1f GB allocated' % (i, torch.cuda.max_memory_allocated() / 2 ** 30))

    def probe(n=3):
        """-> (ms per step, host ms per step spent enqueueing), max over ranks"""
        barrier()
        t = time.perf_counter()
        for _ in range(n):
            ts.step(imgs, tg)
        te = time.perf_counter() - t
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t, te], device=ctl_dev, dtype=torch.float64)
        if ddp:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)      # every rank takes the same decision
        return float(t[0]) / n * 1e3, float(t[1]) / n * 1e3

    probes = {}
    if mode == 'auto':
        # Eager is kept when the host finishes enqueueing a step well before the GPU finishes running it (then the GPU
        # never waits for the host); the graph is only captured -- 25 GB of private pool, thousands of retained tensors
        # that measurably slow later eager steps -- when that is not the case.
        ms, enq = probe()
        probes['eager'] = round(ms, 1)
        probes['eager_host_enqueue'] = round(enq, 1)
        # Backbone launch form per workload: twin launches (both backbones' layer pairs in one grid) win where launches or the
        # host bound the step, the two-lane form (RGB / IR backbones on two HIP streams) where the other lane's GEMM hides this
        # lane's BatchNorm passes (yolov5x at 1280: ~1 %).  Probed like the launch mode, but only where the GPU is the bound
        # (a host-bound step's timing jitters by more than the forms differ, and twin launches halve the host's work), and a
        # switch needs a 1.5 % margin.
        if (getattr(model, 'twin', False) and args.storage == 'f32' and enq < 0.8 * ms and not ddp      # (one GPU only: the N > 1 path keeps twin launches)
                and os.environ.get('MMIDET_FORM_PROBE', '1') != '0'):
            model.twin = False
            ts.step(imgs, tg)                              # first lane-form step: its streams, workspaces and pixel tables
            ms_l, enq_l = probe()
            probes['eager_lanes'] = round(ms_l, 1)
            if ms_l < 0.985 * ms:
                ms, enq = ms_l, enq_l
                probes['eager_host_enqueue'] = round(enq, 1)
            else:
                model.twin = True
        if enq > 0.92 * ms:
            ts.use_graph = True
            ts.step(imgs, tg)                             # capture (runs two eager steps on a side stream first)
            torch.cuda.synchronize()
            probes['graph'] = round(probe()[0], 1)
            ts.use_graph = probes['graph'] < ms
        note('probe: %s -> %s' % (probes, 'graph' if ts.use_graph else 'eager'))
    elif mode == 'graph':
        ts.use_graph = True
        ts.step(imgs, tg)
        torch.cuda.synchronize()
    use_graph = ts.use_graph
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, items = ts.step(imgs, tg)
    t_enq = time.perf_counter() - t0      # host time to enqueue the K steps (the GPU may still be running)
    barrier()
    dt = time.perf_counter() - t0
    enq_ranks = [t_enq / args.steps * 1e3]
    if world > 1:                         # host enqueue per step of every rank: eight Python threads share the box's cores
        te = torch.tensor([t_enq / args.steps * 1e3], device=ctl_dev, dtype=torch.float64)
        allte = [torch.zeros_like(te) for _ in range(world)]
        dist.all_gather(allte, te)
        enq_ranks = [float(x) for x in allte]
    # what the driver's SCALE record can check the line against: the ranks that really took part, the RCCL build, the core sets
    ranks_seen, rccl_version, cpu_sets = world if not ddp else dist.get_world_size(), None, [cpu_set]
    if ddp:
        if comm_kind == 'native':
            from mmidet_hip import lib as _l0
            ranks_seen = min(ranks_seen, _l0.comm_world())
        try:
            rccl_version = '.'.join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            rccl_version = None
        if world > 1:
            cpu_sets = [None] * world
            dist.all_gather_object(cpu_sets, cpu_set)
    timer.on = False
    roof_steps = args.steps
    if not args.no_roofline:
        # Roofline pass: the SAME step issued eagerly right after the timed region (same kernels, shapes, streams) with HIP
        # events around every implicit-GEMM launch; kept out of the timed region because a replayed graph offers no
        # per-kernel event points and, in eager mode, 2 x 1100 event records per step are host time the step does not have.
        ts.use_graph = False
        roof_steps = min(args.steps, 3)
        ts.step(imgs, tg)
        torch.cuda.synchronize()
        timer.start()
        for _ in range(roof_steps):
            ts.step(imgs, tg)
        torch.cuda.synchronize()
        timer.on = False
    split = {}
    if args.gemm == 'fp32' and not args.no_split_probe:
        # Not the headline: the same step with the opt-in split-bf16 GEMM arithmetics, reported beside it.
        ts.use_graph = False
        for name, mode in (('exact_auto', 6), ('bf16x9', 3), ('bf16x6', 2), ('bf16x3', 1)):
            _lib.set_gemm_precision(mode)
            for _ in range(2):
                ts.step(imgs, tg)
            barrier()
            t1 = time.perf_counter()
            n_split = min(args.steps, 5)
            for _ in range(n_split):
                loss_s, _ = ts.step(imgs, tg)
            barrier()
            tt = torch.tensor([time.perf_counter() - t1], device=ctl_dev, dtype=torch.float64)
            if ddp:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            # (an optional probe never fails the run: a non-finite loss is reported in its own object instead)
            split[name] = (float(tt) / n_split, bool(torch.isfinite(loss_s).all()), float(loss_s.detach().sum()))
        _lib.set_gemm_precision(0)
    pcie = None
    if args.h2d:
        # Not `value` (whose inputs are resident in HBM by contract): the same step with every batch coming from host memory.
        from mmidet_hip.feed import PairedBatchFeeder
        ts.use_graph = False
        host = [(imgs.cpu().pin_memory(), tg.cpu()), (imgs.flip(0).cpu().pin_memory(), tg.cpu())]   # as a pin_memory DataLoader
        n_h2d = args.steps + 2

        def gen():
            for i in range(n_h2d):
                yield host[i & 1]
        feeder = PairedBatchFeeder(gen(), dev)
        barrier()
        t1 = None
        for i, (bi, bt) in enumerate(feeder):
            if i == 2:                       # two untimed steps fill the pipeline
                barrier()
                t1 = time.perf_counter()
            ts.step(bi, bt)
        barrier()
        th = (time.perf_counter() - t1) / args.steps
        pcie = {'value': round(world * bs / th, 3), 'unit': 'paired img/s', 'ms_per_step': round(th * 1e3, 3),
                'h2d_MB_per_step': round(feeder.bytes_copied / n_h2d / 1e6, 2),
                'what': 'same step, every batch copied from pinned host memory as uint8 by mmidet_hip.feed.PairedBatchFeeder '
                        '(double-buffered on a copy stream, overlapped with the previous step)'}
    if world > 1:
        t = torch.tensor([dt], device=ctl_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    assert torch.isfinite(loss).all(), 'non-finite loss in the timed region'

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * bs * args.steps / dt
        out = {
            'metric': 'paired RGB+IR img/s (train step, %dx%d %s two-stream)' % (size, size, 'yolov5x' if args.workload == 'x_1280' else 'yolov5s' if args.workload.startswith('s_') else 'yolov5l'), 'value': round(value, 3),
            'unit': 'paired img/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16 storage + bf16 MFMA, fp32 accumulation / weights / BatchNorm statistics / loss (opt-in, NOT the parity headline)' if args.storage == 'bf16' else {'fp32': 'f32', 'bf16x9': 'f32 storage/accumulation, exact products as 9 bf16 MFMAs of a 3-term split',
                      'bf16x6': 'f32 storage/accumulation, products as 6 bf16 MFMAs of a 3-term split',
                      'bf16x3': 'f32 storage/accumulation, products as 3 bf16 MFMAs of a 2-term split',
                      'exact_auto': 'f32 storage/accumulation, every product exact: fp32 MFMA or 9 bf16 MFMAs of a 3-term split, chosen per layer shape'}[args.gemm],
            'data': 'synthetic',
            'config': {'workload': {'l_fourier': 'yolov5l two-stream-fourier (default YAML, CEM+FFM+3xGPT), nc=6',
                                    's_fourier': 'yolov5s two-stream-fourier (GPT1_fourier[64]), nc=6',
                                    's_add': 'yolov5s fusion_add (fusion modules off), nc=9',
                                    'x_1280': 'yolov5x two-stream-fourier (GPT1_fourier[160]), nc=6, 1280x1280'}[args.workload],
                       'batch_per_gpu': bs, 'global_batch': bs * world, 'image': '2x(3,%d,%d)' % (size, size), 'dropout_p': args.dropout,
                       'step': 'fwd+loss+bwd+allreduce+SGD(nesterov)+EMA', 'parallelism': 'dp%d' % world,
                       'launch_mode_probe_ms': probes,
                       'backbone_launch_form': ('twin launches (both backbones\' layer pairs in one grid)' if getattr(model, 'twin', False) and args.storage == 'f32'
                                                else 'two lanes (RGB / IR backbone on two HIP streams)'),
                       'launch_mode': ('eager, wgrad on a side stream' if not use_graph else 'whole-step hipGraph replay' if not ddp else
                                       'whole-step hipGraph replay incl. the RCCL bucket all-reduces' if comm_kind == 'native' else
                                       'hipGraph replay of fwd+bwd, then RCCL bucket all-reduce + fused optimizer'),
                       'gradient_transport': (None if not ddp else 'mmi_allreduce_bucket (RCCL called directly on the reducer stream)'
                                              if comm_kind == 'native' else 'gloo, all ranks on GPU 0 (REHEARSAL of the N > 1 code path: not a measurement)'
                                              if comm_kind == 'gloo' else 'torch.distributed ProcessGroupNCCL'),
                       'ranks_seen': ranks_seen, 'rccl_version': rccl_version, 'cpu_affinity_per_rank': cpu_sets,
                       'host_enqueue_ms_per_step': round(t_enq / args.steps * 1e3, 2),
                       'host_enqueue_ms_per_step_per_rank': [round(v, 2) for v in enq_ranks],
                       'loss': [round(float(v), 5) for v in items.tolist()]},
        }
        if not args.no_roofline:
            tot_f, tot_ms, per = timer.summary()
            if args.dump_gemm:
                with open(args.dump_gemm, 'w') as fh:
                    fh.write('\n'.join(timer.by_shape()) + '\n')
            ach = tot_f / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
            # HBM traffic of the igemm family per step from the PMC counters: collected off line (two rocprofv3 --pmc passes over
            # this very command, tools/pmc_step.sh) and committed under profiles/; reported only for the workload it was taken on
            traffic = None
            try:
                import glob
                pmc_file = sorted(glob.glob(os.path.join(REPO, 'profiles', 'r*_pmc_step_traffic.json')))[-1]   # the latest round's record
                with open(pmc_file) as fh:
                    pt = json.load(fh)
                if pt.get('workload') == args.workload and bs == WORKLOADS[args.workload][5]:
                    traffic = {'read_MB_per_step': pt['read_MB_per_step'].get('igemm'), 'write_MB_per_step': pt['write_MB_per_step'].get('igemm'),
                               'unit': 'MB of HBM/fabric traffic of the igemm family per training step (PMC: FETCH_SIZE x2 + WRITE_SIZE)',
                               'all_families_MB_per_step': pt.get('total_MB_per_step'),
                               'source': 'profiles/%s (tools/pmc_step.sh)' % os.path.basename(pmc_file)}
            except (OSError, ValueError, KeyError, IndexError):
                pass
            out['roofline'] = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': round(ach / PEAK_FP32_MFMA_TFLOPS, 4), 'traffic': traffic,
                               'kernel': 'igemm_kernel / wgrad_kernel (fp32 MFMA implicit GEMM: conv+linear fwd, dgrad, wgrad)',
                               'launches_per_step': len(timer.recs) // max(roof_steps, 1),
                               'gemm_busy_ms_per_step': round(tot_ms / roof_steps, 3),
                               'measured_on': '%d eager steps right after the timed region' % roof_steps,
                               'timing': 'HIP events on each launch stream; busy = union of launch intervals (dgrad and wgrad overlap on two streams)',
                               'per_call': {k: {'TFLOP/s_while_sharing_the_chip': round(v[0] / (v[1] * 1e-3) / 1e12, 2), 'sum_launch_ms_per_step': round(v[1] / roof_steps, 3),
                                                'launches_per_step': v[2] // roof_steps} for k, v in per.items()},
                               'step_algorithmic_TFLOP/s': round(3 * WORKLOADS[args.workload][6] * 1e9 * value / 1e12, 2)}
        if split:
            what = {'exact_auto': 'every fp32 product exact and fp32 accumulation in every GEMM, as the headline: per layer shape and direction the '
                                  'fp32-MFMA kernel or the nine-product bf16 kernel, whichever measured faster in the step (mmi_set_gemm_precision(6): '
                                  '3x3 layers of >= 128 channels and the 1024 x 4096 projections take the latter)',
                    'bf16x9': 'three bf16 terms per operand (an exact representation of the fp32 significand), all nine products on '
                              'v_mfma_f32_32x32x16_bf16, fp32 accumulation: every product exact as with the fp32 MFMA; whole-step '
                              'gradients vs the oracle as the fp32 path (medians 9e-4..1.6e-3 over four seeds at full depth)',
                    'bf16x6': 'the six products of total order <= 2 (dropped terms <= 2^-24): a GEMM differs from the fp32 kernel by '
                              '3e-7..1e-6; full-depth gradients statistically as the fp32 path (four seeds: '
                              'profiles/r01_gemm_modes_full_size_gradients.txt)',
                    'bf16x3': 'two bf16 terms, three products: a GEMM is off by 4.5e-6, predictions 1e-4, full-depth gradients ~1e-2'}
            out['gemm_modes_optional'] = {
                'note': 'NOT the headline (which uses exact fp32 MFMA products): the same step with mmi_set_gemm_precision(3 / 2 / 1); '
                        'off by default, see DESIGN.md',
                **{k: {'value': round(world * bs / v[0], 3), 'unit': 'paired img/s', 'ms_per_step': round(v[0] * 1e3, 3),
                       'loss_finite': v[1], 'last_loss': v[2], 'arithmetic': what[k],
                       # step-level roofline of the mode: the step's algorithmic FLOPs against the bf16 MFMA peak (16x the fp32
                       # MFMA rate, MI355X_MICROARCH.md) divided by the bf16 products one fp32 product costs in this mode
                       'roofline': {'bound': 'mfma', 'unit': 'TFLOP/s (fp32-equivalent, whole step)',
                                    'achieved': round(3 * WORKLOADS[args.workload][6] * 1e9 * world * bs / v[0] / 1e12, 2),
                                    'peak': round(16 * PEAK_FP32_MFMA_TFLOPS / int(k[5:]), 1) if k.startswith('bf16x') else None,
                                    'frac': round(3 * WORKLOADS[args.workload][6] * 1e9 * world * bs / v[0] / 1e12
                                                  / (16 * PEAK_FP32_MFMA_TFLOPS / int(k[5:])), 4) if k.startswith('bf16x') else None}} for k, v in split.items()}}
        if pcie is not None:
            out['pcie_inclusive'] = pcie
        if not args.no_roofline and args.workload != 's_add':
            try:
                out['hbm_bound_ops'] = hbm_ops(model, bs, size, dev)
            except Exception as e:          # (an optional report never fails the run)
                out['hbm_bound_ops'] = {'error': repr(e)[:200]}
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(args.workload)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + '\n').encode())
    if ddp:
        if comm_kind == 'native':
            from mmidet_hip import lib as _l
            torch.cuda.synchronize()
            _l.comm_destroy()
        dist.destroy_process_group()
    if bisect:
        from mmidet_hip import lib as _l
        torch.cuda.synchronize()
        if _l.comm_world() > 0:
            _l.comm_destroy()
        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == '__main__':
    main()
