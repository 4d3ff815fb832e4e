"""Data-parallel gradient reduction for the two-stream step: one process per GPU, RCCL (torch.distributed backend
"nccl") over xGMI, flat gradient buckets all-reduced on a side HIP stream while backward is still running.

Replaces DistributedDataParallel at train.py:683-686 of the reference (whose DDP path cannot actually run: SURVEY.md
§3.1 B1).  Semantics kept: every rank holds the mean over ranks of the per-rank gradients; with `loss *= world_size`
(train.py:790-791) that is the sum of the per-rank mean-loss gradients.

MI355X notes: xGMI is point-to-point (7 links x ~153 GB/s per GPU), so ring all-reduce time is per-link bound and
latency matters more than on a switched fabric -> few LARGE buckets (default 256 MB; 832 MB of fp32 grads for yolov5l =
4 collectives).  Buckets are filled in reverse registration order (the order gradients become ready: head/neck, then
the P5 transformer with 100 M parameters first), and parameter .grad tensors are views of the flat bucket, so there is
no pack/unpack copy on either side of the collective.
"""
import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ('flat', 'params', 'pending', 'work', 'launched')

    def __init__(self, flat, params):
        self.flat, self.params, self.pending, self.work, self.launched = flat, params, len(params), None, False


class GradReducer:
    def __init__(self, params, bucket_mb=256, process_group=None):
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        params = [p for p in params if p.requires_grad]
        cap = int(bucket_mb * 1024 * 1024) // 4
        self.buckets, cur, n = [], [], 0
        for p in reversed(params):
            if cur and n + p.numel() > cap:
                self.buckets.append(self._make(cur))
                cur, n = [], 0
            cur.append(p)
            n += p.numel()
        if cur:
            self.buckets.append(self._make(cur))
        self._of = {}
        for b in self.buckets:
            for p in b.params:
                self._of[p] = b
                p.register_post_accumulate_grad_hook(self._ready)
        self.enabled = True      # False while a hipGraph of forward+backward is captured/replayed: see reduce_now()
        self.cuda = params[0].is_cuda
        self.comm = torch.cuda.Stream(device=params[0].device) if self.cuda else None
        self.avg = dist.ReduceOp.AVG if (self.cuda and dist.get_backend(process_group) == 'nccl') else dist.ReduceOp.SUM

    @staticmethod
    def _make(params):
        total = sum(p.numel() for p in params)
        flat = torch.zeros(total, dtype=params[0].dtype, device=params[0].device)
        off = 0
        for p in params:
            # same dense layout as the parameter (conv weights are channels_last = OHWI)
            p.grad = flat.as_strided(p.shape, p.stride(), off)
            off += p.numel()
        return _Bucket(flat, params)

    def broadcast_parameters(self, module, src=0):
        """Rank-0 weights/buffers to every rank once (what DDP's constructor does)."""
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src, group=self.pg)

    def prepare(self):
        for b in self.buckets:
            b.pending, b.work, b.launched = len(b.params), None, False

    def _launch(self, b):
        b.launched = True
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(ev)                      # the bucket's last gradient has been written
                b.work = dist.all_reduce(b.flat, op=self.avg, group=self.pg, async_op=True)
        else:
            b.work = dist.all_reduce(b.flat, op=self.avg, group=self.pg, async_op=True)

    def _ready(self, p):
        if not self.enabled:
            return
        b = self._of[p]
        b.pending -= 1
        if b.pending == 0 and not b.launched:
            self._launch(b)

    def finish(self):
        """Called after backward: launch stragglers, then make the compute stream wait for every collective."""
        for b in self.buckets:
            if not b.launched:
                self._launch(b)
        for b in self.buckets:
            b.work.wait()
            if self.avg == dist.ReduceOp.SUM:
                b.flat.div_(self.world)

    def reduce_now(self):
        """All buckets on the current stream, after a replayed forward+backward graph (hooks cannot fire inside a replay).
        Not overlapped with backward: 832 MB over xGMI is a few ms against a 150 ms step, and the replay saves far more."""
        for b in self.buckets:
            dist.all_reduce(b.flat, op=self.avg, group=self.pg)
            if self.avg == dist.ReduceOp.SUM:
                b.flat.div_(self.world)

    def zero(self):
        for b in self.buckets:
            b.flat.zero_()
