"""Data-parallel gradient reduction for the two-stream step: one process per GPU, RCCL over xGMI, flat gradient buckets
all-reduced on a side HIP stream while backward is still running.

Two transports, same bucket logic: "native" = the library's own communicator (mmi_comm_init / mmi_allreduce_bucket,
include/mmidet_hip.h: RCCL called directly on the reducer's HIP stream -- no ProcessGroup, no watchdog thread, capturable into
a hipGraph) and "torch" = torch.distributed (backend "nccl" on the GPU, "gloo" in the CPU tests).  init_native_comm() sets the
former up over any torch.distributed group that can carry 128 bytes (gloo is enough).

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


# cost-attribution switches (tools only): 'noreduce' = hooks without collectives, 'nohooks' = all buckets reduced in finish(),
# 'nosidewait' = collectives do not wait for the wgrad streams (WRONG results; timing experiments only)
_DEBUG = __import__('os').environ.get('MMIDET_DDP_DEBUG', '')


def init_native_comm(rank, world, group=None):
    """Create the library's RCCL communicator for this process (idempotent).  Rank 0's 128-byte id travels over `group`
    (any initialised torch.distributed group; not needed at world size 1).  torch.cuda.set_device() must have been called."""
    import ctypes
    from . import lib
    if lib.comm_world() == world and lib.comm_rank() == rank:
        return
    if lib.comm_world() != 0:
        lib.comm_destroy()
    buf = ctypes.create_string_buffer(128)
    if rank == 0:
        lib.comm_unique_id(buf)
    box = [buf.raw if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    lib.comm_init(rank, world, ctypes.create_string_buffer(box[0], 128))


class _Bucket:
    __slots__ = ('flat', 'params', 'pending', 'work', 'launched', 'streams')

    def __init__(self, flat, params):
        self.flat, self.params, self.pending, self.work, self.launched = flat, params, len(params), None, False
        self.streams = {}      # HIP streams that wrote one of this bucket's gradients this step (handle -> Stream)


class GradReducer:
    """direct=True (CUDA): gradients are not accumulated into the buckets by autograd.  `.grad` is None when backward starts;
    the weight-gradient kernels write straight into the parameter's bucket view (ops.GRAD_SLOTS) and autograd adopts that
    view as `.grad`; the remaining small gradients (BatchNorm/LayerNorm/bias vectors, pos_emb) are copied into their view by
    the grad-ready hook.  That removes a read-modify-write pass over 832 MB per step and lets TrainStep keep the
    deferred-join wgrad overlap under data parallelism (nobody reads a weight gradient during backward).
    Direct mode holds only while every backward starts with `.grad is None` (the kernels OVERWRITE the bucket view; autograd
    would then add the view to itself): prepare() enforces it, and gradient accumulation over several backward passes needs
    set_direct(False) (TrainStep switches by itself), where autograd accumulates into the bucket views as usual."""

    def __init__(self, params, bucket_mb=256, process_group=None, direct=None, comm=None):
        from . import lib
        self.pg = process_group
        if comm is None:           # the library's own communicator when one has been set up, torch.distributed otherwise
            comm = 'native' if (params[0].is_cuda and lib.comm_world() > 0) else 'torch'
        assert comm in ('native', 'torch')
        self.native = comm == 'native'
        self._lib = lib
        if self.native:
            assert lib.comm_world() > 0, 'GradReducer(comm="native"): call ddp.init_native_comm(rank, world) first'
        self.world = lib.comm_world() if self.native else dist.get_world_size(process_group)
        params = [p for p in params if p.requires_grad]
        bucket_mb = float(__import__('os').environ.get('MMIDET_BUCKET_MB', bucket_mb))   # (tuning override)
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
                if _DEBUG != 'nohooks':
                    p.register_post_accumulate_grad_hook(self._ready)
        self._stale_ok = False   # .grad tensors left by zero(keep_grads=True) are leftovers, not gradients to accumulate onto
        self.enabled = True      # False while a hipGraph of forward+backward is captured/replayed: see reduce_now()
        self.cuda = params[0].is_cuda
        self._slot = {p: p.grad for b in self.buckets for p in b.params}      # the bucket views made by _make
        self.direct = False
        from . import ops
        self._ops = ops
        # default: direct on the GPU; on the CPU (gloo tests) only on request -- the bookkeeping is the same, the kernels that
        # write the slots are the caller's
        self.set_direct(self.cuda if direct is None else direct)
        self.comm = torch.cuda.Stream(device=params[0].device) if self.cuda else None
        self.avg = dist.ReduceOp.AVG if (self.cuda and (self.native or dist.get_backend(process_group) == 'nccl')) else dist.ReduceOp.SUM

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

    def set_direct(self, flag):
        """Switch between the two gradient forms (see the class docstring).  Only between steps: the buckets are zeroed."""
        flag = bool(flag)
        if flag:
            for p, slot in self._slot.items():
                self._ops.GRAD_SLOTS[p.data_ptr()] = slot
                p.grad = None
        else:
            for b in self.buckets:
                b.flat.zero_()
            for p, slot in self._slot.items():
                self._ops.GRAD_SLOTS.pop(p.data_ptr(), None)
                p.grad = slot
        self._ops.SLOT_HANDED_OUT.clear()
        self.direct = flag

    def broadcast_parameters(self, module, src=0):
        """Rank-0 weights/buffers to every rank once (what DDP's constructor does)."""
        for t in list(module.parameters()) + list(module.buffers()):
            if self.native:        # dense storage (channels_last weights included): the bytes as they lie
                self._lib.broadcast_bytes(t.data_ptr(), t.numel() * t.element_size(), src, torch.cuda.current_stream().cuda_stream)
            else:
                dist.broadcast(t.data, src, group=self.pg)

    def prepare(self):
        """Before every backward."""
        for b in self.buckets:
            b.pending, b.work, b.launched = len(b.params), None, False
            b.streams.clear()
        if self.direct:
            # The kernels overwrite the slots and autograd must ADOPT them.  A `.grad` left over from the previous backward
            # (a replayed graph keeps them: zero(keep_grads=True); a caller that skipped zero()) would make AccumulateGrad
            # compute slot += alias-of-slot = 2*g and lose what was there.  Leftovers of zero(keep_grads=True) are dropped
            # here; anything else means the caller is accumulating over several backward passes, which needs
            # set_direct(False), and is told so instead of getting wrong numbers.
            for p in self._slot:
                if p.grad is not None:
                    if not self._stale_ok:
                        raise RuntimeError('GradReducer: a backward pass starts with .grad set and no zero() since the last '
                                           'one; direct mode cannot accumulate gradients (call set_direct(False) first)')
                    p.grad = None
            self._stale_ok = False
            self._ops.SLOT_HANDED_OUT.clear()

    def _order_behind_writers(self, b):
        """Make the comm stream wait for EVERY stream that wrote into this bucket: backward runs on two lane streams (RGB /
        IR backbone) and a bucket mixes parameters of both, so the stream of the last-ready hook alone is not enough; plus
        the wgrad side streams whose kernels write the slots directly (deferred join)."""
        for st in b.streams.values():
            self.comm.wait_stream(st)
        if self.direct and _DEBUG != 'nosidewait':
            for sd in self._ops.side_streams_in_flight():
                self.comm.wait_stream(sd)

    def _launch(self, b):
        b.launched = True
        if _DEBUG == 'noreduce':
            return
        if self.cuda:
            cur = torch.cuda.current_stream()      # finish() launches stragglers from the caller's stream
            b.streams[cur.cuda_stream] = cur
            with torch.cuda.stream(self.comm):
                self._order_behind_writers(b)
                if self.native:
                    self._lib.allreduce_bucket(b.flat.data_ptr(), b.flat.numel(), 1, self.comm.cuda_stream)
                    b.work = True
                else:
                    b.work = dist.all_reduce(b.flat, op=self.avg, group=self.pg, async_op=True)
        else:
            self._order_behind_writers_host(b)
            b.work = dist.all_reduce(b.flat, op=self.avg, group=self.pg, async_op=True)

    def _order_behind_writers_host(self, b):
        """CPU tensors are written synchronously: nothing to wait for (the gloo tests override this to model side streams)."""

    def _ready(self, p):
        b = self._of[p]
        if self.cuda:
            cur = torch.cuda.current_stream()          # the stream this gradient was written / accumulated on
            b.streams[cur.cuda_stream] = cur
        if self.direct:
            slot = self._slot[p]
            if p.grad.data_ptr() != slot.data_ptr():          # produced elsewhere (small vectors): move it into the bucket
                if p.data_ptr() in self._ops.SLOT_HANDED_OUT:
                    raise RuntimeError('autograd copied a gradient that was written into its bucket view instead of '
                                       'adopting it (the copy may have read it before its wgrad stream finished)')
                if self.cuda:
                    # the gradient may have been produced on a wgrad side stream that the lane has not joined yet (deferred
                    # join): order the copy behind every side stream in flight (rare path: a handful of small tensors)
                    for sd in self._ops.side_streams_in_flight():
                        cur.wait_stream(sd)
                slot.copy_(p.grad)
                p.grad = slot.detach()
        if not self.enabled:
            return
        b.pending -= 1
        if b.pending == 0 and not b.launched:
            self._launch(b)

    def finish(self):
        """Called after backward: launch stragglers, then make the compute stream wait for every collective."""
        for b in self.buckets:
            if not b.launched:
                self._launch(b)
        if self.native:
            torch.cuda.current_stream().wait_stream(self.comm)      # one join for all buckets
            return
        for b in self.buckets:
            if b.work is None:
                continue
            b.work.wait()
            if self.avg == dist.ReduceOp.SUM:
                b.flat.div_(self.world)

    def reduce_now(self):
        """All buckets on the current stream, after a replayed forward+backward graph (hooks cannot fire inside a replay).
        Not overlapped with backward: 832 MB over xGMI is a few ms against a 150 ms step, and the replay saves far more."""
        for b in self.buckets:
            if self.native:       # enqueued on the current stream: also legal inside a stream capture
                self._lib.allreduce_bucket(b.flat.data_ptr(), b.flat.numel(), 1, torch.cuda.current_stream().cuda_stream)
                continue
            dist.all_reduce(b.flat, op=self.avg, group=self.pg)
            if self.avg == dist.ReduceOp.SUM:
                b.flat.div_(self.world)

    def zero(self, keep_grads=False):
        """After the optimizer step.  direct: every gradient is rewritten in full next step, so dropping `.grad` is all
        (keep_grads: a replayed graph rewrites the same views without going through autograd again)."""
        if self.direct:
            self._ops.SLOT_HANDED_OUT.clear()
            self._stale_ok = keep_grads
            if not keep_grads:
                for p in self._slot:
                    p.grad = None
            return
        for b in self.buckets:
            b.flat.zero_()
