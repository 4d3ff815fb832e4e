"""Fused SGD(nesterov)+EMA step over the C ABI (csrc/optim.hip): one HIP launch for every parameter/buffer tensor.

Semantics = torch.optim.SGD(nesterov=True, dampening=0) over the reference's three parameter groups (train.py:572-589)
followed by ModelEMA.update (utils/torch_utils.py:286-299).  `param_groups` is exposed like a torch optimizer's so the
reference's warm-up / scheduler code (train.py:765-773: `x['lr'] = ...`, `x['momentum'] = ...`) keeps working; the values
are copied to a 9-float device block right before the launch, which is what lets the launch be replayed from a hipGraph.
"""
import math

import numpy as np
import torch

from . import alloc, lib
from .ops import _stream

_REC = np.dtype([('p', '<u8'), ('g', '<u8'), ('buf', '<u8'), ('ema', '<u8'), ('n', '<i8'), ('group', '<i4'), ('flags', '<i4')])
_CHUNK = 65536  # MMI_OPT_CHUNK


class _Staging:
    """Host -> device upload of a small table through a ring of pinned buffers.  An asynchronous copy reads its pinned source
    when the stream gets to it, which in an eager run is several steps after the host issued it (the host enqueues a step in
    half the time the GPU needs); writing the next step's table into the same pinned buffer in the meantime hands the GPU the
    wrong step's gradient pointers.  Each ring slot is reused only after the copy that last read it has completed."""

    def __init__(self, nbytes, device, depth=4):
        self.cuda = device.type == 'cuda'
        self.bufs = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() if self.cuda else torch.empty(nbytes, dtype=torch.uint8)
                     for _ in range(depth)]
        self.done = [None] * depth
        self.i = 0
        # a captured copy re-reads its pinned source at every replay: it gets a buffer of its own (allocated here, because
        # pinning memory is not allowed while a stream is capturing) that the ring never overwrites
        self.capture_buf = torch.empty(nbytes, dtype=torch.uint8).pin_memory() if self.cuda else None

    def upload(self, src_u8, dst_dev):
        if self.cuda and torch.cuda.is_current_stream_capturing():
            self.capture_buf.copy_(src_u8)
            dst_dev.copy_(self.capture_buf, non_blocking=True)
            return
        k = self.i
        self.i = (k + 1) % len(self.bufs)
        if self.done[k] is not None:
            self.done[k].synchronize()
        self.bufs[k].copy_(src_u8)
        dst_dev.copy_(self.bufs[k], non_blocking=True)
        if self.cuda:
            self.done[k] = torch.cuda.Event()
            self.done[k].record()


def _same_layout(a, b):
    """Same element order in memory: strides agree on every axis longer than 1 (the stride of a length-1 axis is arbitrary -- a
    (O,I,1,1) weight is the same bytes whether it was cloned or re-seated by ops.pack_pair)."""
    return all(sa == sb for n, sa, sb in zip(a.shape, a.stride(), b.stride()) if n > 1)


class FusedSGDEMA(torch.optim.Optimizer):
    """A torch.optim.Optimizer (so that the reference's `lr_scheduler.LambdaLR(optimizer, ...)`, `GradScaler.step(optimizer)`,
    `optimizer.state_dict()` into the checkpoint and `optimizer.load_state_dict(ckpt['optimizer'])` on resume -- train.py:597,
    800, 888, 609 -- work unchanged) whose step is the one fused launch.  State layout = torch.optim.SGD's
    (`state[p]['momentum_buffer']`), so checkpoints move between the two optimizers."""
    TAIL_PREFIX = 'Enhance.'     # state_dict prefix of the modules whose backward runs last (models/yolo_test.py: self.Enhance)

    def __init__(self, model, groups, ema_model=None, ema_decay=0.9999, ema_updates=0):
        """groups: list of dicts {'params': [...], 'lr':, 'momentum':, 'weight_decay':} (at most 3)."""
        assert 1 <= len(groups) <= 3
        for g in groups:
            g.setdefault('initial_lr', g['lr'])
            g.setdefault('weight_decay', 0.0)
            g.setdefault('nesterov', True)
            g.setdefault('dampening', 0)
        super().__init__(groups, dict(lr=groups[0]['lr'], momentum=groups[0]['momentum'], weight_decay=0.0, nesterov=True, dampening=0))
        self.model, self.ema_model = model, ema_model
        self.ema_decay, self.updates = ema_decay, ema_updates
        self.device = next(model.parameters()).device
        self._bufs = {}        # param -> momentum buffer (same memory layout as the parameter; also state[p]['momentum_buffer'])
        self._steps = 0
        self._gptrs = None
        self._recs_host = self._recs_dev = self._chunks_dev = None
        self._hyper_host = torch.zeros(9, dtype=torch.float32)
        self._hyper_dev = torch.zeros(9, dtype=torch.float32, device=self.device)
        self._hyper_stage = _Staging(36, self.device)
        self._build()

    # ---- table construction ------------------------------------------------------------------------------------------
    def _build(self):
        msd = self.model.state_dict(keep_vars=True)
        esd = self.ema_model.state_dict(keep_vars=True) if self.ema_model is not None else {}
        ema_of = {}
        for k, v in msd.items():
            if k in esd and v.dtype.is_floating_point:
                e = esd[k]
                assert e.shape == v.shape and _same_layout(e, v), 'EMA copy must share the parameter layout: ' + k
                ema_of[v.data_ptr()] = e
        rows, self._sgd_params = [], []
        done = set()
        for gi, g in enumerate(self.param_groups):
            for p in g['params']:
                if not p.requires_grad:
                    continue
                buf = torch.zeros_like(p, memory_format=torch.preserve_format)
                self._bufs[p] = buf
                self.state[p]['momentum_buffer'] = buf
                e = ema_of.get(p.data_ptr())
                rows.append([p.data_ptr(), 0, buf.data_ptr(), e.data_ptr() if e is not None else 0, p.numel(), gi,
                             1 | (2 if e is not None else 0)])
                self._sgd_params.append(p)
                done.add(p.data_ptr())
        for k, v in msd.items():            # EMA-only entries: buffers and parameters that are in no group
            if v.dtype.is_floating_point and v.data_ptr() not in done and v.data_ptr() in ema_of:
                rows.append([v.data_ptr(), 0, 0, ema_of[v.data_ptr()].data_ptr(), v.numel(), 0, 2])
                done.add(v.data_ptr())
        rec = np.zeros(len(rows), dtype=_REC)
        for i, r in enumerate(rows):
            rec[i] = tuple(r)
        self._recs_host = rec
        key_of = {}
        for k, v in msd.items():
            key_of.setdefault(v.data_ptr(), k)
        self._row_keys = [key_of.get(r[0], '') for r in rows]       # state_dict key of every record (set_parts partitions by it)
        chunks = [(i, c) for i, r in enumerate(rows) for c in range((r[4] + _CHUNK - 1) // _CHUNK)]
        self._chunks_host = chunks
        self._parts = None
        ck = np.array(chunks, dtype=np.int32).reshape(-1, 2)
        self._nchunks = len(chunks)
        self._chunks_dev = torch.from_numpy(ck).to(self.device)
        # Two-part launch (TrainStep's early optimizer): "tail" = the records of the modules whose gradients arrive last in
        # backward (the Contour Enhancement Module in front of the RGB stem); everything else can be stepped while the tail's
        # backward is still running.  Same table, two chunk lists.
        tail_ptrs = {v.data_ptr() for k, v in msd.items() if k.startswith(self.TAIL_PREFIX)}
        self._tail_rows = np.array([r[0] in tail_ptrs for r in rows], dtype=bool)
        self._sgd_tail = [p.data_ptr() in tail_ptrs for p in self._sgd_params]
        self._part_chunks = {}
        for name, want in (('head', False), ('tail', True)):
            sel = [c for c in chunks if bool(self._tail_rows[c[0]]) == want]
            arr = np.array(sel, dtype=np.int32).reshape(-1, 2)
            self._part_chunks[name] = (torch.from_numpy(arr).to(self.device) if len(sel) else None, len(sel))
        nbytes = rec.view(np.uint8).reshape(-1).size
        self._recs_stage = _Staging(nbytes, self.device)
        self._recs_dev = alloc.empty(nbytes, dtype=torch.uint8, device=self.device)

    def _refresh_grads(self, part=None):
        """Gradients are fresh tensors every eager step (AccumulateGrad steals them); re-point the table when they move.
        Under graph capture/replay the addresses are static and this is a no-op after the capture pass.
        part='head': only the non-tail parameters must have their gradient yet (the tail's backward is still to run; its rows
        keep last step's pointers, which the head launch never reads).  Returns False if a needed gradient is missing."""
        ptrs = tuple(p.grad.data_ptr() if p.grad is not None else 0 for p in self._sgd_params)
        if part == 'head':
            if any(gp == 0 and not tl for gp, tl in zip(ptrs, self._sgd_tail)):
                return False
            old = self._gptrs or (0,) * len(ptrs)
            ptrs = tuple(gp if not tl else og for gp, tl, og in zip(ptrs, self._sgd_tail, old))
        if ptrs == self._gptrs:
            return True
        self._gptrs = ptrs
        rec = self._recs_host
        n = len(ptrs)
        for i, (p, gp) in enumerate(zip(self._sgd_params, ptrs)):
            if part == 'head' and self._sgd_tail[i]:
                continue
            if gp == 0:
                raise RuntimeError('parameter without gradient in an optimiser group (all are trained in the reference)')
            assert _same_layout(p.grad, p), 'gradient layout differs from the parameter layout'
        rec['g'][:n] = np.array(ptrs, dtype=np.uint64)
        al = ((rec['p'] | rec['g'] | rec['buf'] | rec['ema']) & np.uint64(15)) == 0
        rec['flags'] = (rec['flags'] & ~np.int32(4)) | np.where(al, 4, 0).astype(np.int32)
        self._recs_stage.upload(torch.from_numpy(rec.view(np.uint8).reshape(-1)), self._recs_dev)
        return True

    # ---- per step ----------------------------------------------------------------------------------------------------
    def upload_hyper(self, advance=True):
        """Host -> device copy of the 9 hyper-parameters; call before a graph replay (step() calls it itself)."""
        if advance:
            self.updates += 1
        d = self.ema_decay * (1 - math.exp(-self.updates / 2000)) if self.ema_model is not None else 0.0
        h = self._hyper_host
        for i in range(3):
            g = self.param_groups[min(i, len(self.param_groups) - 1)]
            h[i], h[3 + i] = g['lr'], g['weight_decay']
        h[6], h[7], h[8] = self.param_groups[0]['momentum'], d, 1.0 if self._steps == 0 else 0.0
        self._hyper_stage.upload(h.view(torch.uint8), self._hyper_dev.view(torch.uint8))
        self._steps += 1

    def launch(self):
        self._refresh_grads()
        lib.sgd_ema_step(self._recs_dev.data_ptr(), self._chunks_dev.data_ptr(), self._nchunks, self._hyper_dev.data_ptr(),
                         _stream())

    def launch_part(self, part, stream=None):
        """'head' (everything but the tail records) or 'tail'; head's table refresh happens on the CURRENT stream -- the caller
        orders `stream` behind it.  Returns False (nothing launched) when a head gradient is not there yet."""
        if not self._refresh_grads('head' if part == 'head' else None):
            return False
        ck, n = self._part_chunks[part]
        if n:
            lib.sgd_ema_step(self._recs_dev.data_ptr(), ck.data_ptr(), n, self._hyper_dev.data_ptr(),
                             _stream() if stream is None else stream)
        return True

    # ---- launch parts (TrainStep's early optimizer, generalised): the records in `nparts` groups launched in order, each as soon
    # as the gradients of its parameters exist ------------------------------------------------------------------------------
    def set_parts(self, part_of_key, nparts):
        """part_of_key: state_dict key -> 0 .. nparts-1 (the order in which the parts' gradients complete in backward)."""
        rp = np.array([int(part_of_key(k)) for k in self._row_keys], dtype=np.int32)
        assert rp.min() >= 0 and rp.max() < nparts
        self._sgd_part = [int(rp[i]) for i in range(len(self._sgd_params))]      # (the SGD records are the first rows, in order)
        self._parts = []
        for q in range(nparts):
            sel = [c for c in self._chunks_host if rp[c[0]] == q]
            arr = np.array(sel, dtype=np.int32).reshape(-1, 2)
            self._parts.append((torch.from_numpy(arr).to(self.device) if len(sel) else None, len(sel)))
        # one table upload per part and step: the staging ring must hold four STEPS of them, or the host blocks on the GPU
        nbytes = self._recs_host.view(np.uint8).reshape(-1).size
        self._recs_stage = _Staging(nbytes, self.device, depth=4 * nparts)
        self._hyper_stage = _Staging(36, self.device, depth=8)

    def refresh_upto(self, upto):
        """Point the table at this step's gradients of the parts 0 .. upto (later parts keep the pointers they have: nobody reads
        them yet).  False when one of the needed gradients does not exist yet."""
        ptrs = tuple(p.grad.data_ptr() if p.grad is not None else 0 for p in self._sgd_params)
        if any(gp == 0 and q <= upto for gp, q in zip(ptrs, self._sgd_part)):
            return False
        old = self._gptrs or (0,) * len(ptrs)
        ptrs = tuple(gp if q <= upto else og for gp, q, og in zip(ptrs, self._sgd_part, old))
        if ptrs == self._gptrs:
            return True
        self._gptrs = ptrs
        rec = self._recs_host
        n = len(ptrs)
        rec['g'][:n] = np.array(ptrs, dtype=np.uint64)
        al = ((rec['p'] | rec['g'] | rec['buf'] | rec['ema']) & np.uint64(15)) == 0
        rec['flags'] = (rec['flags'] & ~np.int32(4)) | np.where(al, 4, 0).astype(np.int32)
        self._recs_stage.upload(torch.from_numpy(rec.view(np.uint8).reshape(-1)), self._recs_dev)
        return True

    def launch_parts(self, first, last, stream=None):
        for q in range(first, last + 1):
            ck, n = self._parts[q]
            if n:
                lib.sgd_ema_step(self._recs_dev.data_ptr(), ck.data_ptr(), n, self._hyper_dev.data_ptr(),
                                 _stream() if stream is None else stream)

    def step(self, closure=None):
        assert closure is None, 'FusedSGDEMA.step takes no closure'
        self.upload_hyper()
        self.launch()

    def load_state_dict(self, state_dict):
        """torch's loader replaces the state tensors by copies; the kernel's pointer table holds the addresses of the buffers made
        at construction, so the loaded momenta are copied INTO those (layout conversion included) and the state points back at
        them.  A state with momenta means steps have been taken: the kernel's first-step form (buffer = gradient) is over."""
        super().load_state_dict(state_dict)
        loaded = False
        with torch.no_grad():
            for p, buf in self._bufs.items():
                mb = self.state[p].get('momentum_buffer') if p in self.state else None
                if mb is not None and mb.data_ptr() != buf.data_ptr():
                    buf.copy_(mb)
                    loaded = True
                self.state[p]['momentum_buffer'] = buf
        if loaded:
            self._steps = max(self._steps, 1)

    def zero_grad(self, set_to_none=True):
        for p in self._sgd_params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()
        if getattr(self, '_ungrouped', None) is None:   # parameters in no group (pos_emb, sobel_factor) still receive gradients
            ids = {id(p) for p in self._sgd_params}
            self._ungrouped = [p for p in self.model.parameters() if id(p) not in ids]
        if set_to_none:
            for p in self._ungrouped:
                p.grad = None
