"""The training step of the hot path: what train.py:743-804 of the reference does per batch, as a reusable harness
(the reference's full script - logging, W&B, evolve, checkpoint I/O - is out of scope).

    imgs uint8 (B,6,H,W) + targets (nT,6)  ->  /255, split RGB/IR (743-745)  ->  model(rgb, ir) (788)
    -> ComputeLoss (789)  -> loss *= world_size (790-791)  -> backward (796)  [+ bucketed RCCL all-reduce]
    -> SGD(nesterov) step, zero_grad (799-803)  -> ModelEMA.update (804)

Optimizer grouping follows train.py:572-589: BN weights (no decay) / other .weight (decay) / .bias; parameters that are
in no group (pos_emb, sobel_factor, frozen sobel_weight) are never stepped, as in the reference.

MI355X specifics: the optimizer + EMA is ONE fused multi-tensor HIP launch with device-resident hyper-parameters, the
dropout masks come from a device seed word, and nothing in the step synchronises with the host, so the whole step
(~4000 kernel launches) can be captured once into a hipGraph and replayed (`graph=True`): the host then spends
microseconds per step instead of ~175 ms enqueueing launches.  Graph mode needs a fixed batch shape and target count
(pad the target list with rows whose image index is negative: the assignment kernel skips them).
"""
import os

import time

import torch
import torch.nn as nn

from models.yolo_test import Model  # noqa: F401  (re-export for callers)
from utils.loss import ComputeLoss
from utils.torch_utils import ModelEMA

from . import fusion_ops as F2
from . import ops
from .optim import FusedSGDEMA

HYP_SCRATCH = dict(lr0=0.01, lrf=0.2, momentum=0.937, weight_decay=0.0005, warmup_epochs=3.0, warmup_momentum=0.8,
                   warmup_bias_lr=0.1, box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, iou_t=0.20, anchor_t=4.0,
                   fl_gamma=0.0)  # data/hyp.scratch.yaml:6-22


def scale_hyp(hyp, nc, imgsz, nl=3):
    """train.py:689-691."""
    h = dict(hyp)
    h['box'] *= 3. / nl
    h['cls'] *= nc / 80. * 3. / nl
    h['obj'] *= (imgsz / 640) ** 2 * 3. / nl
    return h


def param_groups(model):
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


def build_optimizer(model, hyp, total_batch_size, fused=True, ema_model=None):
    nbs = 64
    accumulate = max(round(nbs / total_batch_size), 1)
    wd = hyp['weight_decay'] * total_batch_size * accumulate / nbs                      # train.py:568-570
    pg0, pg1, pg2 = param_groups(model)
    if fused:
        groups = [dict(params=pg0, lr=hyp['lr0'], momentum=hyp['momentum'], weight_decay=0.0),
                  dict(params=pg1, lr=hyp['lr0'], momentum=hyp['momentum'], weight_decay=wd),
                  dict(params=pg2, lr=hyp['lr0'], momentum=hyp['momentum'], weight_decay=0.0)]
        return FusedSGDEMA(model, groups, ema_model=ema_model), accumulate
    opt = torch.optim.SGD(pg0, lr=hyp['lr0'], momentum=hyp['momentum'], nesterov=True)  # train.py:585-589
    opt.add_param_group({'params': pg1, 'weight_decay': wd})
    opt.add_param_group({'params': pg2})
    return opt, accumulate


class TrainStep:
    def __init__(self, model, nc, imgsz, batch_size, world_size=1, reducer=None, hyp=None, ema=True, accumulate=None,
                 fused_optimizer=True, graph=False, defer_join=True):
        self.model = model
        self.world_size = world_size
        self.reducer = reducer
        hyp = scale_hyp(HYP_SCRATCH if hyp is None else hyp, nc, imgsz)
        model.nc, model.hyp, model.gr = nc, hyp, 1.0                                    # train.py:693-695
        if reducer is None:            # (a reducer has already keyed its gradient slots on the parameter addresses)
            F2.pack_qkv(model)         # q/k/v projections of the fusion transformers as one GEMM each way
            if ops.PACK_C3:
                ops.pack_pair(model)   # cv1 | cv2 of every C3 as one GEMM
        self.ema = ModelEMA(model) if ema else None
        self.fused = fused_optimizer
        self.optimizer, self.accumulate = build_optimizer(model, hyp, batch_size * world_size, fused=fused_optimizer,
                                                          ema_model=self.ema.ema if self.ema is not None else None)
        if accumulate is not None:                                                      # e.g. 1: optimizer + EMA every batch
            self.accumulate = accumulate
        self.compute_loss = ComputeLoss(model)
        self.ni = 0
        self.use_graph = graph
        self.defer_join = defer_join
        self._graph = None
        self._graph_has_collectives = False
        assert not (graph and self.accumulate != 1), 'graph mode captures one full step: accumulate must be 1'
        assert not (graph and not fused_optimizer), 'graph mode needs the fused optimizer (device-resident hyper-parameters)'
        # Early optimizer (eager, one GPU): the step ends on a lone stream -- the Contour Enhancement Module's backward (3 ms
        # at 640x640) and then the optimizer (1 ms).  Every other gradient is complete when the gradient of the CEM's OUTPUT
        # arrives, so a tensor hook there launches the optimizer for everything but the CEM's own parameters on a stream of
        # its own, next to the CEM backward; the CEM's records follow after backward (FusedSGDEMA.launch_part).
        # Measured (profiles/r02_ab_early_optimizer.txt, three interleaved pairs): 121.47 vs 121.65 ms -- both sides of the overlap
        # stream HBM, so it buys 0.2 ms; bit-identical training (tests/test_step_gpu.py).  Off unless MMIDET_EARLY_OPT=1.
        self.early_opt = os.environ.get('MMIDET_EARLY_OPT', '0') == '1'
        self._opt_stream = None
        self._head_launched = False
        # Generalised (MMIDET_OPT_PARTS=n, n >= 2): the records in n parts by the order their gradients complete in backward
        # (head / neck and the P5 transformer first ... the stems and the CEM last); tensor hooks on the outputs of n - 1 layers
        # launch every part whose gradients are complete on the optimizer's own stream, next to the MFMA-bound rest of the backward.
        # Measured (round 4, three interleaved pairs, yolov5l B=16): n = 3 119.87 / 120.08 ms against 119.86 / 120.09 with the one launch --
        # the optimizer streams 8 GB through HBM and the Infinity Cache, which the co-running GEMMs pay for by as much as the
        # tail gets shorter (the same answer the two-part form gave in round 2).  Off by default; capped at 4 parts (n = 5 made
        # the host the bottleneck, 180 ms per step, not pursued).
        self.opt_parts = min(int(os.environ.get('MMIDET_OPT_PARTS', '0')), 4)
        self._parts_launched = 0
        self._part_hooks = {}
        if self.opt_parts >= 2 and self.fused and reducer is None and hasattr(model, 'model'):
            self._plan_parts(self.opt_parts)

    # ---- the step body (eager; also what gets captured) ----------------------------------------------------------------
    def _body(self, imgs_u8, targets, reduce=True):
        model = self.model
        # wgrad overlap with ONE join after backward (ops.join_pending) instead of one per layer: allowed when nobody reads a
        # weight gradient during backward, i.e. .grad is None (adopted untouched by autograd) -- not with the data-parallel
        # flat buckets or gradient accumulation, which add into existing .grad tensors as soon as a layer is done.
        if self.reducer is not None and self.reducer.direct and self.accumulate != 1:
            self.reducer.set_direct(False)       # accumulation: autograd adds into the bucket views (ddp.GradReducer)
        direct = self.reducer is None or getattr(self.reducer, 'direct', False)
        defer = self.defer_join and direct and self.accumulate == 1 and ops.OVERLAP_WGRAD
        F2.advance_seed(imgs_u8.device)                                                 # new dropout masks this step
        if imgs_u8.dtype == torch.uint8:
            rgb, ir = ops.u8_pair_to_nhwc(imgs_u8)                                      # train.py:743-745 in one kernel
        else:
            imgs = imgs_u8.float() / 255.0                                              # train.py:743
            rgb, ir = imgs[:, :3], imgs[:, 3:]                                          # train.py:744-745 (strided views)
        early = (self.early_opt and defer and self.fused and self.reducer is None and self.accumulate == 1 and imgs_u8.is_cuda
                 and not torch.cuda.is_current_stream_capturing() and getattr(model, 'two_streams', False))
        parts = (bool(self._part_hooks) and not early and defer and self.fused and self.reducer is None and self.accumulate == 1
                 and imgs_u8.is_cuda and not torch.cuda.is_current_stream_capturing())
        model._tail_hook = self._on_tail_gradient if early else None                    # registered on the CEM's output in forward
        model._grad_hooks = self._part_hooks if parts else None                         # layer index -> hook on that layer's output
        self._parts_launched = 0
        self._parts_on = parts
        try:
            pred, comb = model(rgb, ir)                                                 # train.py:788
        finally:
            model._tail_hook = None
            model._grad_hooks = None
        loss, items = self.compute_loss(pred, targets, comb.reshape(-1))                # train.py:789 (+ B2 reshape)
        if self.world_size > 1:
            loss = loss * self.world_size                                               # train.py:790-791
        reduce = reduce and self.reducer is not None
        if reduce:
            self.reducer.prepare()
        if defer and self.reducer is None:
            # Deferred join is only sound when autograd ADOPTS every weight gradient untouched: with a .grad already there
            # AccumulateGrad would read dw on the lane stream while its wgrad stream may still be writing it.
            stale = next((n for n, p in self._named_params() if p.grad is not None), None)
            if stale is not None:
                raise RuntimeError('TrainStep: %s.grad is set at the start of backward (zero_grad(set_to_none=True) was skipped); '
                                   'the deferred-join wgrad overlap needs .grad to be None' % stale)
        ops.DEFER_JOIN = defer
        self._head_launched = False
        try:
            loss.sum().backward()                                                       # train.py:796
        finally:
            ops.DEFER_JOIN = False
            ops.join_pending()
        if reduce:
            self.reducer.finish()                                                       # mean over ranks, as DDP
        return loss, items

    def _on_tail_gradient(self, grad):
        """Tensor hook on the CEM's output (runs in backward right before the CEM's own backward node, on the RGB lane's
        stream): every gradient except the CEM's has been enqueued -- on this stream, on the IR lane's, or on a wgrad stream."""
        if self._head_launched:
            return None
        opt, dev = self.optimizer, grad.device
        cur = torch.cuda.current_stream()
        if self._opt_stream is None:
            self._opt_stream = torch.cuda.Stream(device=dev)
        side = self._opt_stream
        opt.upload_hyper()                                   # (current stream; the tail launch reuses the block)
        if not opt._refresh_grads('head'):
            opt.updates -= 1                                 # some gradient is not there yet: leave everything to _update
            opt._steps -= 1
            return None
        side.wait_stream(cur)                                # table + hyper-parameter uploads, lane gradients, BN vectors
        side.wait_stream(self.model._ir_stream(dev))
        for sd in ops.side_streams_in_flight():
            side.wait_stream(sd)
        opt.launch_part('head', side.cuda_stream)
        self._head_launched = True
        return None

    def _plan_parts(self, nparts):
        """Parts by execution order: layer j runs at position pos(j) = its twin leader's index if it is a follower, else j (a pair
        that falls back to the lane form runs LATER than planned, i.e. its gradients come earlier: still safe).  The hook on the
        output of layer t fires when every autograd node created after it has run, i.e. when the gradients of all layers with
        pos > t are complete (the engine pops ready nodes in descending sequence number).  Cuts are placed so that the parts hold
        about equal parameter counts, counted from the end of the forward; only single-tensor Conv / C3 outputs carry hooks."""
        model = self.model
        leader = getattr(model, '_leader_of', {}) if getattr(model, 'twin', False) else {}
        pos = {m.i: leader.get(m.i, m.i) for m in model.model}
        count = {}
        for m in model.model:
            count[pos[m.i]] = count.get(pos[m.i], 0) + sum(p.numel() for p in m.parameters())
        total = sum(count.values())
        ok = {m.i for m in model.model if type(m).__name__ in ('Conv', 'C3') and m.i not in leader}
        cuts, acc, want = [], 0, 1
        for t in sorted(count, reverse=True):             # t: candidate hook layer; acc = parameters of the layers with pos > t
            if want < nparts and acc >= want * total / nparts and t in ok:
                cuts.append(t)
                want += 1
            acc += count[t]
        if not cuts:
            return
        self._cuts = cuts                                  # descending layer indices: part q = layers with cuts[q] < pos <= cuts[q-1]

        def part_of_key(k):
            if not k.startswith('model.'):
                return len(cuts)                           # the CEM and anything outside the layer list: the last part
            p = pos.get(int(k.split('.')[1]), 0)
            for q, t in enumerate(cuts):
                if p > t:
                    return q
            return len(cuts)
        self.optimizer.set_parts(part_of_key, len(cuts) + 1)
        self._part_hooks = {t: self._make_part_hook(q) for q, t in enumerate(cuts)}

    def _make_part_hook(self, part):
        def hook(grad):
            if not self._parts_on or part < self._parts_launched:
                return None
            opt = self.optimizer
            cur = torch.cuda.current_stream()
            if self._opt_stream is None:
                self._opt_stream = torch.cuda.Stream(device=grad.device)
            side = self._opt_stream
            side.wait_stream(cur)                            # lane gradients, BatchNorm vectors
            side.wait_stream(self.model._ir_stream(grad.device))
            for sd in ops.side_streams_in_flight():          # weight gradients of the layers already walked
                side.wait_stream(sd)
            with torch.cuda.stream(side):                    # table and hyper-parameter uploads behind the earlier parts' launches
                if self._parts_launched == 0:
                    opt.upload_hyper()
                if not opt.refresh_upto(part):
                    if self._parts_launched == 0:            # a gradient is not there yet: leave everything to _update
                        opt.updates -= 1
                        opt._steps -= 1
                    return None
                opt.launch_parts(self._parts_launched, part, side.cuda_stream)
            self._parts_launched = part + 1
            return None
        return hook

    def _named_params(self):
        if getattr(self, '_np_cache', None) is None:
            self._np_cache = [(n, p) for n, p in self.model.named_parameters() if p.requires_grad]
        return self._np_cache

    def _update(self, in_capture=False):
        """train.py:799-804: optimizer step, zero_grad, EMA."""
        if self.fused:
            if in_capture:
                self.optimizer.launch()          # hyper-parameters are uploaded outside the graph, before each replay
            elif self._head_launched:            # the early optimizer has stepped everything but the tail during backward
                self.optimizer.launch_part('tail')
                torch.cuda.current_stream().wait_stream(self._opt_stream)
                self._head_launched = False
            elif getattr(self, '_parts_on', False) and self._parts_launched > 0:
                # the rest (the stems, the CEM) behind the parts already launched: same stream order as their table uploads
                torch.cuda.current_stream().wait_stream(self._opt_stream)
                assert self.optimizer.refresh_upto(len(self.optimizer._parts) - 1), 'a parameter of an optimiser group has no gradient'
                self.optimizer.launch_parts(self._parts_launched, len(self.optimizer._parts) - 1)
                self._parts_launched = 0
            else:
                self.optimizer.step()
        else:
            self.optimizer.step()
            if self.ema is not None:
                self.ema.update(self.model)
        if self.fused and self.ema is not None:
            self.ema.updates = self.optimizer.updates                                   # (what train.py:892 stores as 'updates')
        if self.reducer is not None:
            self.reducer.zero(keep_grads=in_capture)                                    # grads are views of flat buckets
        else:
            self.optimizer.zero_grad(set_to_none=True)
            if not self.fused:                  # parameters in no group (pos_emb, sobel_factor: SURVEY.md §9) still receive
                for _, p in self._named_params():   # gradients; torch's optimizer only clears its own
                    p.grad = None

    def step(self, imgs_u8, targets):
        if self.use_graph:
            return self._graph_step(imgs_u8, targets)
        if self.reducer is not None:
            self.reducer.enabled = True          # (a captured step had switched the grad-ready hooks off)
        loss, items = self._body(imgs_u8, targets)
        self.ni += 1
        if self.ni % self.accumulate == 0:
            self._update()
        return loss, items

    # ---- checkpoint / resume (train.py:881-899 writes, 521-531 + 603-615 read) -----------------------------------------------
    def checkpoint(self):
        """The reference's checkpoint dictionary for this trainer: whole model objects as train.py stores them (models/experimental.
        attempt_load reads them back), the EMA update count, the optimizer state in torch.optim.SGD's format."""
        from copy import deepcopy
        torch.cuda.synchronize()
        return {'model': deepcopy(self.model), 'ema': deepcopy(self.ema.ema) if self.ema is not None else None,
                'updates': self.ema.updates if self.ema is not None else 0,
                'optimizer': deepcopy(self.optimizer.state_dict()),      # (state_dict() hands out the live momentum buffers)
                'ni': self.ni}

    def resume(self, ckpt):
        """Continue from checkpoint(): weights and buffers IN PLACE (the kernels' pointer tables stay valid), EMA, momenta, counters."""
        assert self._graph is None, 'resume before the step graph is captured'
        with torch.no_grad():
            for dst, src in ((self.model, ckpt['model']), (self.ema.ema if self.ema is not None else None, ckpt.get('ema'))):
                if dst is None or src is None:
                    continue
                ssd = src.state_dict()
                for k, v in dst.state_dict().items():
                    v.copy_(ssd[k])
        self.optimizer.load_state_dict(ckpt['optimizer'])
        if self.ema is not None:
            self.ema.updates = ckpt.get('updates', 0)
            if self.fused:
                self.optimizer.updates = self.ema.updates
        self.ni = ckpt.get('ni', 0)

    # ---- whole-step hipGraph -----------------------------------------------------------------------------------------------
    def _graph_step(self, imgs_u8, targets):
        if self._graph is None:
            self._capture(imgs_u8, targets)
        else:
            assert imgs_u8.shape == self._imgs.shape and targets.shape == self._targets.shape, \
                'graph mode: fixed batch shape and (padded) target count'
        self._imgs.copy_(imgs_u8, non_blocking=True)
        self._targets.copy_(targets, non_blocking=True)
        self.optimizer.upload_hyper()
        self._graph.replay()
        if self.reducer is not None and not self._graph_has_collectives:
            # data parallel over torch.distributed: the graph holds forward+backward only (gradients land in the reducer's
            # flat buckets); the collectives and the one-launch optimizer follow on the same stream
            self.reducer.reduce_now()
            self._update(in_capture=True)
        self.ni += 1
        return self._loss, self._items

    def _capture(self, imgs_u8, targets):
        # Which stream forks survive a capture (tools/capture_probe.py, profiles/r02_capture_probe.txt): hipStreamEndCapture of
        # ROCm 7.2 segfaults whenever a stream forked from a NON-origin capturing stream (IR lane -> its wgrad stream) is joined
        # back into that stream; forks from the origin stream are fine.  With twin launches (models/yolo_test.py) both backbones
        # run on the origin stream, so the wgrad side streams fork from the origin only and the captured step KEEPS the
        # dgrad || wgrad overlap.  Only when the IR lane stream carries work of its own (MMIDET_TWIN=0, the bf16 storage mode, a
        # graph whose backbones do not pair up) does the capture fall back to weight gradients on the lanes' own streams.
        prev = ops.OVERLAP_WGRAD
        try:
            self._capture_locked(imgs_u8, targets)
        finally:
            ops.OVERLAP_WGRAD = prev

    def _capture_locked(self, imgs_u8, targets):
        self._imgs, self._targets = imgs_u8.clone(), targets.clone()
        side = torch.cuda.Stream(device=imgs_u8.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # warm-up off the default stream: allocations, lazy inits,
            for _ in range(2):                              # momentum-buffer initialisation (first-step flag)
                self._body(self._imgs, self._targets)
                self._update()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if (self.model.two_streams and getattr(self.model, '_ir_used', True)
                and not getattr(self, '_capture_keeps_wgrad_overlap', False)):   # (tools/capture_probe.py)
            ops.OVERLAP_WGRAD = False
        self._graph = torch.cuda.CUDAGraph()
        if self.reducer is not None:
            self.reducer.enabled = False                   # no collectives inside the capture: see _graph_step
        # The library's own communicator (ddp.init_native_comm) enqueues RCCL on the capturing stream like any kernel: the bucket
        # all-reduces and the optimizer are part of the graph, a step is ONE replay.  Over torch.distributed they stay
        # outside: ProcessGroupNCCL's watchdog thread polls events of finished collectives, and under the default (global)
        # capture error mode that query aborts the process while this thread is capturing.
        native = self.reducer is not None and getattr(self.reducer, 'native', False)
        self._graph_has_collectives = native
        if self.reducer is not None and not native:
            # The warm-up steps above issued collectives; the watchdog polls the events of finished ones every 100 ms until it has
            # retired them.  Such a query from its thread while this thread captures is legal in thread_local mode, yet ONE full
            # `-m gpu` run of round 4 died with SIGABRT inside the capture of tests/test_step_gpu.py::
            # test_data_parallel_reducer_corner_cases (no message, autograd thread in a weight-gradient launch; five other runs of
            # the same code passed).  Let the watchdog finish its list before the capture begins: nothing is in flight afterwards.
            time.sleep(0.3)
        mode = 'thread_local' if (self.reducer is not None and not native) else 'global'
        with torch.cuda.graph(self._graph, capture_error_mode=mode):
            self._loss, self._items = self._body(self._imgs, self._targets, reduce=False)
            if native:
                self.reducer.reduce_now()
            if self.reducer is None or native:
                self._update(in_capture=True)
