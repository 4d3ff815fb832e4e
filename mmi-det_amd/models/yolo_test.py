"""Two-stream MMI-Det model on the MI355X-native op set: drop-in for the reference's models/yolo_test.py
(Model 77-273, Detect 29-73, parse_model 548-639) as used by train.py:550,788 / test.py:123 / detect_twostream.py:89.

API kept: Model(cfg, ch=3, nc=None, anchors=None); model(x_rgb, x_ir, augment=False, profile=False) -> (det, Combine_loss);
model.model[-1] is Detect with .nl .na .nc .no .stride .anchors .anchor_grid .m; layers carry .i .f .type .np;
model.save/.stride/.names/.yaml; aux attributes ContrastiveValue/SSIMloss/PTLoss/Entropy_loss/Combine_loss; identical
state_dict keys.  Inputs are NCHW (possibly strided views, train.py:744-745); internally everything is NHWC.
Reference behaviours deliberately NOT reproduced: per-forward prints (yolo_test.py:253,269 -> device syncs).
"""
import logging
import math
import os
from copy import deepcopy
from pathlib import Path

import torch
import torch.nn as nn

from mmidet_hip import alloc
from mmidet_hip import fusion_ops as F2
from mmidet_hip import ops
from mmidet_hip import twin_ops as T2
from models.common import (C3, GPT, SPP, Add, Add2, AdaptiveModule3, Bottleneck, Concat, Conv, Focus, FusedTokens,
                           GPT1_fourier, _holder_conv)
from utils.autoanchor import check_anchor_order
from utils.general import make_divisible
from utils.torch_utils import fuse_conv_and_bn, initialize_weights, model_info

logger = logging.getLogger(__name__)


def _tensors_of(obj):
    if torch.is_tensor(obj):
        return [obj]
    if isinstance(obj, FusedTokens):
        return list(obj.maps)
    if isinstance(obj, (list, tuple)):
        return [t for o in obj for t in _tensors_of(o)]
    return []


class Lane:
    """Placeholder for one lane (0 = RGB, 1 = IR) of the twin tensor a layer pair produced (Model._twin_out[leader]): what the
    saved-output list holds for layers that ran as a twin launch.  Twin-aware consumers take the twin tensor itself; anything
    else gets the lane as an (N,H,W,C) view through Model._single()."""
    __slots__ = ('leader', 'g')

    def __init__(self, leader, g):
        self.leader, self.g = leader, g


class Upsample2x(nn.Upsample):
    """nn.Upsample(None, 2, 'nearest') of the YAML head, on NHWC."""

    fan_skip = True

    def forward(self, x, skip=False, dest=None):
        assert self.mode == 'nearest' and float(self.scale_factor) == 2.0
        return ops.upsample2x(x, skip, dest)


class Detect(nn.Module):
    stride = None
    export = False

    def __init__(self, nc=80, anchors=(), ch=()):
        super().__init__()
        self.nc, self.no = nc, nc + 5
        self.nl, self.na = len(anchors), len(anchors[0]) // 2
        self.grid = [torch.zeros(1)] * self.nl
        a = torch.tensor(anchors).float().view(self.nl, -1, 2)
        self.register_buffer('anchors', a)
        self.register_buffer('anchor_grid', a.clone().view(self.nl, 1, -1, 1, 1, 2))
        self.m = nn.ModuleList(_holder_conv(x, self.no * self.na, 1, 1, bias=True) for x in ch)

    def forward(self, x):
        x = list(x)
        self.training |= self.export
        for i in range(self.nl):
            y = ops.conv_bias(x[i], self.m[i].weight, self.m[i].bias, 1)         # (B,ny,nx,na*no)
            if self.training and os.environ.get('MMIDET_HEAD_VIEW', '1') != '0':
                # the NHWC output of the 1x1 conv IS the permuted layout: (B,na,ny,nx,no) as a strided view, no copy either way
                # (the loss kernels read it in place: utils/loss.py; models/yolo_test.py:54-55 of the reference copies here)
                b, ny, nx, _ = y.shape
                x[i] = y.view(b, ny, nx, self.na, self.no).permute(0, 3, 1, 2, 4)
            else:
                x[i] = ops.head_permute(y, self.na)                               # (B,na,ny,nx,no) contiguous
        if self.training:
            return x
        # eval (models/yolo_test.py:57-68): sigmoid + grid/anchor decode of every level straight into the cat buffer
        z = ops.detect_decode([t.detach() for t in x], self.stride.tolist(), self.anchor_grid, self.no)
        return z, x

    @staticmethod
    def _make_grid(nx=20, ny=20):
        yv, xv = torch.meshgrid([torch.arange(ny), torch.arange(nx)], indexing='ij')
        return torch.stack((xv, yv), 2).view((1, 1, ny, nx, 2)).float()


class Model(nn.Module):
    def __init__(self, cfg='yolov5s.yaml', ch=3, nc=None, anchors=None):
        super().__init__()
        if isinstance(cfg, dict):
            self.yaml = cfg
        else:
            import yaml
            self.yaml_file = Path(cfg).name
            with open(cfg) as f:
                self.yaml = yaml.safe_load(f)
        self.Enhance = AdaptiveModule3(in_channels=int(ch), out_channels=int(ch))
        ch = self.yaml['ch'] = self.yaml.get('ch', ch)
        if nc and nc != self.yaml['nc']:
            logger.info("Overriding model.yaml nc=%s with nc=%s", self.yaml['nc'], nc)
            self.yaml['nc'] = nc
        if anchors:
            logger.info('Overriding model.yaml anchors with anchors=%s', anchors)
            self.yaml['anchors'] = round(anchors)
        self.model, self.save = parse_model(deepcopy(self.yaml), ch=[ch])
        self.names = [str(i) for i in range(self.yaml['nc'])]
        m = self.model[-1]
        if isinstance(m, Detect):
            m.stride = torch.tensor([8.0, 16.0, 32.0])            # hard-coded by the reference (yolo_test.py:127)
            m.anchors /= m.stride.view(-1, 1, 1)
            check_anchor_order(m)
            self.stride = m.stride
            self._initialize_biases()
        initialize_weights(self)
        self.ContrastiveValue = self.SSIMloss = self.PTLoss = self.Entropy_loss = self.Combine_loss = torch.zeros(0)
        self.two_streams = os.environ.get('MMIDET_TWO_STREAMS', '1') != '0'
        # activation storage of the backbone / neck: 'f32' (parity mode, the default) or 'bf16' (opt-in AMP-like mode, SURVEY.md
        # §8 f-4: bf16 maps in HBM, fp32 weights / statistics / loss; the images, the CEM and Focus' first conv stay fp32)
        self.storage = os.environ.get('MMIDET_STORAGE', 'f32')
        self._plan_lanes()

    def _plan_lanes(self):
        """Static lane plan: lane[i] is True for layers of the IR backbone (fed by `from=-4` and by single-input layers /
        Add2 that continue an IR tensor); everything that mixes the streams stays on the caller's stream."""
        lanes, srcs = [], []
        for m in self.model:
            i, f = m.i, m.f
            if f == -4:
                lanes.append(True)
                srcs.append([])
                continue
            fl = [f] if isinstance(f, int) else list(f)
            src = [(i - 1 if j == -1 else (j if j >= 0 else i + j)) for j in fl]
            src = [j for j in src if j >= 0]
            if isinstance(m, Add2):
                lane = lanes[src[0]]
            elif isinstance(f, int):
                lane = lanes[src[0]] if src else False
            else:
                lane = False
            lanes.append(lane)
            srcs.append(src)
        self._lanes, self._srcs = lanes, srcs
        self._ir_streams = {}
        # Fan-out plan: a saved map with exactly two consumers whose FIRST consumer can hand its input on (`fan_skip`: Conv, the
        # fusion transformers' token pooling, nn.Upsample) is consumed as (output, alias) there; the second consumer reads the
        # alias, so its gradient reaches the first consumer's backward as a second incoming gradient and is added inside that
        # kernel (dgrad epilogue / pool gradient / upsample gradient) instead of by the autograd engine's ATen add.
        cons = {}
        for m in self.model:
            for j in srcs[m.i]:
                cons.setdefault(j, []).append(m.i)
        self._fan_skip = {}
        for j, c in cons.items():
            first = self.model[c[0]]
            if len(c) == 2 and c[0] != c[1] and getattr(first, 'fan_skip', False) and os.environ.get('MMIDET_FAN_SKIP', '1') != '0':
                if isinstance(first.f, int) or isinstance(first, GPT):
                    self._fan_skip.setdefault(c[0], []).append(j)
        self._plan_twins(cons)
        self._plan_concats(cons)

    def _plan_concats(self, cons):
        """Neck Concat layers without the copy (models/common.py:740-748 of the reference): when every input of a Concat comes
        from a layer that can write into a channel slice of a wider buffer (Conv, the nearest up-sampling, Add) and feeds no other
        Concat, the producers write straight into the Concat's buffer and the Concat itself is an alias.
        _cat_plan[producer] = (concat layer, channel offset); _cat_total[concat layer] = channels of its buffer."""
        self._cat_plan, self._cat_total = {}, {}
        if os.environ.get('MMIDET_CAT_DEST', '1') == '0':
            return
        cout = {}
        for m in self.model:          # static output channel counts of the layers that matter here
            src = self._srcs[m.i]
            if isinstance(m, Conv):
                cout[m.i] = m.conv.weight.shape[0]
            elif isinstance(m, (C3, SPP)):
                cout[m.i] = (m.cv3 if isinstance(m, C3) else m.cv2).conv.weight.shape[0]
            elif isinstance(m, Focus):
                cout[m.i] = m.conv.conv.weight.shape[0]
            elif isinstance(m, (Add, Add2, Upsample2x)) and src and src[0] in cout:
                cout[m.i] = cout[src[0]]
            elif isinstance(m, Concat) and all(j in cout for j in src):
                cout[m.i] = sum(cout[j] for j in src)
        for m in self.model:
            if not isinstance(m, Concat):
                continue
            src = self._srcs[m.i]
            ok = len(src) >= 2 and len(set(src)) == len(src) and all(
                j in cout and isinstance(self.model[j], (Conv, Add, Upsample2x)) and j not in self._cat_plan
                and j not in self._leader_of and j not in self._follower_of
                and sum(1 for c in cons.get(j, []) if isinstance(self.model[c], Concat)) == 1 for j in src)
            if ok:
                off = 0
                for j in src:
                    self._cat_plan[j] = (m.i, off)
                    off += cout[j]
                self._cat_total[m.i] = off

    def _plan_twins(self, cons):
        """Twin plan: pair every IR-backbone layer (follower) with the RGB-backbone layer (leader) that is the same module on the
        other stream -- same type, same parameter shapes, and inputs that are themselves such a pair (or the two images) -- so that
        forward_once can run the pair as ONE set of launches over a twin tensor (mmidet_hip/twin_ops.py).  The default YAML pairs
        rows (0,3) (1,4) (2,5) (7,8) (9,11) (10,12) ...; the fusion_add graphs pair row i with row i + 10."""
        self._leader_of, self._follower_of, self._twin_fan = {}, {}, {}
        self.twin = os.environ.get('MMIDET_TWIN', '1') != '0'
        lanes, srcs = self._lanes, self._srcs
        shapes = lambda m: [tuple(p.shape) for p in m.parameters()]   # noqa: E731
        for j, mj in enumerate(self.model):
            if not lanes[j] or not hasattr(mj, 'twin_ok'):
                continue
            for i in range(j):
                mi = self.model[i]
                if lanes[i] or i in self._follower_of or type(mi) is not type(mj) or shapes(mi) != shapes(mj):
                    continue
                si, sj = srcs[i], srcs[j]
                if isinstance(mj, Focus):
                    ok = mj.f == -4 and mi.f == -1 and i == 0
                elif isinstance(mj, Add2):
                    ok = (len(si) == 2 and len(sj) == 2 and self._leader_of.get(sj[0]) == si[0] and si[1] == sj[1]
                          and mi.index == 0 and mj.index == 1)
                else:
                    ok = isinstance(mj.f, int) and len(si) == 1 and len(sj) == 1 and self._leader_of.get(sj[0]) == si[0]
                if ok:
                    self._follower_of[i], self._leader_of[j] = j, i
                    break
        # last layer that reads a pair's output (either lane): without autograd the twin tensor can go once that layer has run
        self._twin_last = {lead: max(cons.get(lead, []) + cons.get(fol, []) + [fol]) for lead, fol in self._follower_of.items()}
        # fan-out plan of the twin tensors (see _plan_lanes): a pair's output with exactly two consuming executions whose first
        # can hand its input on (a Conv pair, a fusion transformer's token pooling) is consumed as (output, alias) there
        for lead, fol in self._follower_of.items():
            execs = []
            for c in sorted(set(cons.get(lead, []) + cons.get(fol, []))):
                e = self._leader_of.get(c, c)
                if e not in execs:
                    execs.append(e)
            if len(execs) == 2 and os.environ.get('MMIDET_FAN_SKIP', '1') != '0':
                first = self.model[execs[0]]
                if isinstance(first, GPT) or (isinstance(first, Conv) and execs[0] in self._follower_of):
                    self._twin_fan[execs[0]] = lead

    def __getstate__(self):
        """Pickling (train.py:881-899 stores whole model objects) and deepcopy (ModelEMA): HIP stream handles stay behind."""
        state = dict(self.__dict__)
        state['_ir_streams'] = {}
        for k in ('_side_streams', '_stats_fork', '_main_stream', '_grad_hooks', '_tail_hook'):
            state.pop(k, None)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        if '_lanes' not in state or '_fan_skip' not in state or '_leader_of' not in state or '_cat_plan' not in state or '_twin_last' not in state:   # an object written by the reference (or an older build): same modules, no launch plan
            self.two_streams = os.environ.get('MMIDET_TWO_STREAMS', '1') != '0'
            self._plan_lanes()

    def _ir_stream(self, device):
        s = self._ir_streams.get(device)
        if s is None:
            s = torch.cuda.Stream(device=device)
            self._ir_streams[device] = s
        return s

    def half(self):
        """Reference callers switch inference to fp16 (test.py:66-68,107; detect_twostream.py:44-45,78-82) and the trainer
        rounds the weights through fp16 once (`model.half().float()`, train.py:680).  The kernels compute in fp32, so here
        half() rounds every floating parameter and buffer to its fp16 value IN PLACE (storage stays fp32: same values as
        the reference's `.half().float()`) and marks the model's I/O as fp16: fp16 images are accepted and the detections
        come back as fp16.  float() clears the mark."""
        with torch.no_grad():
            for t in list(self.parameters()) + list(self.buffers()):
                if t.dtype == torch.float32:
                    t.copy_(t.half().float())
        self._io_half = True
        return self

    def float(self):
        self._io_half = False
        return super().float()

    def forward(self, x, x2, augment=False, profile=False):
        if augment:
            raise NotImplementedError('TTA is dead code in the reference too (yolo_test.py:149 drops x2)')
        io_half = getattr(self, '_io_half', False)
        if x.dtype != torch.float32 or x2.dtype != torch.float32:
            assert x.dtype == torch.float16 and io_half, 'inputs are fp32 (or fp16 after model.half())'
            x, x2 = x.float(), x2.float()
        det, comb = self.forward_once(x, x2, profile)
        if io_half:
            det = [t.half() for t in det] if self.training else (det[0].half(), [t.half() for t in det[1]])
        return det, comb

    def forward_once(self, x, x2, profile=False):
        dev = x.device
        empty = torch.zeros(0, device=dev)
        self.ContrastiveValue = self.SSIMloss = self.PTLoss = self.Entropy_loss = empty
        x = ops.nchw_to_nhwc(x)
        x2 = ops.nchw_to_nhwc(x2)
        # AMP (train.py:706,784 of the reference: fp16 autocast + GradScaler).  The kernels compute in fp32 whatever the autocast
        # context says, so the reference's loop runs the exact path unchanged (tests/test_reference_loop_gpu.py).  The MI355X
        # counterpart of its reduced-precision mode is bf16 STORAGE of the activation maps (same exponent range as
        # fp32, no loss scaling needed, a GradScaler around it is harmless); MMIDET_AMP=bf16 maps an active autocast context to it.
        bf16 = getattr(self, 'storage', 'f32') == 'bf16' or (os.environ.get('MMIDET_AMP', '') == 'bf16' and x.is_cuda
                                                             and torch.is_autocast_enabled())
        # Twin launches (mmidet_hip/twin_ops.py): every layer pair of the two backbones runs as one set of launches over a twin
        # tensor, on the caller's stream.  Layers outside a pair (and every layer when MMIDET_TWIN=0, in the bf16 storage mode or
        # after fuse()) take the lane form: the IR backbone on its own HIP stream, events at every cross-lane hand-off.
        twin = getattr(self, 'twin', False) and x.is_cuda and not bf16 and x.dtype == torch.float32
        if x.is_cuda:
            self._pack_for_twin()                                  # (C3 pairs, q/k/v projections: once per placement of the weights)
        lanes = self._lanes if (self.two_streams and x.is_cuda) else None
        main = torch.cuda.current_stream() if lanes else None
        if lanes:
            ir = self._ir_stream(dev)
            ir.wait_stream(main)
            if not torch.cuda.is_current_stream_capturing():
                x2.record_stream(ir)
            done = {}
        x = self.Enhance(x)                                        # CEM on the RGB stream only
        hook = getattr(self, '_tail_hook', None)
        if hook is not None and x.requires_grad:                   # (TrainStep's early optimizer: see _on_tail_gradient)
            x.register_hook(hook)
        grad = torch.is_grad_enabled()
        fan = self._fan_skip if grad else {}
        tfan = self._twin_fan if grad else {}
        self._twin_out, self._twin_lanes = {}, {}                  # leader index -> twin tensor / its two lane views
        self._stats_fork, self._main_stream = None, (torch.cuda.current_stream() if x.is_cuda else None)
        self._ir_used = False
        tw = self._twin_out
        y = []
        prev = x

        def resolve(m):
            if m.f == -1:
                return prev
            if m.f == -4:
                return x2
            return y[m.f] if isinstance(m.f, int) else [prev if j == -1 else y[j] for j in m.f]

        use_cat = bool(self._cat_plan) and not bf16 and x.is_cuda
        cat_bufs = {}                                              # Concat layer -> ops.Dest holding its buffer

        def cat_dest(m, xin):
            """(Dest, channel offset) when layer m writes into a neck Concat's buffer (_plan_concats), else None."""
            plan = self._cat_plan.get(m.i) if use_cat else None
            if plan is None:
                return None
            k, off = plan
            holder = cat_bufs.get(k)
            if holder is None:
                t = xin[0] if isinstance(xin, (list, tuple)) else xin
                n, h, w = t.shape[:3]
                if isinstance(m, Conv):
                    ks, st = m.conv.kernel_size[0], m.conv.stride[0]
                    h, w = (h + 2 * (ks // 2) - ks) // st + 1, (w + 2 * (ks // 2) - ks) // st + 1
                elif isinstance(m, Upsample2x):
                    h, w = 2 * h, 2 * w
                holder = cat_bufs[k] = ops.Dest(alloc.empty((n, h, w, self._cat_total[k]), dtype=torch.float32, device=dev))
            return (holder, off)

        def twin_of(v):
            """The twin tensor behind a pair of inputs [Lane(L,0), Lane(L,1)] of one pair's output, else None."""
            if (isinstance(v, (list, tuple)) and len(v) == 2 and isinstance(v[0], Lane) and isinstance(v[1], Lane)
                    and v[0].leader == v[1].leader and (v[0].g, v[1].g) == (0, 1) and v[0].leader in tw):
                return v[0].leader
            return None

        twin_last = getattr(self, '_twin_last', {})
        ghooks = getattr(self, '_grad_hooks', None) if grad else None          # TrainStep: optimizer parts launched from backward

        def ghook(i, out):
            if ghooks and i in ghooks and torch.is_tensor(out) and out.requires_grad:
                out.register_hook(ghooks[i])
        for m in self.model:
            i = m.i
            if not grad and tw:                                    # inference: a pair's output lives until its last reader has run
                for L in [L for L in tw if twin_last.get(L, len(self.model)) < i]:
                    del tw[L]
                    self._twin_lanes.pop(L, None)
            if twin and i in self._leader_of and self._leader_of[i] in tw:      # ran with its leader
                prev = Lane(self._leader_of[i], 1)
                y.append(prev if i in self.save else None)
                continue
            xin = resolve(m)
            out = None
            on_ir = bool(lanes and lanes[i])
            j = self._follower_of.get(i) if twin else None
            if j is not None and m.twin_ok(self.model[j]):
                mj = self.model[j]
                if isinstance(m, Focus):
                    if lanes:
                        main.wait_stream(ir)                       # (x2's layout pass ran there)
                    out = m.twin(mj, self._single(xin), x2)
                elif isinstance(m, Add2):
                    lead = xin[0].leader if isinstance(xin[0], Lane) and xin[0].g == 0 and xin[0].leader in tw else None
                    if lead is not None and self._leader_of.get(self._srcs[j][0]) == lead:
                        out = m.twin(mj, tw[lead], xin[1])
                elif isinstance(xin, Lane) and xin.g == 0 and xin.leader in tw:
                    if i in tfan:
                        out, alias = m.twin(mj, tw[xin.leader], skip=True)
                        tw[tfan[i]] = alias
                        self._twin_lanes.pop(tfan[i], None)
                    else:
                        out = m.twin(mj, tw[xin.leader])
                if out is not None:
                    ghook(i, out)
                    tw[i] = out
                    prev = Lane(i, 0)
                    y.append(prev if i in self.save else None)
                    continue
            # ---- not a twin pair (or one that cannot run as such right now): twin-aware consumers of a pair's output, then the lane form
            lead = twin_of(xin) if twin else None
            if lead is not None and isinstance(m, GPT):
                if isinstance(m, GPT1_fourier):
                    T = tw[lead]
                    if i in tfan:
                        (x, pt), alias = m(T, skip=True)
                        tw[tfan[i]] = alias[0]
                        self._twin_lanes.pop(tfan[i], None)
                    else:
                        x, pt = m(T)
                    st = self._fusion_stats(T[..., 0, :], T[..., 1, :], m.last_tokens)
                    self.SSIMloss, self.Entropy_loss, self.ContrastiveValue, self.PTLoss = st[0], st[1], st[2], pt
                elif i in tfan:
                    x, alias = m(tw[lead], skip=True)
                    tw[tfan[i]] = alias[0]
                    self._twin_lanes.pop(tfan[i], None)
                else:
                    x = m(tw[lead])
                prev = x
                y.append(x if i in self.save else None)
                continue
            if lead is not None and isinstance(m, Add):
                prev = x = m.twin(tw[lead], dest=cat_dest(m, tw[lead]))
                y.append(x if i in self.save else None)
                continue
            from_twin = twin and any(isinstance(v, Lane) for v in (xin if isinstance(xin, (list, tuple)) else [xin]))
            xin = self._single(xin)
            x = xin
            if lanes:
                st_ = ir if on_ir else main
                self._ir_used = self._ir_used or on_ir
                if on_ir and from_twin:
                    ir.wait_stream(main)                           # twin launches run on the caller's stream
                for k in self._srcs[i]:                            # hand-offs from the other lane
                    if k in done and done[k][1] is not st_:
                        st_.wait_event(done[k][0])
                        if not torch.cuda.is_current_stream_capturing():   # (a graph's private pool replays fixed addresses)
                            for t in _tensors_of(self._single(y[k]) if y[k] is not None else xin):   # unsaved => k is the previous layer
                                t.record_stream(st_)
                ctx = torch.cuda.stream(st_)
                ctx.__enter__()
            if m.f == -4:
                x = m(x2)
                if bf16:
                    x = ops.cast(x, torch.bfloat16)
            elif isinstance(m, GPT1_fourier):
                in_rgb, in_ir = x[0], x[1]
                if i in fan:
                    (x, pt), alias = m(x, skip=True)
                    for k, a in zip(self._srcs[i], alias):
                        y[k] = a
                else:
                    x, pt = m(x)
                st = self._fusion_stats(in_rgb, in_ir, m.last_tokens)
                self.SSIMloss, self.Entropy_loss, self.ContrastiveValue, self.PTLoss = st[0], st[1], st[2], pt
            else:
                kw = {}
                if isinstance(m, Concat):
                    if i in cat_bufs:                              # every input already is its slice of the buffer
                        kw['holder'] = cat_bufs.pop(i)
                elif use_cat and i in self._cat_plan:
                    kw['dest'] = cat_dest(m, x)
                if i in fan and not any(isinstance(y[k], Lane) for k in self._srcs[i]):   # (output, alias of the input for its second consumer)
                    x, alias = m(x, skip=True, **kw)
                    if isinstance(alias, (list, tuple)):
                        for k, a in zip(self._srcs[i], alias):
                            y[k] = a
                    else:
                        y[self._srcs[i][0]] = alias
                else:
                    x = m(x, **kw)
                if bf16 and i == 0:
                    x = ops.cast(x, torch.bfloat16)          # behind the RGB stem (Focus): everything downstream is bf16
            if lanes:
                ev = torch.cuda.Event()
                ev.record(st_)
                done[i] = (ev, st_)
                ctx.__exit__(None, None, None)
            ghook(i, x)
            prev = x
            y.append(x if i in self.save else None)
        if lanes:
            main.wait_stream(ir)
        if self._stats_fork is not None:                           # the CBM / IGM statistics ran beside the layers after the FFM
            torch.cuda.current_stream().wait_stream(self._stats_fork)
            self._stats_fork = None
        self._twin_out, self._twin_lanes = {}, {}
        self.Combine_loss = self.SSIMloss                          # yolo_test.py:266-268 (detached SSIM term)
        return x, self.Combine_loss

    def _fusion_stats(self, in_rgb, in_ir, tokens):
        """CBM + IGM statistics (yolo_test.py:338-486 of the reference: SSIM, entropy, contrastive value): values only, detached in the
        reference, and nobody reads them before the loss -- so the kernel (LDS-atomic bound: three 256-bin histograms over 52 M
        values, ~0.25 ms at 16 x 160 x 160 x 128) leaves the critical path: it runs on a stream of its own beside the layers that
        follow the FFM and is joined at the end of the forward.  Only when the caller's stream is the forward's main stream (a
        fork from a lane stream would be the nested fork hipStreamEndCapture cannot take: tools/capture_probe.py)."""
        with torch.no_grad():
            cur = torch.cuda.current_stream() if in_rgb.is_cuda else None
            if cur is None or os.environ.get('MMIDET_STATS_STREAM', '1') == '0' or cur != getattr(self, '_main_stream', None):
                return F2.fusion_stats(in_rgb, in_ir, tokens)
            side = self._side_streams.get(in_rgb.device) if hasattr(self, '_side_streams') else None
            if side is None:
                if not hasattr(self, '_side_streams'):
                    self._side_streams = {}
                side = self._side_streams[in_rgb.device] = torch.cuda.Stream(device=in_rgb.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                st = F2.fusion_stats(in_rgb, in_ir, tokens)
            self._stats_fork = side
            return st

    def _pack_for_twin(self):
        """C3's cv1 | cv2 run as one GEMM -- and the two backbones' C3 as one twin launch -- only when their parameters share
        buffers (ops.pack_pair: same Parameters, state_dict keys and values).  TrainStep packs before it builds its optimizer;
        a caller that drives the model itself (the reference's train.py) gets the packing at the first TRAINING forward on the
        device.  Re-seating parameters is a side effect only the owner of their addresses may trigger, so it never happens for
        a model that is not being trained (eval mode, or no parameter that requires a gradient: the EMA copy, whose addresses
        FusedSGDEMA's pointer table and ModelEMA's pair list have cached -- ModelEMA packs its copy itself, at construction),
        and never while a data-parallel reducer holds gradient slots keyed on the parameter addresses.  An unpacked model runs
        the same kernels on the lane / gathered forms."""
        if not self.training or ops.GRAD_SLOTS or torch.cuda.is_current_stream_capturing():
            return
        first = last = None
        for p in self.parameters():                                # placement key: no _version (every optimizer step bumps it)
            if p.is_cuda:
                last = p
                if first is None:
                    first = p
        if first is None or not first.requires_grad:
            return
        key = (first.data_ptr(), last.data_ptr())
        if getattr(self, '_pack_key', None) == key:
            return
        self.pack_parameters()
        self._pack_key = (first.data_ptr(), last.data_ptr())

    def pack_parameters(self):
        """Explicit form of the above for whoever owns the parameter addresses (TrainStep, ModelEMA for its copy)."""
        if ops.PACK_C3:
            ops.pack_pair(self)
        if os.environ.get('MMIDET_PACK_QKV', '1') != '0':
            F2.pack_qkv(self)          # q/k/v projections of the fusion transformers as one GEMM each way
        return self

    def _single(self, v):
        """Inputs of a single-lane layer: Lane placeholders become (N,H,W,C) views of their twin tensor (one autograd node per
        twin tensor: the two lanes' gradients come back together)."""
        if isinstance(v, Lane):
            views = self._twin_lanes.get(v.leader)
            if views is None:
                views = self._twin_lanes[v.leader] = T2.lanes(self._twin_out[v.leader])
            return views[v.g]
        if isinstance(v, (list, tuple)):
            return [self._single(t) for t in v]
        return v

    def _initialize_biases(self, cf=None):
        m = self.model[-1]
        for mi, s in zip(m.m, m.stride):
            b = mi.bias.view(m.na, -1)
            b.data[:, 4] += math.log(8 / (640 / s) ** 2)
            b.data[:, 5:] += math.log(0.6 / (m.nc - 0.99)) if cf is None else torch.log(cf / cf.sum())
            mi.bias = torch.nn.Parameter(b.view(-1), requires_grad=True)

    def fuse(self):
        """Fold every Conv's BatchNorm into its convolution (models/yolo_test.py:304-312, utils/torch_utils.py:181-201).
        Only `Conv` modules are folded, exactly as the reference's `type(m) is Conv` test (the CEM's bare conv+BN pairs
        stay as they are)."""
        from models.common import Conv as _Conv
        for m in self.model.modules():
            if type(m) is _Conv and hasattr(m, 'bn'):
                m.conv = fuse_conv_and_bn(m.conv, m.bn)
                delattr(m, 'bn')
                m.forward = m.fuseforward
        self._fan_skip = {}                                        # (fuseforward has no hand-on form; inference does not need one)
        self._twin_fan, self.twin = {}, False
        self._cat_plan, self._cat_total = {}, {}                   # (fuseforward has no destination form either)
        self.info()
        return self

    def info(self, verbose=False, img_size=640):
        model_info(self, verbose, img_size)


_NAMES = dict(Conv=Conv, Bottleneck=Bottleneck, C3=C3, SPP=SPP, Focus=Focus, Concat=Concat, Add=Add, Add2=Add2, GPT=GPT,
              GPT1_fourier=GPT1_fourier, Detect=Detect)
_LITERALS = {'None': None, 'True': True, 'False': False}


def parse_model(d, ch):
    """[from, number, module, args] rows -> nn.Sequential + save list; same grammar and channel/depth rules as the
    reference, module names resolved through a whitelist instead of eval()."""
    anchors, nc, gd, gw = d['anchors'], d['nc'], d['depth_multiple'], d['width_multiple']
    na = (len(anchors[0]) // 2) if isinstance(anchors, list) else anchors
    no = na * (nc + 5)
    layers, save, c2 = [], [], ch[-1]
    for i, (f, n, m, args) in enumerate(d['backbone'] + d['head']):
        args = [({'nc': nc, 'anchors': anchors}.get(a, _LITERALS.get(a, a)) if isinstance(a, str) else a) for a in args]
        n = max(round(n * gd), 1) if n > 1 else n
        if m == 'nn.Upsample':
            cls, tname = Upsample2x, 'torch.nn.modules.upsampling.Upsample'
            c2 = ch[f]
        else:
            if m not in _NAMES:
                raise ValueError('module %r is outside the two-stream hot path' % m)
            cls = _NAMES[m]
            tname = 'Detect' if cls is Detect else 'models.common.' + m
            if cls in (Conv, Bottleneck, SPP, Focus, C3):
                c1, c2 = (3 if cls is Focus else ch[f]), args[0]
                if c2 != no:
                    c2 = make_divisible(c2 * gw, 8)
                args = [c1, c2, *args[1:]]
                if cls is C3:
                    args.insert(2, n)
                    n = 1
            elif cls is Concat:
                c2 = sum(ch[x] for x in f)
            elif cls is Add:
                c2 = ch[f[0]]
                args = [c2]
            elif cls is Add2:
                c2 = ch[f[0]]
                args = [c2, args[1]]
            elif cls is GPT:
                c2 = ch[f[0]]
                args = [c2]
            elif cls is GPT1_fourier:
                c2 = args[0]                                        # not width-scaled by the reference (yolo_test.py:607-609)
                args = [c2]
            elif cls is Detect:
                args.append([ch[x] for x in f])
                if isinstance(args[1], int):
                    args[1] = [list(range(args[1] * 2))] * len(f)
        m_ = nn.Sequential(*[cls(*args) for _ in range(n)]) if n > 1 else cls(*args)
        m_.i, m_.f, m_.type = i, f, tname
        m_.np = sum(x.numel() for x in m_.parameters())
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        layers.append(m_)
        if i == 0:
            ch = []
        ch.append(c2)
    return nn.Sequential(*layers), sorted(save)
