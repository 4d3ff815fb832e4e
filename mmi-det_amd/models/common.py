"""MI355X-native module library for the two-stream MMI-Det graph (drop-in for the reference's models/common.py on the
training hot path).

Same class names, constructor arguments, attribute names and state_dict keys as the reference
(/root/reference/models/common.py: Conv 108, Bottleneck 602, C3 637, SPP 681, Focus 696, Concat 740, AdaptiveModule3
751, EnhanceConv2d 806, Add 914, Add2 924, SelfAttention 1147, myTransformerBlock 1237, GPT 1270, GPT1_fourier 299), but
every forward runs hand-written HIP kernels through the C ABI (mmidet_hip.ops / fusion_ops) on NHWC activations.
Parameters live in ordinary nn.Conv2d / nn.BatchNorm2d / nn.Linear / nn.LayerNorm holders (so optimiser grouping by
isinstance, EMA and checkpoints behave exactly like the reference); the holders' own forward is never called.
Conv weights are stored channels_last, i.e. physically OHWI, which is the layout the MFMA kernels read.
"""
import math

import torch
import torch.nn as nn

from mmidet_hip import alloc
from mmidet_hip import fusion_ops as F2
from mmidet_hip import ops
from mmidet_hip import twin_ops as T2
from mmidet_hip.ops import ACT_LEAKY, ACT_NONE, ACT_SILU


def autopad(k, p=None):
    return k // 2 if p is None else p


def _holder_conv(c1, c2, k, s, bias=False):
    m = nn.Conv2d(c1, c2, k, s, autopad(k), bias=bias)
    m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    return m


def _bn_args(bn):
    return bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked


class Conv(nn.Module):
    """SiLU(BN(conv(x))) as conv-with-statistics-epilogue + one normalise/activate pass."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__()
        assert g == 1, 'grouped convolutions are outside the two-stream hot path'
        self.conv = _holder_conv(c1, c2, k, s)
        self.bn = nn.BatchNorm2d(c2)
        self.act = nn.SiLU() if act is True else (act if isinstance(act, nn.Module) else nn.Identity())

    def _act_id(self):
        if isinstance(self.act, nn.SiLU):
            return ACT_SILU
        if isinstance(self.act, nn.LeakyReLU):
            return ACT_LEAKY
        assert isinstance(self.act, nn.Identity), 'unsupported activation %r' % self.act
        return ACT_NONE

    fan_skip = True      # (yolo_test.forward_once: a saved map with a second consumer is handed on as the alias)

    def forward(self, x, residual=None, skip=False, dest=None):
        """skip=True: also return x as a second output (the Bottleneck shortcut, see ops._ConvBnAct); dest=(ops.Dest, channel):
        write the output into that channel slice of a wider buffer."""
        w, b, rm, rv, nbt = _bn_args(self.bn)
        return ops.conv_bn_act(x, self.conv.weight, w, b, rm, rv, nbt, stride=self.conv.stride[0], act=self._act_id(),
                               residual=residual, training=self.bn.training, eps=self.bn.eps,
                               momentum=self.bn.momentum, skip=skip, dest=dest)

    def twin_ok(self, other):
        """Can this layer and `other` (the same layer of the other backbone) run as one twin launch (mmidet_hip/twin_ops.py)?"""
        co, ci, kh, _ = self.conv.weight.shape
        if kh == 3 and self.conv.stride[0] == 1 and (ci, co) in ((3, 24), (24, 3)):
            return False           # the CEM-shaped direct convolutions (csrc/cem.hip) have no twin form: the pair takes the lane form
        return (type(other) is type(self) and hasattr(self, 'bn') and hasattr(other, 'bn') and self.conv.weight.shape == other.conv.weight.shape
                and self.conv.stride == other.conv.stride and self._act_id() == other._act_id() and self.bn.eps == other.bn.eps
                and self.bn.momentum == other.bn.momentum and self.bn.training == other.bn.training)

    def twin(self, other, x, residual=None, skip=False, dest=None):
        """Both backbones' copies of this layer on a twin tensor (N,H,W,2,C): one GEMM launch, one normalise pass."""
        return T2.conv_bn_act2(x, self, other, residual=residual, skip=skip, dest=dest)

    def fuseforward(self, x, residual=None):
        """After Model.fuse() (models/common.py:124-125): the folded conv carries the bias, BN is gone."""
        return ops.conv_bias_act(x, self.conv.weight, self.conv.bias, stride=self.conv.stride[0], act=self._act_id(),
                                 residual=residual)


class Bottleneck(nn.Module):
    def __init__(self, c1, c2, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_, c2, 3, 1, g=g)
        self.add = shortcut and c1 == c2

    def forward(self, x, dest=None):
        # the residual add rides in cv2's normalise/activate pass; its gradient in cv1's input-gradient GEMM
        if hasattr(self.cv1, 'bn') and hasattr(self.cv2, 'bn'):
            if self.add and ops.SKIP_FUSE:
                h, xs = self.cv1(x, skip=True)
                return self.cv2(h, residual=xs, dest=dest)
            return self.cv2(self.cv1(x), residual=x if self.add else None, dest=dest)
        return self.cv2(self.cv1(x), residual=x if self.add else None)        # after Model.fuse()

    def twin_ok(self, other):
        return type(other) is type(self) and self.add == other.add and self.cv1.twin_ok(other.cv1) and self.cv2.twin_ok(other.cv2)

    def twin(self, other, x, dest=None):
        if self.add and ops.SKIP_FUSE:
            h, xs = self.cv1.twin(other.cv1, x, skip=True)
            return self.cv2.twin(other.cv2, h, residual=xs, dest=dest)
        return self.cv2.twin(other.cv2, self.cv1.twin(other.cv1, x), residual=x if self.add else None, dest=dest)


class C3(nn.Module):
    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*[Bottleneck(c_, c_, shortcut, g, e=1.0) for _ in range(n)])

    def packed(self):
        """True when cv1 / cv2 share buffers (ops.pack_pair) and agree in everything the merged kernel assumes."""
        a, b = self.cv1, self.cv2
        if not (hasattr(a, 'bn') and hasattr(b, 'bn')):
            return False
        key = tuple(t.data_ptr() for t in (a.conv.weight, b.conv.weight, a.bn.weight, b.bn.weight, a.bn.bias, b.bn.bias,
                                           a.bn.running_mean, b.bn.running_mean, a.bn.running_var, b.bn.running_var,
                                           a.bn.num_batches_tracked, b.bn.num_batches_tracked))
        if getattr(self, '_pack_key', None) != key:
            ok = (a.conv.weight.is_cuda and a.conv.weight.shape[0] % 4 == 0 and a._act_id() == b._act_id()
                  and a.bn.eps == b.bn.eps and a.bn.momentum == b.bn.momentum
                  and a.conv.weight.is_contiguous(memory_format=torch.channels_last)
                  and b.conv.weight.is_contiguous(memory_format=torch.channels_last)
                  and all(ops.back_to_back(u, v) for u, v in (
                      (a.conv.weight, b.conv.weight), (a.bn.weight, b.bn.weight), (a.bn.bias, b.bn.bias),
                      (a.bn.running_mean, b.bn.running_mean), (a.bn.running_var, b.bn.running_var),
                      (a.bn.num_batches_tracked, b.bn.num_batches_tracked))))
            self._pack_ok, self._pack_key = ok, key
        return self._pack_ok and a.bn.training == b.bn.training

    def forward(self, x):
        if not self.packed():
            return self.cv3(ops.concat([self.m(self.cv1(x)), self.cv2(x)]))
        # cv1 | cv2 as one GEMM; the concat buffer is written in place by its two producers (models/common.py:650 without the copy)
        a, b = self.cv1, self.cv2
        c_ = a.conv.weight.shape[0]
        cat = ops.Dest(alloc.empty((*x.shape[:-1], 2 * c_), dtype=x.dtype, device=x.device))
        h, b_out = ops.dual_conv_bn_act(x, a.conv.weight, b.conv.weight, a.bn.weight, a.bn.bias, b.bn.weight, b.bn.bias,
                                        a.bn.running_mean, a.bn.running_var, a.bn.num_batches_tracked,
                                        b.bn.num_batches_tracked, a._act_id(), a.bn.training, a.bn.eps, a.bn.momentum,
                                        dest=(cat, c_))
        last = len(self.m) - 1
        for i, blk in enumerate(self.m):
            h = blk(h, dest=(cat, 0) if i == last else None)
        return self.cv3(ops.cat_alias(h, b_out, cat))

    def twin_ok(self, other):
        return (type(other) is type(self) and len(self.m) == len(other.m) and ops.PACK_C3 and self.packed() and other.packed()
                and self.cv1.twin_ok(other.cv1) and self.cv2.twin_ok(other.cv2) and self.cv3.twin_ok(other.cv3)
                and all(a.twin_ok(b) for a, b in zip(self.m, other.m)))

    def twin(self, other, x):
        """Both backbones' C3 on a twin tensor: cv1|cv2 of both lanes as one GEMM, the lanes' concat buffers side by side
        (N,H,W,2,2c_) and written in place by their producers, cv3 of both lanes as one GEMM."""
        c_ = self.cv1.conv.weight.shape[0]
        cat = alloc.empty((*x.shape[:3], 2, 2 * c_), dtype=x.dtype, device=x.device)
        h, b_out = T2.dual_conv_bn_act2(x, self, other, ops.Dest(cat))
        last = len(self.m) - 1
        for i, (ba, bb) in enumerate(zip(self.m, other.m)):
            h = ba.twin(bb, h, dest=ops.Dest(cat[..., :c_]) if i == last else None)
        return self.cv3.twin(other.cv3, ops.cat_alias(h, b_out, ops.Dest(cat)))


class SPP(nn.Module):
    def __init__(self, c1, c2, k=(5, 9, 13)):
        super().__init__()
        assert tuple(k) == (5, 9, 13), 'the SPP kernel cascades 5x5 pools: k must be (5, 9, 13)'
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * (len(k) + 1), c2, 1, 1)
        self.m = nn.ModuleList([nn.MaxPool2d(kernel_size=x, stride=1, padding=x // 2) for x in k])

    def forward(self, x):
        return self.cv2(ops.spp_pool(self.cv1(x)))

    def twin_ok(self, other):
        return type(other) is type(self) and self.cv1.twin_ok(other.cv1) and self.cv2.twin_ok(other.cv2)

    def twin(self, other, x):
        return self.cv2.twin(other.cv2, T2.spp_pool2(self.cv1.twin(other.cv1, x)))


class Focus(nn.Module):
    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__()
        self.conv = Conv(c1 * 4, c2, k, s, p, g, act)

    def forward(self, x):
        return self.conv(ops.space_to_depth(x))

    def twin_ok(self, other):
        return type(other) is type(self) and self.conv.twin_ok(other.conv)

    def twin(self, other, xa, xb):
        """Both stems: the two images' slicings side by side (N,H/2,W/2,2,12), one twin conv."""
        return self.conv.twin(other.conv, T2.space_to_depth2(xa, xb))


class Concat(nn.Module):
    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension  # NCHW dim 1 == the contiguous channel axis of the NHWC activations

    def forward(self, x, holder=None):
        """holder (ops.Dest): the inputs already are the channel slices of this buffer, written there by their producers
        (Model._plan_concats): nothing to copy."""
        if holder is not None:
            return ops.cat_alias_n(list(x), holder)
        return ops.concat(list(x))


class Add(nn.Module):
    def __init__(self, arg):
        super().__init__()
        self.arg = arg

    def forward(self, x, dest=None):
        return ops.add(x[0], x[1], dest)

    def twin(self, x, dest=None):
        return T2.add_lanes(x, dest)


class FusedTokens:
    """Output of GPT / GPT1_fourier: the two 8x8 token maps whose bilinear up-sampling is fused into Add2."""

    def __init__(self, rgb, ir, hw):
        self.maps, self.hw = (rgb, ir), hw

    def __getitem__(self, i):  # materialised (B,H,W,C) map, for callers that index the transformer output directly
        return F2.upsample_only(self.maps[i], *self.hw)


class Add2(nn.Module):
    def __init__(self, c1, index):
        super().__init__()
        self.index = index

    def forward(self, x):
        if isinstance(x[1], FusedTokens):
            return F2.upsample_add(x[0], x[1].maps[self.index])
        return ops.add(x[0], x[1][self.index])

    def twin_ok(self, other):
        return type(other) is type(self) and self.index == 0 and other.index == 1

    def twin(self, other, x, fused):
        """The Add2 pair of a fusion point on the twin tensor of the two streams: lane g + up-sampled transformer map g."""
        if isinstance(fused, FusedTokens):
            return T2.upsample_add2(x, fused.maps[0], fused.maps[1])
        return None


class EnhanceConv2d(nn.Module):
    """Fixed edge-stencil bank x trainable per-output factor + bias (parameters as in the reference)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1, groups=1, bias=True,
                 requires_grad=True):
        super().__init__()
        assert kernel_size == 3 and out_channels % 8 == 0 and groups == 1 and stride == 1 and padding == 1
        bank = ((-1, -2, -1, 0, 0, 0, 1, 2, 1), (-1, 0, 1, -2, 0, 2, -1, 0, 1), (-2, -1, 0, -1, 0, 1, 0, 1, 2),
                (-2, -1, 0, -1, 0, 1, 0, 1, 2), (0, 1, 0, 1, -4, 1, 0, 1, 0), (0, 1, 0, 1, 4, 1, 0, 1, 0),
                (-1, -1, -1, 0, 0, 0, 1, 1, 1), (-1, 0, 1, -1, 0, 1, -1, 0, 1))
        self.bias = nn.Parameter(torch.zeros(out_channels), requires_grad=True) if (bias and requires_grad) else None
        w = torch.stack([torch.tensor(bank[o % 8], dtype=torch.float32).view(1, 3, 3).expand(in_channels, 3, 3)
                         for o in range(out_channels)])
        self.sobel_weight = nn.Parameter(w.contiguous(), requires_grad=False)
        self.sobel_factor = nn.Parameter(torch.ones(out_channels, 1, 1, 1), requires_grad=requires_grad)

    def is_standard_bank(self):
        """True when sobel_weight is the reference's frozen stencil bank (it always is unless a checkpoint overwrote
        it), the bias exists and there are 24 channels: the case csrc/cem.hip evaluates in closed form."""
        key = (self.sobel_weight.data_ptr(), self.sobel_weight._version)
        if getattr(self, '_bank_key', None) != key:
            w = self.sobel_weight.detach()
            ref = EnhanceConv2d(w.shape[1], w.shape[0]).sobel_weight.detach().to(w.device) if w.shape[0] % 8 == 0 else None
            self._bank_ok = bool(ref is not None and w.shape == (24, 24, 3, 3) and self.bias is not None and torch.equal(w, ref))
            self._bank_key = key
        return self._bank_ok

    def forward(self, x):
        # 5 184-element parameter product; the convolution itself is the HIP implicit GEMM
        w = (self.sobel_weight * self.sobel_factor).contiguous(memory_format=torch.channels_last)
        return ops.conv_bias(x, w, self.bias, 1)


class AdaptiveModule3(nn.Module):
    """Contour Enhancement Module."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        c = int(in_channels)
        self.conv2 = _holder_conv(c, c * 8, 3, 1)
        self.bn2 = nn.BatchNorm2d(c * 8)
        self.relu2 = nn.LeakyReLU(0.1)
        self.sobel = EnhanceConv2d(c * 8, c * 8)
        self.conv3 = _holder_conv(c * 8, c, 3, 1)
        self.bn3 = nn.BatchNorm2d(c)
        self.relu3 = nn.LeakyReLU(0.1)

    def forward(self, x):
        if (ops.CEM_FUSED and self.sobel.is_standard_bank() and x.dtype == torch.float32 and x.shape[-1] == 3
                and self.bn2.training == self.bn3.training and self.bn2.eps == self.bn3.eps and self.bn2.momentum == self.bn3.momentum
                and isinstance(self.relu2, nn.LeakyReLU) and self.relu2.negative_slope == 0.1):
            # one forward kernel behind a statistics pre-pass (csrc/cem.hip::cem_fused_fwd_kernel)
            return ops.cem_fused(x, self.conv2.weight, self.bn2, self.sobel.sobel_factor, self.sobel.bias, self.conv3.weight, self.bn3)
        w, b, rm, rv, nbt = _bn_args(self.bn2)
        r = ops.conv_bn_act(x, self.conv2.weight, w, b, rm, rv, nbt, 1, ACT_LEAKY, None, self.bn2.training, self.bn2.eps,
                            self.bn2.momentum)
        if self.sobel.is_standard_bank():
            # r + EnhanceConv2d(r) without ever forming the 24x24x3x3 conv: 8 fixed stencils of the channel-sum map
            t = ops.sobel_add(r, self.sobel.sobel_factor, self.sobel.bias)
        else:
            t = ops.add(r, self.sobel(r))
        w, b, rm, rv, nbt = _bn_args(self.bn3)
        return ops.conv_bn_act(t, self.conv3.weight, w, b, rm, rv, nbt, 1, ACT_LEAKY, x, self.bn3.training, self.bn3.eps,
                               self.bn3.momentum)


class SelfAttention(nn.Module):
    def __init__(self, d_model, d_k, d_v, h, attn_pdrop=.1, resid_pdrop=.1):
        super().__init__()
        assert d_k % h == 0
        self.d_model, self.h = d_model, h
        self.d_k = self.d_v = d_model // h
        self.que_proj = nn.Linear(d_model, h * self.d_k)
        self.key_proj = nn.Linear(d_model, h * self.d_k)
        self.val_proj = nn.Linear(d_model, h * self.d_v)
        self.out_proj = nn.Linear(h * self.d_v, d_model)
        self.attn_drop = nn.Dropout(attn_pdrop)
        self.resid_drop = nn.Dropout(resid_pdrop)

    def forward(self, x):
        q = ops.linear(x, self.que_proj.weight, self.que_proj.bias)
        k = ops.linear(x, self.key_proj.weight, self.key_proj.bias)
        v = ops.linear(x, self.val_proj.weight, self.val_proj.bias)
        o = F2.attention(q, k, v, self.h, self.attn_drop.p, self.training)
        o = ops.linear(o, self.out_proj.weight, self.out_proj.bias)
        return F2.dropout_add(o, None, self.resid_drop.p, self.training)


class myTransformerBlock(nn.Module):
    def __init__(self, d_model, d_k, d_v, h, block_exp, attn_pdrop, resid_pdrop):
        super().__init__()
        self.ln_input = nn.LayerNorm(d_model)
        self.ln_output = nn.LayerNorm(d_model)
        self.sa = SelfAttention(d_model, d_k, d_v, h, attn_pdrop, resid_pdrop)
        self.mlp = nn.Sequential(nn.Linear(d_model, block_exp * d_model), nn.GELU(),
                                 nn.Linear(block_exp * d_model, d_model), nn.Dropout(resid_pdrop))

    def forward(self, x):
        # one autograd node per block (mmidet_hip/fusion_ops.py:_TransformerBlock): dropout + residual add and GELU ride
        # in the Linear epilogues, the residual gradient and the dropout mask in the LayerNorm backward
        li, lo, sa = self.ln_input, self.ln_output, self.sa
        params = (li.weight, li.bias, sa.que_proj.weight, sa.que_proj.bias, sa.key_proj.weight, sa.key_proj.bias,
                  sa.val_proj.weight, sa.val_proj.bias, sa.out_proj.weight, sa.out_proj.bias, lo.weight, lo.bias,
                  self.mlp[0].weight, self.mlp[0].bias, self.mlp[2].weight, self.mlp[2].bias)
        return F2.transformer_block(x, sa.h, (sa.attn_drop.p, sa.resid_drop.p, self.mlp[3].p), (li.eps, lo.eps), params,
                                    self.training)


def _init_gpt(module):
    if isinstance(module, nn.Linear):
        module.weight.data.normal_(mean=0.0, std=0.02)
        if module.bias is not None:
            module.bias.data.zero_()
    elif isinstance(module, nn.LayerNorm):
        module.bias.data.zero_()
        module.weight.data.fill_(1.0)


class GPT(nn.Module):
    """Cross-modal fusion transformer: pool both streams to 2x8x8 tokens, 8 pre-LN blocks, ln_f, hand the token maps to
    the Add2 layers (which up-sample and add in one kernel)."""

    def __init__(self, d_model, h=8, block_exp=4, n_layer=8, vert_anchors=8, horz_anchors=8, embd_pdrop=0.1,
                 attn_pdrop=0.1, resid_pdrop=0.1):
        super().__init__()
        assert vert_anchors == 8 and horz_anchors == 8
        self.n_embd, self.vert_anchors, self.horz_anchors = d_model, vert_anchors, horz_anchors
        self.pos_emb = nn.Parameter(torch.zeros(1, 2 * vert_anchors * horz_anchors, d_model))
        self.trans_blocks = nn.Sequential(*[myTransformerBlock(d_model, d_model, d_model, h, block_exp, attn_pdrop,
                                                               resid_pdrop) for _ in range(n_layer)])
        self.ln_f = nn.LayerNorm(d_model)
        self.drop = nn.Dropout(embd_pdrop)
        self.avgpool = nn.AdaptiveAvgPool2d((vert_anchors, horz_anchors))
        self.apply(_init_gpt)

    def _transform(self, tok):
        x = F2.dropout_add(tok, self.pos_emb, self.drop.p, self.training)
        x = self.trans_blocks(x)
        return F2.layernorm(x, self.ln_f.weight, self.ln_f.bias, self.ln_f.eps)

    fan_skip = True      # forward(x, skip=True) -> (output, [rgb alias, ir alias]) for the maps' other consumers (yolo_test.forward_once)

    def _pool(self, x, skip):
        """tokens (B,128,C), alias list (or None), (H, W): x is the pair (rgb, ir) of maps or their twin tensor (N,H,W,2,C)."""
        if torch.is_tensor(x) and x.dim() == 5:
            pooled = T2.pool_tokens2(x, skip)
            if skip:
                return pooled[0], [pooled[1]], tuple(x.shape[1:3])
            return pooled, None, tuple(x.shape[1:3])
        rgb, ir = x[0], x[1]
        assert rgb.shape[0] == ir.shape[0]
        pooled = F2.pool_tokens(rgb, ir, skip)
        alias = None
        if skip:
            pooled, *alias = pooled
        return pooled, alias, tuple(rgb.shape[1:3])

    def forward(self, x, skip=False):
        pooled, alias, hw = self._pool(x, skip)
        y = self._transform(pooled)
        a, b = F2.split_tokens(y)
        out = FusedTokens(a, b, hw)
        return (out, alias) if skip else out


class GPT1_fourier(GPT):
    """Fusion Focus Module: GPT plus the pooled-feature gate (conv1 -> sigmoid -> conv2 -> multiply) and the spectral
    "pattern" separation loss (no gradient, as in the reference where torch.tensor() detaches it)."""

    def __init__(self, d_model, h=8, block_exp=4, n_layer=8, vert_anchors=8, horz_anchors=8, embd_pdrop=0.1,
                 attn_pdrop=0.1, resid_pdrop=0.1):
        super().__init__(d_model, h, block_exp, n_layer, vert_anchors, horz_anchors, embd_pdrop, attn_pdrop, resid_pdrop)
        self.conv1 = _holder_conv(d_model, 8, 1, 1)
        self.sig = nn.Sigmoid()
        self.conv2 = _holder_conv(8, d_model, 1, 1)

    def forward(self, x, skip=False):
        pooled, alias, hw = self._pool(x, skip)                              # (B,128,C): rgb tokens then ir tokens
        bs, c = pooled.shape[0], pooled.shape[-1]
        gate = F2.sigmoid(ops.conv_bias(pooled, self.conv1.weight.view(8, c), None, 1))   # (B,128,8)
        with torch.no_grad():                                                # pattern loss: value only
            hi = F2.ffm_highpass_mul(pooled.detach().view(bs * 2, 64, c))
            ghi = F2.sigmoid(ops.conv_bias(hi, self.conv1.weight.detach().view(8, c), None, 1)).view(bs, 2, 64, 8)
            g = gate.detach().view(bs, 2, 64, 8)
            self.pattenLoss = F2.separation_loss(g[:, 0], g[:, 1], ghi[:, 0], ghi[:, 1])
        pt = ops.conv_bias(gate, self.conv2.weight.view(c, 8), None, 1)      # (B,128,C)
        y = self._transform(F2.mul(pt, pooled))
        self.last_tokens = y.detach()
        a, b = F2.split_tokens(y)
        out = FusedTokens(a, b, hw)
        return ((out, self.pattenLoss), alias) if skip else (out, self.pattenLoss)


class RecContrastiveLoss(nn.Module):
    """Parameter-free member of the reference's Model (`contrastive_loss_func`, yolo_test.py:94) that its forward never
    calls.  Present so that checkpoints pickled by the reference -- which store the class by name -- unpickle against this
    package (models/experimental.py).  Formula of the reference class (common.py:1431-1443): hinge on the positive pair
    distance only."""

    def __init__(self, margin=1.0):
        super().__init__()
        self.margin = margin

    def forward(self, anchor, positive, negative=None):
        return torch.relu(torch.nn.functional.pairwise_distance(anchor, positive, 2) + self.margin).mean()


# names this module does not define (the reference's helpers outside the hot path) come from the reference checkout's
# module of the same name when one is overlaid: mmidet_hip/overlay.py
from mmidet_hip.overlay import fall_through as _fall_through  # noqa: E402

__getattr__ = _fall_through(__name__)
