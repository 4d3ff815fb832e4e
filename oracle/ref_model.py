"""Oracle (test infrastructure, CPU/fp32 torch): restatement of the reference two-stream graph.

Follows /root/reference/models/yolo_test.py and /root/reference/models/common.py; each class/function
cites the lines it restates.  Attribute names equal the reference's so that ``state_dict()`` keys are
identical (checkpoint compatibility is part of the boundary) and a reference state dict
loads here with ``strict=True``.
"""
import math
from copy import deepcopy

import torch
import torch.nn as nn
import torch.nn.functional as F

BN_EPS, BN_MOMENTUM = 1e-3, 0.03  # utils/torch_utils.py:144-153 (initialize_weights)


def make_divisible(x, divisor):  # utils/general.py:230-232
    return math.ceil(x / divisor) * divisor


def _bn(c):
    return nn.BatchNorm2d(c, eps=BN_EPS, momentum=BN_MOMENTUM)


class Conv(nn.Module):
    """SiLU(BN(conv(x))), bias-free conv, pad = k//2.  common.py:108-125."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__()
        assert g == 1
        self.conv = nn.Conv2d(c1, c2, k, s, k // 2 if p is None else p, bias=False)
        self.bn = _bn(c2)
        self.act = nn.SiLU() if act is True else nn.Identity()

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))

    def fuseforward(self, x):  # common.py:124-125
        return self.act(self.conv(x))


def fuse_conv_and_bn(conv, bn):
    """utils/torch_utils.py:181-201: W' = diag(gamma / sqrt(eps + var)) W,  b' = beta - gamma * mean / sqrt(var + eps)."""
    fused = nn.Conv2d(conv.in_channels, conv.out_channels, kernel_size=conv.kernel_size, stride=conv.stride,
                      padding=conv.padding, groups=conv.groups, bias=True).requires_grad_(False)
    w_conv = conv.weight.clone().view(conv.out_channels, -1)                                  # :192
    w_bn = torch.diag(bn.weight.div(torch.sqrt(bn.eps + bn.running_var)))                     # :193
    fused.weight.copy_(torch.mm(w_bn, w_conv).view(fused.weight.shape))                       # :194
    b_conv = torch.zeros(conv.weight.size(0)) if conv.bias is None else conv.bias             # :197
    b_bn = bn.bias - bn.weight.mul(bn.running_mean).div(torch.sqrt(bn.running_var + bn.eps))  # :198
    fused.bias.copy_(torch.mm(w_bn, b_conv.reshape(-1, 1)).reshape(-1) + b_bn)                # :199
    return fused


class Bottleneck(nn.Module):
    """x + cv2(cv1(x)), 1x1 then 3x3.  common.py:602-613."""

    def __init__(self, c1, c2, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_, c2, 3, 1)
        self.add = shortcut and c1 == c2

    def forward(self, x):
        y = self.cv2(self.cv1(x))
        return x + y if self.add else y


class C3(nn.Module):
    """cv3(cat(m(cv1(x)), cv2(x))).  common.py:637-651."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*[Bottleneck(c_, c_, shortcut, g, e=1.0) for _ in range(n)])

    def forward(self, x):
        return self.cv3(torch.cat((self.m(self.cv1(x)), self.cv2(x)), dim=1))


class SPP(nn.Module):
    """cv2(cat(x, mp5, mp9, mp13)) on x = cv1(input).  common.py:681-693."""

    def __init__(self, c1, c2, k=(5, 9, 13)):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * (len(k) + 1), c2, 1, 1)
        self.m = nn.ModuleList([nn.MaxPool2d(kernel_size=x, stride=1, padding=x // 2) for x in k])

    def forward(self, x):
        x = self.cv1(x)
        return self.cv2(torch.cat([x] + [m(x) for m in self.m], 1))


class Focus(nn.Module):
    """space-to-depth (4 phase slices, order (0,0),(1,0),(0,1),(1,1) in (row,col)) then Conv.  common.py:696-709."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__()
        self.conv = Conv(c1 * 4, c2, k, s, p, g, act)

    def forward(self, x):
        return self.conv(torch.cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1))


class Concat(nn.Module):  # common.py:740-748
    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def forward(self, x):
        return torch.cat(x, self.d)


class Add(nn.Module):  # common.py:914-921
    def __init__(self, arg):
        super().__init__()
        self.arg = arg

    def forward(self, x):
        return x[0] + x[1]


class Add2(nn.Module):  # common.py:924-935
    def __init__(self, c1, index):
        super().__init__()
        self.index = index

    def forward(self, x):
        return x[0] + x[1][self.index]


# stencil bank of EnhanceConv2d, index = out_channel % 8, row-major 3x3.  common.py:838-882
STENCILS = (
    (-1, -2, -1, 0, 0, 0, 1, 2, 1),
    (-1, 0, 1, -2, 0, 2, -1, 0, 1),
    (-2, -1, 0, -1, 0, 1, 0, 1, 2),
    (-2, -1, 0, -1, 0, 1, 0, 1, 2),  # the reference's "other diagonal" branch is identical (849-862)
    (0, 1, 0, 1, -4, 1, 0, 1, 0),
    (0, 1, 0, 1, 4, 1, 0, 1, 0),
    (-1, -1, -1, 0, 0, 0, 1, 1, 1),
    (-1, 0, 1, -1, 0, 1, -1, 0, 1),
)


class EnhanceConv2d(nn.Module):
    """Fixed 3x3 edge-stencil bank replicated over every input channel, scaled by a trainable per-output
    factor, plus bias.  common.py:806-911."""

    def __init__(self, cin, cout):
        super().__init__()
        assert cout % 8 == 0
        self.bias = nn.Parameter(torch.zeros(cout))
        w = torch.zeros(cout, cin, 3, 3)
        for o in range(cout):
            w[o] = torch.tensor(STENCILS[o % 8], dtype=torch.float32).view(1, 3, 3)
        self.sobel_weight = nn.Parameter(w, requires_grad=False)
        self.sobel_factor = nn.Parameter(torch.ones(cout, 1, 1, 1))

    def forward(self, x):
        return F.conv2d(x, self.sobel_weight * self.sobel_factor, self.bias, 1, 1)


class AdaptiveModule3(nn.Module):
    """Contour Enhancement Module.  common.py:751-803."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        c = int(in_channels)
        self.conv2 = nn.Conv2d(c, c * 8, 3, 1, 1, bias=False)
        self.bn2 = _bn(c * 8)
        self.relu2 = nn.LeakyReLU(0.1)
        self.sobel = EnhanceConv2d(c * 8, c * 8)
        self.conv3 = nn.Conv2d(c * 8, c, 3, 1, 1, bias=False)
        self.bn3 = _bn(c)
        self.relu3 = nn.LeakyReLU(0.1)

    def forward(self, x):
        r = self.relu2(self.bn2(self.conv2(x)))
        s = self.sobel(r)
        return self.relu3(self.bn3(self.conv3(r + s))) + x


class SelfAttention(nn.Module):
    """h-head softmax(QK^T/sqrt(dk))V with 4 biased projections.  common.py:1147-1234."""

    def __init__(self, d_model, d_k, d_v, h, attn_pdrop=.1, resid_pdrop=.1):
        super().__init__()
        assert d_k % h == 0
        self.d_model, self.h = d_model, h
        self.d_k = self.d_v = d_model // h
        self.que_proj = nn.Linear(d_model, d_model)
        self.key_proj = nn.Linear(d_model, d_model)
        self.val_proj = nn.Linear(d_model, d_model)
        self.out_proj = nn.Linear(d_model, d_model)
        self.attn_drop = nn.Dropout(attn_pdrop)
        self.resid_drop = nn.Dropout(resid_pdrop)

    def forward(self, x):
        b, n, _ = x.shape
        q = self.que_proj(x).view(b, n, self.h, self.d_k).permute(0, 2, 1, 3)
        k = self.key_proj(x).view(b, n, self.h, self.d_k).permute(0, 2, 3, 1)
        v = self.val_proj(x).view(b, n, self.h, self.d_v).permute(0, 2, 1, 3)
        att = torch.matmul(q, k) / math.sqrt(self.d_k)
        att = self.attn_drop(torch.softmax(att, -1))
        out = torch.matmul(att, v).permute(0, 2, 1, 3).contiguous().view(b, n, self.h * self.d_v)
        return self.resid_drop(self.out_proj(out))


class myTransformerBlock(nn.Module):
    """pre-LN block: x += SA(LN(x)); x += MLP(LN(x)), MLP = Linear-GELU(erf)-Linear-Dropout.  common.py:1237-1267."""

    def __init__(self, d_model, d_k, d_v, h, block_exp, attn_pdrop, resid_pdrop):
        super().__init__()
        self.ln_input = nn.LayerNorm(d_model)
        self.ln_output = nn.LayerNorm(d_model)
        self.sa = SelfAttention(d_model, d_k, d_v, h, attn_pdrop, resid_pdrop)
        self.mlp = nn.Sequential(nn.Linear(d_model, block_exp * d_model), nn.GELU(),
                                 nn.Linear(block_exp * d_model, d_model), nn.Dropout(resid_pdrop))

    def forward(self, x):
        x = x + self.sa(self.ln_input(x))
        return x + self.mlp(self.ln_output(x))


def _init_gpt(module):  # common.py:1306-1314 / 347-355 (runs after SelfAttention.init_weights, so it wins)
    if isinstance(module, nn.Linear):
        module.weight.data.normal_(mean=0.0, std=0.02)
        if module.bias is not None:
            module.bias.data.zero_()
    elif isinstance(module, nn.LayerNorm):
        module.bias.data.zero_()
        module.weight.data.fill_(1.0)


def _tokens_to_maps(x, bs, va, ha, c, h, w):
    """(B,2*va*ha,C) -> two (B,C,h,w) maps by bilinear upsampling.  common.py:1351-1368 / 535-550."""
    x = x.view(bs, 2, va, ha, c).permute(0, 1, 4, 2, 3)
    a = x[:, 0].contiguous().view(bs, c, va, ha)
    b = x[:, 1].contiguous().view(bs, c, va, ha)
    return (F.interpolate(a, size=(h, w), mode='bilinear', align_corners=False),
            F.interpolate(b, size=(h, w), mode='bilinear', align_corners=False))


class GPT(nn.Module):
    """Cross-modal fusion transformer over 2x8x8 pooled tokens.  common.py:1270-1368."""

    def __init__(self, d_model, h=8, block_exp=4, n_layer=8, vert_anchors=8, horz_anchors=8,
                 embd_pdrop=0.1, attn_pdrop=0.1, resid_pdrop=0.1):
        super().__init__()
        self.n_embd, self.vert_anchors, self.horz_anchors = d_model, vert_anchors, horz_anchors
        self.pos_emb = nn.Parameter(torch.zeros(1, 2 * vert_anchors * horz_anchors, d_model))
        self.trans_blocks = nn.Sequential(*[myTransformerBlock(d_model, d_model, d_model, h, block_exp, attn_pdrop,
                                                               resid_pdrop) for _ in range(n_layer)])
        self.ln_f = nn.LayerNorm(d_model)
        self.drop = nn.Dropout(embd_pdrop)
        self.avgpool = nn.AdaptiveAvgPool2d((vert_anchors, horz_anchors))
        self.apply(_init_gpt)

    def forward(self, x):
        rgb, ir = x[0], x[1]
        assert rgb.shape[0] == ir.shape[0]
        bs, c, h, w = rgb.shape
        tok = torch.cat([self.avgpool(rgb).view(bs, c, -1), self.avgpool(ir).view(bs, c, -1)], dim=2)
        tok = tok.permute(0, 2, 1).contiguous()
        y = self.ln_f(self.trans_blocks(self.drop(self.pos_emb + tok)))
        return _tokens_to_maps(y, bs, self.vert_anchors, self.horz_anchors, self.n_embd, h, w)


def extract_frequency2(image):
    """fft2 -> fftshift -> box masks -> ifftshift -> ifft2 -> .half() (keeps the real part).  common.py:37-69."""
    f_shift = torch.fft.fftshift(torch.fft.fftn(image, dim=(-2, -1)), dim=(-2, -1))
    _, _, rows, cols = image.shape
    crow, ccol = rows // 2, cols // 2
    thr = crow + ccol // 4
    hi = f_shift.clone()
    hi[:, :, crow - thr:crow + thr, ccol - thr:ccol + thr] = 0
    lo = f_shift.clone()
    lo[:, :, :crow - thr, :] = 0
    lo[:, :, crow + thr:, :] = 0
    lo[:, :, :, :ccol - thr] = 0
    lo[:, :, :, ccol + thr:] = 0
    img_hi = torch.fft.ifftn(torch.fft.ifftshift(hi, dim=(-2, -1)), dim=(-2, -1))
    img_lo = torch.fft.ifftn(torch.fft.ifftshift(lo, dim=(-2, -1)), dim=(-2, -1))
    return img_lo.real.half(), img_hi.real.half()  # complex .half() discards the imaginary part


def separation_loss(M):
    """sum_{i<j} <M_i, M_j> / (l (l-1)) by the reference's literal double loop.  common.py:128-139."""
    l = M.size(0)
    tot = torch.tensor(0.0, device=M.device)
    for i in range(l - 1):
        for j in range(i + 1, l):
            tot = tot + M[i] @ M[j]
    return tot / (l * (l - 1))


class GPT1_fourier(nn.Module):
    """Fusion Focus Module.  common.py:299-552."""

    def __init__(self, d_model, h=8, block_exp=4, n_layer=8, vert_anchors=8, horz_anchors=8,
                 embd_pdrop=0.1, attn_pdrop=0.1, resid_pdrop=0.1):
        super().__init__()
        self.n_embd, self.vert_anchors, self.horz_anchors = d_model, vert_anchors, horz_anchors
        self.pos_emb = nn.Parameter(torch.zeros(1, 2 * vert_anchors * horz_anchors, d_model))
        self.trans_blocks = nn.Sequential(*[myTransformerBlock(d_model, d_model, d_model, h, block_exp, attn_pdrop,
                                                               resid_pdrop) for _ in range(n_layer)])
        self.ln_f = nn.LayerNorm(d_model)
        self.drop = nn.Dropout(embd_pdrop)
        self.avgpool = nn.AdaptiveAvgPool2d((vert_anchors, horz_anchors))
        self.conv1 = nn.Conv2d(d_model, 8, 1, bias=False)
        self.sig = nn.Sigmoid()
        self.conv2 = nn.Conv2d(8, d_model, 1, bias=False)
        self.apply(_init_gpt)

    def forward(self, x):
        rgb_in, ir_in = x[0], x[1]
        assert rgb_in.shape[0] == ir_in.shape[0]
        bs, c, h, w = rgb_in.shape
        rgb, ir = self.avgpool(rgb_in), self.avgpool(ir_in)                      # :395-396
        hp, wp = rgb.shape[2:]
        _, rgb_hi = extract_frequency2(rgb)                                      # :408-409
        _, ir_hi = extract_frequency2(ir)
        rgb_hi_m = self.sig(self.conv1(rgb_hi * rgb)).view(-1, hp * wp)          # :440-448
        ir_hi_m = self.sig(self.conv1(ir_hi * ir)).view(-1, hp * wp)             # :441-455
        rgb_m = self.sig(self.conv1(rgb))                                        # :476-477
        ir_m = self.sig(self.conv1(ir))                                          # :479-480
        keep = len(rgb_hi_m) // 8                                                # :487
        cat = torch.cat((rgb_m.view(-1, hp * wp), ir_m.view(-1, hp * wp), rgb_hi_m[:keep], ir_hi_m[:keep]), 0)
        self.pattenLoss = separation_loss(cat)                                   # :494
        p_rgb = self.conv2(rgb_m) * rgb                                          # :499-503
        p_ir = self.conv2(ir_m) * ir
        tok = torch.cat([p_rgb.view(bs, c, -1), p_ir.view(bs, c, -1)], dim=2).permute(0, 2, 1).contiguous()
        y = self.ln_f(self.trans_blocks(self.drop(self.pos_emb + tok)))          # :529-535
        a, b = _tokens_to_maps(y, bs, self.vert_anchors, self.horz_anchors, self.n_embd, h, w)
        return a, b, self.pattenLoss


class Detect(nn.Module):
    """Per-level 1x1 biased conv -> (B,na,ny,nx,no); eval adds sigmoid + grid/anchor decode.  yolo_test.py:29-73."""
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
        self.m = nn.ModuleList(nn.Conv2d(x, self.no * self.na, 1) for x in ch)

    def forward(self, x):
        x = list(x)
        z = []
        for i in range(self.nl):
            x[i] = self.m[i](x[i])
            bs, _, ny, nx = x[i].shape
            x[i] = x[i].view(bs, self.na, self.no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
            if not self.training:
                if self.grid[i].shape[2:4] != x[i].shape[2:4]:
                    yv, xv = torch.meshgrid([torch.arange(ny), torch.arange(nx)], indexing='ij')
                    self.grid[i] = torch.stack((xv, yv), 2).view(1, 1, ny, nx, 2).float().to(x[i].device)
                y = x[i].sigmoid()
                xy = (y[..., 0:2] * 2. - 0.5 + self.grid[i]) * self.stride[i]
                wh = (y[..., 2:4] * 2) ** 2 * self.anchor_grid[i]
                y = torch.cat((xy, wh, y[..., 4:]), -1)
                z.append(y.view(bs, -1, self.no))
        return x if self.training else (torch.cat(z, 1), x)


_MODULES = dict(Conv=Conv, Bottleneck=Bottleneck, C3=C3, SPP=SPP, Focus=Focus, Concat=Concat, Add=Add, Add2=Add2,
                GPT=GPT, GPT1_fourier=GPT1_fourier, Detect=Detect)


def parse_model(d, ch):
    """YAML rows [from, number, module, args] -> nn.Sequential + save list.  yolo_test.py:548-639."""
    anchors, nc, gd, gw = d['anchors'], d['nc'], d['depth_multiple'], d['width_multiple']
    na = (len(anchors[0]) // 2) if isinstance(anchors, list) else anchors
    no = na * (nc + 5)
    layers, save, c2 = [], [], ch[-1]
    for i, (f, n, m, args) in enumerate(d['backbone'] + d['head']):
        args = list(args)
        for j, a in enumerate(args):
            if a == 'nc':
                args[j] = nc
            elif a == 'anchors':
                args[j] = anchors
            elif a == 'None':
                args[j] = None
            elif a in ('False', 'True'):
                args[j] = (a == 'True')
        n = max(round(n * gd), 1) if n > 1 else n
        if m == 'nn.Upsample':
            mod = nn.Upsample(*args)
            c2 = ch[f]
            tname = 'torch.nn.modules.upsampling.Upsample'
        else:
            cls = _MODULES[m]
            tname = 'models.common.' + m if m != 'Detect' else 'Detect'
            if cls in (Conv, Bottleneck, SPP, Focus, C3):
                c1 = 3 if cls is Focus else ch[f]                       # :571-576 Focus always sees 3 channels
                c2 = args[0]
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
            elif cls is GPT1_fourier:                                    # :607-609 raw arg, NOT width-scaled (B3)
                c2 = args[0]
                args = [c2]
            elif cls is Detect:
                args.append([ch[x] for x in f])
                if isinstance(args[1], int):
                    args[1] = [list(range(args[1] * 2))] * len(f)
            mod = nn.Sequential(*[cls(*args) for _ in range(n)]) if n > 1 else cls(*args)
        mod.i, mod.f, mod.type = i, f, tname
        mod.np = sum(x.numel() for x in mod.parameters())
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        layers.append(mod)
        if i == 0:
            ch = []
        ch.append(c2)
    return nn.Sequential(*layers), sorted(save)


class Model(nn.Module):
    """Two-stream YOLOv5 with CEM/FFM/CBM/IGM.  yolo_test.py:77-273, 338-486.

    ``dropout`` overrides every nn.Dropout.p (parity runs use 0.0; the reference default is 0.1)."""

    def __init__(self, cfg, ch=3, nc=None, anchors=None, dropout=None):
        super().__init__()
        if isinstance(cfg, dict):
            self.yaml = deepcopy(cfg)
        else:
            import yaml
            from pathlib import Path
            self.yaml_file = Path(cfg).name
            with open(cfg) as f:
                self.yaml = yaml.safe_load(f)
        self.Enhance = AdaptiveModule3(int(ch), int(ch))                       # :98-99
        ch = self.yaml['ch'] = self.yaml.get('ch', ch)
        if nc and nc != self.yaml['nc']:
            self.yaml['nc'] = nc
        self.model, self.save = parse_model(deepcopy(self.yaml), ch=[ch])
        self.names = [str(i) for i in range(self.yaml['nc'])]
        m = self.model[-1]
        if isinstance(m, Detect):
            m.stride = torch.tensor([8.0, 16.0, 32.0])                          # :127 hard-coded
            m.anchors /= m.stride.view(-1, 1, 1)
            a = m.anchor_grid.prod(-1).view(-1)                                 # autoanchor.py:12-20
            if (a[-1] - a[0]).sign() != (m.stride[-1] - m.stride[0]).sign():
                m.anchors[:] = m.anchors.flip(0)
                m.anchor_grid[:] = m.anchor_grid.flip(0)
            self.stride = m.stride
            self._initialize_biases()
        if dropout is not None:
            for mod in self.modules():
                if isinstance(mod, nn.Dropout):
                    mod.p = dropout

    def _initialize_biases(self):  # yolo_test.py:280-290
        m = self.model[-1]
        for mi, s in zip(m.m, m.stride):
            b = mi.bias.view(m.na, -1)
            b.data[:, 4] += math.log(8 / (640 / s) ** 2)
            b.data[:, 5:] += math.log(0.6 / (m.nc - 0.99))
            mi.bias = nn.Parameter(b.view(-1), requires_grad=True)

    def fuse(self):  # yolo_test.py:304-312 (only `type(m) is Conv`: the CEM's conv+BN pairs are left alone)
        with torch.no_grad():
            for m in self.model.modules():
                if type(m) is Conv and hasattr(m, 'bn'):
                    m.conv = fuse_conv_and_bn(m.conv, m.bn)
                    delattr(m, 'bn')
                    m.forward = m.fuseforward
        return self

    def forward(self, x, x2, augment=False, profile=False):
        return self.forward_once(x, x2)

    def forward_once(self, x, x2):  # yolo_test.py:162-273
        dev = x.device
        self.ContrastiveValue = torch.zeros(0, device=dev)
        self.SSIMloss = torch.zeros(0, device=dev)
        self.PTLoss = torch.zeros(0, device=dev)
        self.Entropy_loss = torch.zeros(0, device=dev)
        x = self.Enhance(x)                                                     # :187 CEM on RGB only
        y = []
        for m in self.model:
            if m.f != -1 and m.f != -4:                                         # :192-196
                x = y[m.f] if isinstance(m.f, int) else [x if j == -1 else y[j] for j in m.f]
            if m.f == -4:                                                       # :222-223 feed the IR image
                x = m(x2)
            elif isinstance(m, GPT1_fourier):
                in_rgb, in_ir = x[0], x[1]
                self.ContrastiveValue = contrastive_value(in_rgb, in_ir).detach()    # :216-220
                o_rgb, o_ir, pt = m(x)                                          # :228-230
                x = [o_rgb, o_ir]
                self.PTLoss = pt.detach()
                avg = torch.mean(torch.stack([o_rgb, o_ir]), dim=0)             # :248-249
                self.SSIMloss = fusing_loss2(in_rgb, in_ir, avg, avg)           # :251-252
                self.Entropy_loss = entropy_loss(in_rgb, in_ir, avg)            # :254-255
            else:
                x = m(x)
            y.append(x if m.i in self.save else None)
        # :263-268: the weighted sum is overwritten by SSIMloss, then detached by torch.tensor(...)
        self.Combine_loss = self.SSIMloss.detach().clone()
        return x, self.Combine_loss


def _contrastive(e1, e2, label_mean, margin=1.0):  # yolo_test.py:338-354
    d = F.normalize(e1 - e2, dim=1)
    md = torch.mean(torch.square(d))
    return (1 - label_mean) * torch.exp(md) + label_mean * (torch.exp(md) - margin)


def contrastive_value(rgb, ir):  # yolo_test.py:356-404 (Contrast Bridge Module)
    pos = _contrastive(rgb[0:-1], ir[0:-1], 0.0)
    neg = _contrastive(rgb[0:-1], ir[1:], 1.0)
    neg2 = _contrastive(rgb[1:], ir[0:-1], 1.0)
    return (pos * 2 + neg + neg2) / 4.0


def _entropy(img):  # yolo_test.py:424-429
    hist = torch.histc(img.float(), bins=256, min=0, max=1)
    hist = hist / hist.sum()
    nz = hist[hist > 0]
    return -torch.sum(nz * torch.log2(nz))


def entropy_loss(rgb, ir, fused):  # yolo_test.py:406-422
    return (_entropy(rgb) + _entropy(ir)) - _entropy(fused)


def ssim_loss(a, b):  # yolo_test.py:461-486 (global, not windowed)
    mu1, mu2 = torch.mean(a), torch.mean(b)
    var1, var2 = torch.mean((a - mu1) ** 2), torch.mean((b - mu2) ** 2)
    cov = torch.mean((a - mu1) * (b - mu2))
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    ssim = (2 * mu1 * mu2 + c1) * (2 * cov + c2) / ((mu1 ** 2 + mu2 ** 2 + c1) * (var1 + var2 + c2))
    return 1 - ssim


def fusing_loss2(rgb, ir, f_rgb, f_ir):  # yolo_test.py:444-459
    w = 0.5 * ssim_loss(rgb, f_rgb) + 0.5 * ssim_loss(ir, f_ir)
    return w + torch.mean(torch.abs(torch.std(f_rgb) - torch.std(f_ir)))
