"""GPU parity: Contour Enhancement Module kernels (csrc/cem.hip) against the reference formulation
(models/common.py:751-911): EnhanceConv2d as a real 24->24 conv2d with sobel_weight*sobel_factor, AdaptiveModule3 end
to end (forward, all parameter gradients, BN running stats) at a size that spans several 16x16 tiles and ragged edges."""
import pytest
import torch
import torch.nn.functional as F

from test_ops_gpu import close, dev, nchw, nhwc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('hw', [(40, 56), (17, 19), (16, 16), (5, 3)])
def test_sobel_add_matches_conv2d_bank(hw):
    from mmidet_hip import ops
    from oracle.ref_model import EnhanceConv2d
    h, w = hw
    g = torch.Generator().manual_seed(h * w)
    ref = EnhanceConv2d(24, 24)
    ref.sobel_factor.data = torch.rand(24, 1, 1, 1, generator=g) + 0.5
    ref.bias.data = torch.randn(24, generator=g) * 0.1
    r = torch.randn(2, 24, h, w, generator=g, requires_grad=True)
    t = r + ref(r)
    gt = torch.randn(t.shape, generator=g)
    t.backward(gt)
    d = dev()
    rg = nhwc(r.detach()).to(d).requires_grad_()
    fg = ref.sobel_factor.detach().to(d).requires_grad_()
    bg = ref.bias.detach().to(d).requires_grad_()
    tg = ops.sobel_add(rg, fg, bg)
    tg.backward(nhwc(gt).to(d))
    close(nchw(tg), t, what='t', tol=1e-5)
    close(nchw(rg.grad), r.grad, what='dr', tol=1e-5)
    close(fg.grad, ref.sobel_factor.grad, what='dfactor', tol=1e-4)
    close(bg.grad, ref.bias.grad, what='dbias', tol=1e-5)


@pytest.mark.parametrize('shape', [(2, 48, 40), (1, 33, 17)])
def test_cem_module_end_to_end(shape):
    """AdaptiveModule3: product module (small-channel direct convs + stencil bank) vs the oracle module."""
    from models.common import AdaptiveModule3
    from oracle import portable_init
    from oracle.ref_model import AdaptiveModule3 as OCem
    b, h, w = shape
    o = OCem(3, 3)
    sd = portable_init.fill_({'Enhance.' + k: v for k, v in o.state_dict().items()})
    sd = {k[len('Enhance.'):]: v for k, v in sd.items()}
    o.load_state_dict(sd)
    m = AdaptiveModule3(3, 3)
    for bn in (m.bn2, m.bn3):
        bn.eps, bn.momentum = 1e-3, 0.03
    m.load_state_dict(sd, strict=True)
    m = m.to(dev()).train()
    o.train()
    g = torch.Generator().manual_seed(b * h)
    x = torch.rand(b, 3, h, w, generator=g)
    xo = x.clone().requires_grad_()
    yo = o(xo)
    gy = torch.randn(yo.shape, generator=g)
    yo.backward(gy)
    xg = nhwc(x).to(dev()).requires_grad_()
    assert m.sobel.is_standard_bank()
    yg = m(xg)
    yg.backward(nhwc(gy).to(dev()))
    close(nchw(yg), yo, what='CEM out', tol=1e-4)
    close(nchw(xg.grad), xo.grad, what='CEM dx', tol=1e-3)
    og = dict(o.named_parameters())
    for n, p in m.named_parameters():
        if p.grad is not None:
            close(p.grad, og[n].grad, what='d ' + n, tol=2e-3)
    for k in ('bn2.running_mean', 'bn2.running_var', 'bn3.running_mean', 'bn3.running_var'):
        close(m.state_dict()[k], o.state_dict()[k], what=k, tol=1e-5)


def test_nonstandard_bank_falls_back_to_the_real_conv():
    from models.common import AdaptiveModule3
    m = AdaptiveModule3(3, 3).to(dev())
    assert m.sobel.is_standard_bank()
    with torch.no_grad():
        m.sobel.sobel_weight[0, 0, 0, 0] += 1.0
    assert not m.sobel.is_standard_bank()
    x = torch.rand(1, 8, 8, 3, device=dev())
    assert m(x).shape == x.shape
