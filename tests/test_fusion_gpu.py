"""GPU parity: fusion-stack kernels (token side + spatial side + CBM/IGM statistics) against plain torch / the oracle."""
import math

import pytest
import torch
import torch.nn.functional as F

from test_ops_gpu import close, dev, nchw, nhwc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('rows,C', [(256, 128), (2048, 1024), (130, 32), (64, 2048)])
def test_layernorm(rows, C):
    from mmidet_hip import fusion_ops as F2
    g = torch.Generator().manual_seed(rows + C)
    x, w, b = torch.randn(rows, C, generator=g) * 2 + 0.3, torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    xr, wr, br = (t.clone().requires_grad_() for t in (x, w, b))
    yr = F.layer_norm(xr, (C,), wr, br, 1e-5)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    d = dev()
    xg, wg, bg = (t.to(d).requires_grad_() for t in (x, w, b))
    yg = F2.layernorm(xg, wg, bg, 1e-5)
    yg.backward(gy.to(d))
    close(yg, yr, what='y')
    close(xg.grad, xr.grad, what='dx')
    close(wg.grad, wr.grad, what='dgamma')
    close(bg.grad, br.grad, what='dbeta')


def test_pointwise():
    from mmidet_hip import fusion_ops as F2
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1000, 37, generator=g) * 3
    y2 = torch.randn(1000, 37, generator=g)
    d = dev()
    for name, fn, ref in (('gelu', F2.gelu, F.gelu), ('sigmoid', F2.sigmoid, torch.sigmoid)):
        xr = x.clone().requires_grad_()
        r = ref(xr)
        r.backward(y2)
        xg = x.to(d).requires_grad_()
        o = fn(xg)
        o.backward(y2.to(d))
        close(o, r, what=name, tol=1e-5)
        close(xg.grad, xr.grad, what='d' + name, tol=1e-5)
    ar, br = x.clone().requires_grad_(), y2.clone().requires_grad_()
    (ar * br).backward(x)
    ag, bg = x.to(d).requires_grad_(), y2.to(d).requires_grad_()
    o = F2.mul(ag, bg)
    o.backward(x.to(d))
    assert torch.equal(o.cpu(), (x * y2))
    assert torch.equal(ag.grad.cpu(), ar.grad) and torch.equal(bg.grad.cpu(), br.grad)


def test_dropout_add():
    from mmidet_hip import fusion_ops as F2
    d = dev()
    g = torch.Generator().manual_seed(2)
    a = torch.randn(4, 128, 64, generator=g)
    pos = torch.randn(1, 128, 64, generator=g)
    ag, pg = a.to(d).requires_grad_(), pos.to(d).requires_grad_()
    o = F2.dropout_add(ag, pg, 0.0, True)                       # p = 0: exact broadcast add
    assert torch.equal(o.cpu(), a + pos)
    o.backward(torch.ones_like(o))
    assert torch.equal(ag.grad.cpu(), torch.ones_like(a))
    close(pg.grad, torch.full_like(pos, 4.0), what='dpos', tol=1e-6)
    big = torch.ones(64, 128, 256, device=d, requires_grad=True)
    o = F2.dropout_add(big, None, 0.1, True)
    keep = (o != 0).float().mean().item()
    assert abs(keep - 0.9) < 0.005, keep
    assert torch.allclose(o[o != 0], torch.tensor(1 / 0.9, device=d))
    o.backward(torch.ones_like(o))
    assert torch.equal(big.grad, o.detach())                    # the backward regenerates the same mask
    assert F2.dropout_add(big, None, 0.1, False) is big         # eval: identity


@pytest.mark.parametrize('B,C,heads', [(2, 128, 8), (2, 256, 8), (3, 512, 8), (2, 1024, 8), (2, 32, 8),
                                         (2, 160, 8), (2, 320, 8), (1, 640, 8), (1, 1280, 8)])   # yolov5x widths: dk 20..160
def test_attention(B, C, heads):
    """softmax(QK^T/sqrt(dk))V per head (models/common.py:1206-1231), forward and backward, no dropout."""
    from mmidet_hip import fusion_ops as F2
    g = torch.Generator().manual_seed(B * C)
    q, k, v = (torch.randn(B, 128, C, generator=g) for _ in range(3))
    dk = C // heads

    def ref(q, k, v):
        qh = q.view(B, 128, heads, dk).permute(0, 2, 1, 3)
        kh = k.view(B, 128, heads, dk).permute(0, 2, 3, 1)
        vh = v.view(B, 128, heads, dk).permute(0, 2, 1, 3)
        att = torch.softmax(torch.matmul(qh, kh) / math.sqrt(dk), -1)
        return torch.matmul(att, vh).permute(0, 2, 1, 3).contiguous().view(B, 128, C)
    qr, kr, vr = (t.clone().requires_grad_() for t in (q, k, v))
    o = ref(qr, kr, vr)
    go = torch.randn(o.shape, generator=g)
    o.backward(go)
    d = dev()
    qg, kg, vg = (t.to(d).requires_grad_() for t in (q, k, v))
    og = F2.attention(qg, kg, vg, heads, 0.0, True)
    og.backward(go.to(d))
    close(og, o, what='out')
    close(qg.grad, qr.grad, what='dq')
    close(kg.grad, kr.grad, what='dk')
    close(vg.grad, vr.grad, what='dv')


def test_attention_dropout_is_consistent():
    """With attention dropout the backward must use the forward's mask: check against finite differences of the kernel."""
    from mmidet_hip import fusion_ops as F2
    from mmidet_hip import lib
    d = dev()
    torch.manual_seed(0)
    B, C, heads = 1, 32, 8
    q, k, v = (torch.randn(B, 128, C, device=d) for _ in range(3))
    seed = 1234567
    st = torch.cuda.current_stream().cuda_stream

    def run(qq):
        out, probs = torch.empty_like(qq), torch.empty(B, heads, 128, 128, device=d)
        lib.attention_fwd(qq.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), probs.data_ptr(), B, heads, C // heads,
                          C, 0.3, seed, None, st)
        return out, probs
    out, probs = run(q)
    w = torch.randn_like(out)
    dq, dk_, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    lib.attention_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), probs.data_ptr(), w.data_ptr(), dq.data_ptr(), dk_.data_ptr(),
                      dv.data_ptr(), B, heads, C // heads, C, 0.3, seed, None, st)
    dirn = torch.randn_like(q)
    eps = 1e-2
    fd = ((run(q + eps * dirn)[0] - run(q - eps * dirn)[0]) * w).sum().item() / (2 * eps)
    an = (dq * dirn).sum().item()
    assert abs(fd - an) < 2e-2 * max(1.0, abs(an)), (fd, an)


@pytest.mark.parametrize('hw,C', [((160, 160), 128), ((20, 20), 64), ((40, 24), 32), ((2, 2), 32), ((12, 20), 256)])
def test_pool_and_upsample_add(hw, C):
    """AdaptiveAvgPool2d(8) -> tokens and bilinear(8x8 -> HxW) + Add2, forward and backward."""
    from mmidet_hip import fusion_ops as F2
    H, W = hw
    B = 2
    g = torch.Generator().manual_seed(H * W + C)
    rgb, ir = torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)
    rr, ii = rgb.clone().requires_grad_(), ir.clone().requires_grad_()
    tok_r = torch.cat([F.adaptive_avg_pool2d(rr, 8).view(B, C, -1), F.adaptive_avg_pool2d(ii, 8).view(B, C, -1)], 2).permute(0, 2, 1)
    gt = torch.randn(B, 128, C, generator=g)
    tok_r.backward(gt)
    d = dev()
    rg, ig = nhwc(rgb).to(d).requires_grad_(), nhwc(ir).to(d).requires_grad_()
    tok = F2.pool_tokens(rg, ig)
    tok.backward(gt.to(d))
    close(tok, tok_r, what='tokens', tol=1e-5)
    close(nchw(rg.grad), rr.grad, what='d rgb', tol=1e-5)
    close(nchw(ig.grad), ii.grad, what='d ir', tol=1e-5)
    # upsample + add
    t = torch.randn(B, 128, C, generator=g)
    tr, xr = t.clone().requires_grad_(), rgb.clone().requires_grad_()
    maps = tr.view(B, 2, 8, 8, C).permute(0, 1, 4, 2, 3)
    outs_r = [xr + F.interpolate(maps[:, i].contiguous(), size=(H, W), mode='bilinear', align_corners=False) for i in range(2)]
    go = torch.randn(B, C, H, W, generator=g)
    (outs_r[0] * go + outs_r[1] * (go * 0.5)).sum().backward()
    tg, xg = t.to(d).requires_grad_(), nhwc(rgb).to(d).requires_grad_()
    a, b = F2.split_tokens(tg)
    o0, o1 = F2.upsample_add(xg, a), F2.upsample_add(xg, b)
    (o0 * nhwc(go).to(d) + o1 * (nhwc(go).to(d) * 0.5)).sum().backward()
    close(nchw(o0), outs_r[0], what='up0', tol=1e-5)
    close(nchw(o1), outs_r[1], what='up1', tol=1e-5)
    close(tg.grad, tr.grad, what='dtok', tol=1e-4)
    close(nchw(xg.grad), xr.grad, what='dx', tol=1e-5)


def test_ffm_spectral_and_separation_loss():
    """extract_frequency2 high-pass * pooled (common.py:37-69, 440-441) and Seperation_loss (128-139) vs the oracle."""
    from mmidet_hip import fusion_ops as F2
    from oracle.ref_model import extract_frequency2, separation_loss
    g = torch.Generator().manual_seed(4)
    B, C = 3, 40
    pooled = torch.randn(B, C, 8, 8, generator=g) * 2
    _, hi = extract_frequency2(pooled)
    ref = hi * pooled                                              # fp16 * fp32 -> fp32
    d = dev()
    tok = pooled.view(B, C, 64).permute(0, 2, 1).contiguous().to(d)   # (B,64,C)
    out = F2.ffm_highpass_mul(tok)
    got = out.cpu().permute(0, 2, 1).reshape(B, C, 8, 8)
    # fp16 rounding of the high-pass can flip by one fp16 ulp (2^-11 relative) where the two fp32 DFTs differ in the last bit
    err = (got - ref).abs()
    assert float((err > 1e-6 + 1.1e-3 * ref.abs()).float().mean()) == 0.0
    assert float((err > 1e-6 + 1e-5 * ref.abs()).float().mean()) < 0.02
    gates = [torch.rand(B, 8, 8, 8, generator=g) for _ in range(4)]  # NCHW gate maps (B,8ch,8,8)
    rows = torch.cat([gates[0].view(-1, 64), gates[1].view(-1, 64), gates[2].view(-1, 64)[:B], gates[3].view(-1, 64)[:B]], 0)
    ref_l = separation_loss(rows)
    gg = [t.view(B, 8, 64).permute(0, 2, 1).contiguous().to(d) for t in gates]   # (B,64,8)
    got_l = F2.separation_loss(*gg)
    assert abs(float(got_l) - float(ref_l)) < 1e-5 * abs(float(ref_l))


@pytest.mark.parametrize('B,H,W,C', [(3, 32, 32, 32), (2, 40, 24, 64), (1, 16, 16, 32)])
def test_fusion_stats(B, H, W, C):
    """SSIMloss / Entropy_loss / ContrastiveValue (models/yolo_test.py:338-486) in one pass vs the oracle functions."""
    from mmidet_hip import fusion_ops as F2
    from oracle.ref_model import contrastive_value, entropy_loss, fusing_loss2
    g = torch.Generator().manual_seed(B + H)
    a = torch.randn(B, C, H, W, generator=g) * 0.5 + 0.3
    b = torch.randn(B, C, H, W, generator=g) * 0.4 + 0.2
    tok = torch.randn(B, 128, C, generator=g) * 0.5 + 0.2
    maps = tok.view(B, 2, 8, 8, C).permute(0, 1, 4, 2, 3)
    o = [F.interpolate(maps[:, i].contiguous(), size=(H, W), mode='bilinear', align_corners=False) for i in range(2)]
    avg = torch.mean(torch.stack(o), dim=0)
    ref = (float(fusing_loss2(a, b, avg, avg)), float(entropy_loss(a, b, avg)), float(contrastive_value(a, b)))
    d = dev()
    st = F2.fusion_stats(nhwc(a).to(d), nhwc(b).to(d), tok.to(d)).cpu().tolist()
    assert abs(st[0] - ref[0]) < 1e-4 * max(1, abs(ref[0])), (st, ref)
    assert abs(st[1] - ref[1]) < 2e-3 * max(1, abs(ref[1])), (st, ref)     # bin-edge rounding of histc
    if B > 1:
        assert abs(st[2] - ref[2]) < 1e-5 * abs(ref[2]), (st, ref)
    else:
        assert math.isnan(st[2]) and math.isnan(ref[2])                       # reference quirk at B=1


def _block_params(blk):
    sa = blk.sa
    return (blk.ln_input.weight, blk.ln_input.bias, sa.que_proj.weight, sa.que_proj.bias, sa.key_proj.weight,
            sa.key_proj.bias, sa.val_proj.weight, sa.val_proj.bias, sa.out_proj.weight, sa.out_proj.bias,
            blk.ln_output.weight, blk.ln_output.bias, blk.mlp[0].weight, blk.mlp[0].bias, blk.mlp[2].weight, blk.mlp[2].bias)


@pytest.mark.parametrize('packed', [False, True], ids=['three_projections', 'packed_qkv'])
@pytest.mark.parametrize('B,C', [(2, 128), (16, 256), (3, 512), (16, 1024), (1, 160)])
def test_transformer_block_matches_oracle(B, C, packed):
    """myTransformerBlock (models/common.py:1237-1267) as one fused autograd node against the oracle's module: output,
    input gradient and the 16 parameter gradients.  (16, 256) and (16, 1024) are bench shapes (stream-K schedules).
    packed: after fusion_ops.pack_qkv the q/k/v projections run as one GEMM forward and one input-gradient GEMM."""
    import models.common as mc
    from mmidet_hip import fusion_ops as F2
    from oracle import ref_model as R
    torch.manual_seed(C + B)
    ref = R.myTransformerBlock(C, C, C, 8, 4, 0.0, 0.0)
    for p in ref.parameters():                       # default Linear init leaves tiny biases; make every term count
        p.data.add_(0.05 * torch.randn_like(p))
    x = torch.randn(B, 128, C)
    xr = x.clone().requires_grad_()
    yr = ref(xr)
    go = torch.randn_like(yr)
    yr.backward(go)
    d = dev()
    blk = mc.myTransformerBlock(C, C, C, 8, 4, 0.0, 0.0)
    blk.load_state_dict(ref.state_dict())
    blk.to(d).train()
    if packed:
        assert F2.pack_qkv(blk) == 1 and F2._back_to_back(blk.sa.que_proj.weight, blk.sa.key_proj.weight, blk.sa.val_proj.weight)
        assert all(torch.equal(v.cpu(), ref.state_dict()[k]) for k, v in blk.state_dict().items())
    xg = x.to(d).requires_grad_()
    yg = blk(xg)
    yg.backward(go.to(d))
    torch.cuda.synchronize()
    close(yg, yr, what='out')
    close(xg.grad, xr.grad, tol=2e-3, what="dx")
    scale = float(ref.sa.que_proj.bias.grad.abs().max())
    for (n, pr), pg in zip(ref.named_parameters(), blk.parameters()):
        if n == 'sa.key_proj.bias':     # softmax is invariant to a key bias: the true gradient is 0, both sides hold rounding noise
            assert float(pg.grad.abs().max()) < 1e-3 * scale and float(pr.grad.abs().max()) < 1e-3 * scale
            continue
        close(pg.grad, pr.grad, tol=2e-3, what=n)


def test_transformer_block_unpacked_parameters_take_the_packed_kernels_bit_for_bit():
    """Parameters that do not lie back to back (never packed, or moved after packing) are gathered into a (3d, d) matrix per call
    and run the SAME kernels as the packed layout: outputs and all gradients equal bit for bit, and run to run (round 3: the
    separate three-projection path this replaces was not run-to-run bit-identical)."""
    import models.common as mc
    from mmidet_hip import fusion_ops as F2
    d = dev()
    torch.manual_seed(11)
    B, C = 16, 256
    blk = mc.myTransformerBlock(C, C, C, 8, 4, 0.1, 0.1).to(d).train()
    for p in blk.parameters():
        p.data.add_(0.05 * torch.randn_like(p))
    x = torch.randn(B, 128, C, device=d)
    go = torch.randn(B, 128, C, device=d)
    ps, seeds, eps = (0.1, 0.1, 0.1), (5, 6, 7), (1e-5, 1e-5)

    def run():
        prm = _block_params(blk)
        for p in prm:
            p.grad = None
        xg = x.clone().requires_grad_()
        y = F2._TransformerBlock.apply(xg, 8, ps, eps, seeds, *prm)
        y.backward(go)
        torch.cuda.synchronize()
        return [y.detach().clone(), xg.grad.clone()] + [p.grad.clone() for p in prm]

    assert not F2._back_to_back(blk.sa.que_proj.weight, blk.sa.key_proj.weight, blk.sa.val_proj.weight)
    a, b = run(), run()
    for p in blk.parameters():
        p.grad = None            # (pack_qkv leaves parameters alone whose .grad is set)
    assert F2.pack_qkv(blk) == 1
    c = run()
    for i, (u, v, w) in enumerate(zip(a, b, c)):
        assert torch.equal(u, v), 'tensor %d differs between two runs of the unpacked layout' % i
        assert torch.equal(u, w), 'tensor %d differs between the unpacked and the packed layout' % i


def test_transformer_block_dropout_matches_unfused_kernels():
    """With dropout the fused node must draw the masks the separate kernels draw from the same salts (same hash, same
    element index) and use them again in the backward: compare with the block composed of the single-op kernels."""
    import models.common as mc
    from mmidet_hip import fusion_ops as F2
    from mmidet_hip import ops
    d = dev()
    torch.manual_seed(5)
    B, C, heads = 4, 256, 8
    blk = mc.myTransformerBlock(C, C, C, heads, 4, 0.1, 0.1).to(d).train()
    for p in blk.parameters():
        p.data.add_(0.05 * torch.randn_like(p))
    x = torch.randn(B, 128, C, device=d)
    go = torch.randn(B, 128, C, device=d)
    ps, seeds, eps = (0.1, 0.2, 0.3), (11, 222, 3333), (1e-5, 1e-5)
    prm = _block_params(blk)

    xf = x.clone().requires_grad_()
    yf = F2._TransformerBlock.apply(xf, heads, ps, eps, seeds, *prm)
    yf.backward(go)
    fused = [xf.grad.clone()] + [p.grad.clone() for p in prm]
    for p in prm:
        p.grad = None

    xu = x.clone().requires_grad_()
    g1, b1, wq, bq, wk, bk, wv, bv, wo, bo, g2, b2, w1, c1, w2, c2 = prm
    ln = F2.layernorm(xu, g1, b1, eps[0])
    att = F2._Attention.apply(ops.linear(ln, wq, bq), ops.linear(ln, wk, bk), ops.linear(ln, wv, bv), heads, ps[0], seeds[0])
    x1 = ops.add(xu, F2._DropoutAdd.apply(ops.linear(att, wo, bo), None, ps[1], seeds[1]))
    hdn = F2.gelu(ops.linear(F2.layernorm(x1, g2, b2, eps[1]), w1, c1))
    yu = ops.add(x1, F2._DropoutAdd.apply(ops.linear(hdn, w2, c2), None, ps[2], seeds[2]))
    yu.backward(go)
    torch.cuda.synchronize()
    assert (yf == 0).float().mean() < 0.01 and not torch.equal(yf, x)
    close(yf, yu, tol=1e-5, what="out")
    names = ['dx', 'ln1.w', 'ln1.b', 'wq', 'bq', 'wk', 'bk', 'wv', 'bv', 'wo', 'bo', 'ln2.w', 'ln2.b', 'w1', 'b1', 'w2', 'b2']
    for n, a, b in zip(names, fused, [xu.grad] + [p.grad for p in prm]):
        if n != 'bk':                   # (true gradient 0, see above)
            close(a, b, tol=1e-4, what=n)


@pytest.mark.parametrize('hw,C', [((40, 24), 32), ((20, 20), 64)])
def test_fan_out_skip_forms_add_the_second_gradient_in_kernel(hw, C):
    """pool_tokens(skip=True) and upsample2x(skip=True) hand their input on as an alias; the gradient of the alias' consumer is
    added inside the backward kernel (mmi_avgpool8_bwd_acc / mmi_upsample2x_bwd_acc, the latter reading a channel SLICE of a
    concat gradient through its row stride).  Same numbers as letting autograd accumulate: a + b is one rounding either way."""
    from mmidet_hip import fusion_ops as F2
    from mmidet_hip import ops
    H, W = hw
    d = dev()
    g = torch.Generator().manual_seed(H + C)
    rgb, ir = torch.randn(2, H, W, C, generator=g).to(d), torch.randn(2, H, W, C, generator=g).to(d)
    gt = torch.randn(2, 128, C, generator=g).to(d)
    ga, gb = torch.randn(2, H, W, C, generator=g).to(d), torch.randn(2, H, W, C, generator=g).to(d)
    res = []
    for skip in (False, True):
        r, i = rgb.clone().requires_grad_(), ir.clone().requires_grad_()
        if skip:
            tok, ra, ia = F2.pool_tokens(r, i, True)
        else:
            tok, ra, ia = F2.pool_tokens(r, i), r, i
        ((tok * gt).sum() + (ops.add(ra, ia) * ga).sum() + (ra * gb).sum()).backward()
        res.append((tok.detach(), r.grad, i.grad))
    assert torch.equal(res[0][0], res[1][0])
    close(res[1][1], res[0][1], what='d rgb', tol=1e-6)
    close(res[1][2], res[0][2], what='d ir', tol=1e-6)
    # upsample: the incoming gradient is the middle channel slice of a wider (concat) gradient
    x = torch.randn(2, H, W, C, generator=g).to(d)
    other = torch.randn(2, 2 * H, 2 * W, C // 2, generator=g).to(d)
    gcat = torch.randn(2, 2 * H, 2 * W, C + C // 2, generator=g).to(d)
    gx = torch.randn(2, H, W, C, generator=g).to(d)
    res = []
    for skip in (False, True):
        xg = x.clone().requires_grad_()
        if skip:
            up, xa = ops.upsample2x(xg, True)
        else:
            up, xa = ops.upsample2x(xg), xg
        cat = ops.concat([other, up])
        ((cat * gcat).sum() + (xa * gx).sum()).backward()
        res.append(xg.grad)
    ref = F.avg_pool2d(nchw(gcat[..., C // 2:]), 2) * 4 + nchw(gx)
    close(nchw(res[1]), ref, what='upsample dx vs torch', tol=1e-6)
    close(res[1], res[0], what='upsample dx', tol=1e-6)
