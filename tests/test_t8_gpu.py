"""GPU: pre-split ("T8") operands of the three-term bf16 GEMM modes (include/mmidet_hip.h: mmi_split_t8, mmi_gemm_operands_t8;
csrc/t8.h).  The image holds exactly the terms the kernels' own split produces, so every GEMM on images must equal the same GEMM
splitting in the kernel BIT FOR BIT -- forward, data gradient and weight gradient, both three-term modes, single and stream-K
schedules -- and the image itself is pinned against a plain torch restatement of the split."""
import pytest
import torch

from test_ops_gpu import dev

pytestmark = pytest.mark.gpu


def torch_t8(x):
    """x (..., C) fp32 -> (..., C/8, 3, 8) bf16: t0 = bf16(x), t1 = bf16(x - t0), t2 = bf16(x - t0 - t1)."""
    t0 = x.bfloat16()
    r1 = x - t0.float()
    t1 = r1.bfloat16()
    r2 = r1 - t1.float()
    t2 = r2.bfloat16()
    c = x.shape[-1]
    return torch.stack([t.reshape(*x.shape[:-1], c // 8, 8) for t in (t0, t1, t2)], dim=-2)


def test_image_is_the_three_term_split():
    from mmidet_hip import ops
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(37, 5, 64, generator=g) * torch.logspace(-6, 6, 64)).to(dev())
    img = ops.t8_image(x)
    ref = torch_t8(x)
    assert img.shape == ref.shape and torch.equal(img.view(torch.int16), ref.view(torch.int16))
    # the three terms represent every fp32 value exactly
    assert torch.equal(img.float().sum(dim=-2).reshape(x.shape), x)
    # a channel slice of a wider buffer (row stride > C) addresses its image like the tensor
    wide = torch.randn(11, 96, generator=g).to(dev())
    sl = wide[:, 32:64]
    assert torch.equal(ops.t8_image(sl).view(torch.int16), torch_t8(sl.contiguous()).view(torch.int16))


SHAPES = [(4, 40, 40, 128, 128, 3, 1),       # 64x64 tiles, one workgroup per tile
          (16, 40, 40, 128, 256, 3, 1),      # stream-K, 128-wide tiles
          (4, 40, 40, 64, 128, 3, 2),        # stride 2: parity-class dgrad
          (4, 20, 20, 256, 128, 1, 1),       # 1x1: linear loaders
          (2, 17, 19, 32, 64, 3, 1),         # ragged tiles
          (16, 80, 80, 128, 128, 3, 1)]      # BASELINE-size layer


@pytest.mark.parametrize('mode', [2, 3], ids=['bf16x6', 'bf16x9'])
@pytest.mark.parametrize('shape', SHAPES)
def test_gemms_on_images_equal_the_in_kernel_split_bit_for_bit(shape, mode):
    from mmidet_hip import alloc, lib, ops
    N, H, W, Cin, Cout, k, s = shape
    d = dev()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, H, W, Cin, generator=g).to(d)
    w = (torch.randn(Cout, k, k, Cin, generator=g) / (Cin * k * k) ** 0.5).to(d)
    desc = ops._desc((N, H, W, Cin), Cout, k, s, Cin, Cout)
    dy = torch.randn(N, desc.Ho, desc.Wo, Cout, generator=g).to(d)
    x8, w8, dy8 = ops.t8_image(x), ops.t8_image(w.reshape(Cout, -1)), ops.t8_image(dy)
    tb = lib.conv_wgrad_table_bytes(desc)
    tab = None
    if tb:
        tab = torch.empty(tb, dtype=torch.uint8, device=d)
        lib.conv_wgrad_table_build(tab.data_ptr(), desc, st)

    def run(a8, b8, w_only=False):
        y = alloc.empty((N, desc.Ho, desc.Wo, Cout), dtype=torch.float32, device=d)
        dx, dw = alloc.empty_like(x), alloc.empty_like(w)
        nb = lib.conv_wgrad_workspace(desc)
        ws = torch.zeros(max(nb // 4, 1), device=d)
        P = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        if b8 is not None:
            lib.gemm_operands_t8(P(a8), P(b8), None, None)
        ops.conv_fwd(x, w, None, y, None, desc, st)
        if b8 is not None:
            lib.gemm_operands_t8(P(dy8) if a8 is not None else None, P(b8), None, None)
        ops.conv_dgrad(dy, w, dx, desc, st)
        if a8 is not None:
            lib.gemm_operands_t8(P(dy8), P(a8), None, None)
        lib.conv_wgrad_tab(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), nb, P(tab), desc, st)
        torch.cuda.synchronize()
        return y, dx, dw

    lib.set_gemm_precision(mode)
    try:
        base = run(None, None)
        both = run(x8, w8)
        wonly = run(None, w8)
    finally:
        lib.set_gemm_precision(0)
    for name, a, b, c in zip(('y', 'dx', 'dw'), base, both, wonly):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b), '%s: images of both operands differ from the in-kernel split' % name
        assert torch.equal(a, c), '%s: weight image alone differs from the in-kernel split' % name


def test_announced_images_are_consumed_by_the_next_launch_only():
    """mmi_gemm_operands_t8 applies to ONE launch: a following launch without an announcement must not read a stale image."""
    from mmidet_hip import alloc, lib, ops
    d = dev()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(9)
    x = torch.randn(4, 20, 20, 64, generator=g).to(d)
    w = (torch.randn(64, 3, 3, 64, generator=g) / 24).to(d)
    desc = ops._desc((4, 20, 20, 64), 64, 3, 1, 64, 64)
    junk = torch.full((4, 20, 20, 8, 3, 8), 7.0, dtype=torch.bfloat16, device=d)
    wj = torch.full((64, 72, 3, 8), 7.0, dtype=torch.bfloat16, device=d)
    lib.set_gemm_precision(3)
    try:
        y0 = alloc.empty((4, 20, 20, 64), dtype=torch.float32, device=d)
        ops.conv_fwd(x, w, None, y0, None, desc, st)
        y1, y2 = alloc.empty_like(y0), alloc.empty_like(y0)
        lib.gemm_operands_t8(junk.data_ptr(), wj.data_ptr(), None, None)
        ops.conv_fwd(x, w, None, y1, None, desc, st)        # reads the (wrong) images
        ops.conv_fwd(x, w, None, y2, None, desc, st)        # must not
        torch.cuda.synchronize()
    finally:
        lib.set_gemm_precision(0)
    assert not torch.equal(y1, y0) and torch.equal(y2, y0)


@pytest.mark.parametrize('mode', [0, 3], ids=['fp32', 'bf16x9'])
@pytest.mark.parametrize('shape', SHAPES)
def test_deep_prefetch_k_loop_is_bit_identical(shape, mode):
    """mmi_set_deep_prefetch(1): double-buffered LDS tile, global loads two slabs ahead -- the same products summed in the same order."""
    from mmidet_hip import alloc, lib, ops
    N, H, W, Cin, Cout, k, s = shape
    d = dev()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(6)
    x = torch.randn(N, H, W, Cin, generator=g).to(d)
    w = (torch.randn(Cout, k, k, Cin, generator=g) / (Cin * k * k) ** 0.5).to(d)
    desc = ops._desc((N, H, W, Cin), Cout, k, s, Cin, Cout)
    dy = torch.randn(N, desc.Ho, desc.Wo, Cout, generator=g).to(d)
    part = torch.empty((lib.conv_fwd_row_blocks(desc) + 64) * 2 * Cout, device=d)

    def run():
        y = alloc.empty((N, desc.Ho, desc.Wo, Cout), dtype=torch.float32, device=d)
        dx = alloc.empty_like(x)
        ops.conv_fwd(x, w, None, y, part, desc, st)
        ops.conv_dgrad(dy, w, dx, desc, st)
        torch.cuda.synchronize()
        return y, dx, part.clone()

    lib.set_gemm_precision(mode)
    try:
        base = run()
        lib.set_deep_prefetch(1)
        deep = run()
    finally:
        lib.set_deep_prefetch(0)
        lib.set_gemm_precision(0)
    for name, a, b in zip(('y', 'dx', 'BN partials'), base, deep):
        n = a.numel() if name != 'BN partials' else lib.conv_fwd_row_blocks(desc) * 2 * Cout
        assert torch.equal(a.reshape(-1)[:n], b.reshape(-1)[:n]), name
