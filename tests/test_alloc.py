"""mmidet_hip/alloc.py: the allocation layer of the hot path and its poison mode (MMIDET_POISON=1, what the GPU suite runs under).
CPU part: shapes, strides and fill of the poisoned forms.  GPU part: the detectors detect -- a store into a guard zone and a
counter left non-zero both fail check_guards() / check_counters() by name."""
import pytest
import torch


def test_poisoned_forms_keep_torchs_shapes_and_strides_and_are_all_ones_bytes():
    from mmidet_hip import alloc
    a = alloc._p_empty((3, 5, 7), dtype=torch.float32)
    assert a.shape == (3, 5, 7) and a.is_contiguous() and bool(torch.isnan(a).all())
    b = alloc._p_empty(4, 6, dtype=torch.int64)
    assert b.shape == (4, 6) and bool((b == -1).all())
    src = torch.zeros(2, 4, 6, 8).permute(0, 2, 3, 1)                      # dense, permuted: empty_like keeps the strides
    c = alloc._p_empty_like(src)
    assert c.shape == src.shape and c.stride() == src.stride() == torch.empty_like(src).stride()
    w = torch.zeros(16, 8, 3, 3).contiguous(memory_format=torch.channels_last)
    d = alloc._p_empty_strided(w.shape, w.stride(), dtype=w.dtype)
    assert d.stride() == w.stride() and d.is_contiguous(memory_format=torch.channels_last) and bool(torch.isnan(d).all())
    e = alloc._p_empty((0, 5), dtype=torch.bfloat16)
    assert e.numel() == 0
    # writes through the view stay inside: the guard zones of the base keep their pattern
    base = a._base
    a.zero_()
    u = base.view(torch.uint8)
    assert bool((u[:alloc.GUARD_BYTES] == 0xFF).all()) and bool((u[-alloc.GUARD_BYTES:] == 0xFF).all())
    assert a.data_ptr() % 16 == 0


@pytest.mark.gpu
def test_a_store_outside_an_output_is_reported_with_its_allocation_site():
    from mmidet_hip import alloc
    if not alloc.POISON:
        pytest.skip('poison mode is off (MMIDET_POISON=0)')
    alloc.check_guards()
    t = alloc.empty((4, 256), dtype=torch.float32, device='cuda:0')
    t.zero_()
    alloc.check_guards()                                   # clean
    t2 = alloc.empty((4, 256), dtype=torch.float32, device='cuda:0')
    t2._base[t2._base.numel() - 3] = 1.0                   # what a kernel writing one row too many would do
    with pytest.raises(AssertionError, match='back guard.*test_alloc.py'):
        alloc.check_guards()
    alloc.check_guards()                                   # (the damaged block was released with the report)


@pytest.mark.gpu
def test_a_counter_left_nonzero_is_reported():
    from mmidet_hip import alloc, ops
    if not alloc.POISON:
        pytest.skip('poison mode is off (MMIDET_POISON=0)')
    dev = torch.device('cuda:0')
    ws = ops.zeroed_scratch(1 << 20, dev, stream=12345, tag='w')
    ops.check_counters()
    ws[100] = 1                                            # an election that did not reset its counter
    try:
        with pytest.raises(AssertionError, match="counter bytes in the 'w' workspace"):
            ops.check_counters()
    finally:
        ws[100] = 0
        del ops._zeroed[(dev, 'w', 12345)]
