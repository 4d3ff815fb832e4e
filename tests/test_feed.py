"""The pinned, double-buffered host -> device feeder (mmidet_hip/feed.py): batches arrive intact and in
order, ragged target counts and a changing batch size included; on the GPU the copies overlap the consumer's work."""
import pytest
import torch


def _batches(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n):
        b = 2 + (i % 3)
        imgs = torch.randint(0, 256, (b, 6, 32, 64), dtype=torch.uint8, generator=g)
        tg = torch.rand(3 + 5 * (i % 4), 6, generator=g)
        out.append((imgs, tg))
    return out


def test_feeder_cpu_passthrough():
    from mmidet_hip.feed import PairedBatchFeeder
    src = _batches(5)
    got = list(PairedBatchFeeder(iter(src), 'cpu'))
    assert len(got) == 5
    for (a, t), (b, u) in zip(src, got):
        assert torch.equal(a, b) and torch.equal(t, u)
    assert list(PairedBatchFeeder(iter([]), 'cpu')) == []


@pytest.mark.gpu
def test_feeder_gpu_delivers_every_batch_intact():
    from mmidet_hip.feed import PairedBatchFeeder
    assert torch.cuda.is_available()
    dev = torch.device('cuda:0')
    src = _batches(9, seed=3)
    feeder = PairedBatchFeeder(iter(src), dev)
    sums = []
    spin = torch.randn(2048, 2048, device=dev)
    for imgs, tg in feeder:
        assert imgs.is_cuda and imgs.dtype == torch.uint8 and tg.is_cuda
        # consumer work that keeps the compute stream busy while the next copy is in flight; reads the buffers at the END
        for _ in range(4):
            spin = torch.tanh(spin @ spin * 1e-3)
        sums.append((imgs.clone(), tg.clone()))
    torch.cuda.synchronize()
    assert len(sums) == len(src)
    for (a, t), (b, u) in zip(src, sums):
        assert torch.equal(a, b.cpu()) and torch.equal(t, u.cpu())
    assert feeder.bytes_copied == sum(a.numel() + t.numel() * 4 for a, t in src)


@pytest.mark.gpu
def test_feeder_drives_the_training_step():
    """uint8 batches from host memory straight into TrainStep (which fuses /255 + split + NHWC): same losses as with the
    batches already on the device."""
    from mmidet_hip.feed import PairedBatchFeeder
    from test_step_gpu import make
    from oracle import portable_init
    m1, ts1, cfg = make()
    m2, ts2, _ = make()
    host = [portable_init.synth_batch(2, 128, cfg['nc'], per_image=4, seed=200 + i) for i in range(4)]
    l_dev = [ts1.step(a.cuda(), t.cuda())[0].clone() for a, t in host]
    l_fed = [ts2.step(a, t)[0].clone() for a, t in PairedBatchFeeder(iter(host), 'cuda:0')]
    torch.cuda.synchronize()
    for x, y in zip(l_dev, l_fed):
        assert torch.equal(x, y)
