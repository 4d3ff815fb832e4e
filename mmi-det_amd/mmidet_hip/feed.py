"""Input side of the training step: pinned-memory, double-buffered host -> device feed of the loader's
paired batches.

The reference moves each batch with `imgs.to(device, non_blocking=True).float() / 255.0` (train.py:743) from a DataLoader
with pin_memory=True (utils/datasets.py:1604-1611): the copy is asynchronous only if the source is pinned, and the fp32
conversion + /255 + RGB/IR split then cost three more passes over the batch on the device.  Here the uint8 batch is what
travels (39 MB at B=16 640x640 instead of 157 MB of fp32) and `/255`, the split and the NHWC re-layout are one kernel
(mmi_u8_pair_to_nhwc, run by TrainStep).  The feeder keeps `depth` slots, each a pinned host buffer + a device buffer + two
events: the copy of batch k+1 is enqueued on a dedicated copy stream while the compute stream is still working on batch k,
a slot's device buffer is overwritten only after the compute stream has passed the step that read it, and its pinned buffer
only after the copy that read it has completed.

    for imgs_u8, targets in PairedBatchFeeder(loader, device):      # loader yields (uint8 (B,6,H,W), float32 (nT,6)) on the host
        ts.step(imgs_u8, targets)
"""
import torch


class _Slot:
    __slots__ = ('pin_img', 'dev_img', 'pin_tgt', 'dev_tgt', 'ready', 'free', 'used')

    def __init__(self):
        self.pin_img = self.dev_img = self.pin_tgt = self.dev_tgt = None
        self.ready = self.free = None
        self.used = False


class PairedBatchFeeder:
    def __init__(self, loader, device, depth=2):
        assert depth >= 2, 'double buffering needs two slots'
        self.loader, self.device, self.depth = loader, torch.device(device), depth
        self.cuda = self.device.type == 'cuda'
        self.slots = [_Slot() for _ in range(depth)]
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.bytes_copied = 0

    @staticmethod
    def _fit(buf, t, headroom=1.0, **kw):
        """(buffer of at least t.numel() elements of t's dtype (grow-only), whether it was (re)allocated).  `headroom` > 1 sizes
        a new buffer generously (the target list changes length from batch to batch: no regrowth per batch)."""
        if buf is None or buf.numel() < t.numel() or buf.dtype != t.dtype:
            return torch.empty(max(int(t.numel() * headroom), 1), dtype=t.dtype, **kw), True
        return buf, False

    def _stage(self, k, batch):
        imgs, targets = batch[0], batch[1]
        assert imgs.dtype == torch.uint8 and imgs.dim() == 4 and imgs.shape[1] == 6, 'loader batches are uint8 (B,6,H,W)'
        targets = targets.float()
        if not self.cuda:
            return imgs, targets, None
        s = self.slots[k]
        if s.ready is not None:
            s.ready.synchronize()               # the copy that last read this slot's pinned buffers has completed
        s.pin_img, _ = self._fit(s.pin_img, imgs, pin_memory=True)
        s.pin_tgt, _ = self._fit(s.pin_tgt, targets, headroom=2.0, pin_memory=True)
        s.dev_img, new_i = self._fit(s.dev_img, imgs, device=self.device)
        s.dev_tgt, new_t = self._fit(s.dev_tgt, targets, headroom=2.0, device=self.device)
        if new_i or new_t:
            # A fresh device buffer comes from the caching allocator of the COMPUTE stream (the stream current here): the block
            # may be one the host has already freed while kernels queued on the compute stream still use it.  Its first write is
            # the copy below, on the copy stream, so that copy must be ordered behind everything the compute stream has queued
            # (later uses of the slot are ordered by `free`).
            self.copy_stream.wait_stream(torch.cuda.current_stream())
        pi, pt = s.pin_img[:imgs.numel()].view(imgs.shape), s.pin_tgt[:targets.numel()].view(targets.shape)
        pi.copy_(imgs)                          # (a loader with pin_memory=True makes this a pinned -> pinned memcpy)
        pt.copy_(targets)
        di, dt = s.dev_img[:imgs.numel()].view(imgs.shape), s.dev_tgt[:targets.numel()].view(targets.shape)
        with torch.cuda.stream(self.copy_stream):
            if s.free is not None:
                self.copy_stream.wait_event(s.free)     # the step that read this slot's device buffers has been passed
            di.copy_(pi, non_blocking=True)
            dt.copy_(pt, non_blocking=True)
            if s.ready is None:
                s.ready = torch.cuda.Event()
            s.ready.record(self.copy_stream)
        self.bytes_copied += imgs.numel() + targets.numel() * 4
        return di, dt, s

    def __iter__(self):
        it = iter(self.loader)
        k = 0
        try:
            nxt = self._stage(k, next(it))
        except StopIteration:
            return
        while nxt is not None:
            imgs, targets, slot = nxt
            if slot is not None:
                torch.cuda.current_stream().wait_event(slot.ready)     # compute waits for ITS batch only
            k = (k + 1) % self.depth
            try:
                nxt = self._stage(k, next(it))                         # the next batch's copy is in flight during this step
            except StopIteration:
                nxt = None
            yield imgs, targets
            if slot is not None:
                if slot.free is None:
                    slot.free = torch.cuda.Event()
                slot.free.record(torch.cuda.current_stream())          # everything the consumer enqueued on this batch
