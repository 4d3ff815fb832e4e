"""Every output / workspace tensor of the hot path is allocated here (torch's caching allocator underneath).

Normal mode: the three names below ARE torch.empty / torch.empty_like / torch.empty_strided.

MMIDET_POISON=1 (the `-m gpu` test suite sets it, tests/conftest.py): a kernel that leaves an output element unwritten, or
writes outside its output, cannot hide behind whatever the reused block happened to hold:
  * every byte of a fresh tensor is 0xFF -- a NaN in fp32 / bf16 / fp16 / fp64, -1 in the integer types -- so an element no
    kernel stored shows up in the first comparison that reads it;
  * the tensor sits between two guard zones of GUARD_BYTES of the same pattern, and every block handed out is kept alive (up to
    QUARANTINE_BYTES) until `check_guards()` has looked at its zones: an out-of-range store of any later kernel is reported with
    the allocation site instead of landing in a neighbour;
  * `check_counters()` (ops.py) asserts that the arrival counters at the head of every zero-initialised workspace are zero
    again, i.e. that no last-arriver election was left half done.
tests/conftest.py runs both checks after every GPU test.  Nothing here is on the product path in normal mode.
"""
import os
import traceback
import weakref

import torch

POISON = os.environ.get('MMIDET_POISON', '0') == '1'
GUARD_BYTES = 1024                      # each side; keeps fp32 tensors 1 KiB aligned
QUARANTINE_BYTES = int(float(os.environ.get('MMIDET_POISON_QUARANTINE_GB', '48')) * (1 << 30))

_held = []          # (base tensor, guard elements, allocation site): everything handed out since the last check
_held_bytes = 0
_persist = []       # (weakref of base, guard elements, site): long-lived workspaces, looked at by EVERY check while they live
stats = {'allocs': 0, 'checked': 0, 'checks': 0}


def _site():
    for fr in reversed(traceback.extract_stack(limit=8)[:-3]):
        if os.path.basename(fr.filename) != 'alloc.py':
            return '%s:%d' % (os.path.basename(fr.filename), fr.lineno)
    return '?'


def _storage_elems(shape, stride):
    n = 1
    for s, st in zip(shape, stride):
        if s == 0:
            return 0
        n += (s - 1) * st
    return n


def _guarded(shape, stride, dtype, device, persistent=False):
    global _held_bytes
    shape, stride = tuple(int(s) for s in shape), tuple(int(s) for s in stride)
    item = torch.empty((), dtype=dtype).element_size()
    g = GUARD_BYTES // item
    n = _storage_elems(shape, stride)
    base = torch.empty(n + 2 * g, dtype=dtype, device=device)
    base.view(torch.uint8).fill_(0xFF)
    stats['allocs'] += 1
    if base.is_cuda and persistent:
        _persist.append((weakref.ref(base), g, _site()))
    elif base.is_cuda:
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing and _held_bytes + base.numel() * item > QUARANTINE_BYTES:
            check_guards()      # (synchronises; everything allocated so far has been looked at and may go)
        _held.append((base, g, _site()))
        _held_bytes += base.numel() * item
    return base.as_strided(shape, stride, g)


def _p_empty(*size, dtype=None, device=None, **kw):
    assert not kw, 'alloc.empty: unsupported arguments %s' % sorted(kw)
    if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
        size = tuple(size[0])
    dtype = dtype or torch.get_default_dtype()
    stride, acc = [], 1
    for s in reversed(size):
        stride.append(acc)
        acc *= max(int(s), 1)
    return _guarded(size, tuple(reversed(stride)), dtype, device)


def _p_empty_like(t, **kw):
    assert not kw, 'alloc.empty_like: unsupported arguments %s' % sorted(kw)
    m = torch.empty_like(t, device='meta')          # torch's own layout rule (dense permuted inputs keep their strides)
    return _guarded(m.shape, m.stride(), t.dtype, t.device)


def _p_empty_strided(size, stride, dtype=None, device=None):
    return _guarded(size, stride, dtype or torch.get_default_dtype(), device)


def workspace(n, dtype, device, zero=False):
    """A long-lived 1-D workspace (ops.scratch / ops.zeroed_scratch / pixel tables): in poison mode its guard zones are checked
    by every check_guards() for as long as it lives."""
    if not POISON:
        return torch.zeros(n, dtype=dtype, device=device) if zero else torch.empty(n, dtype=dtype, device=device)
    t = _guarded((int(n),), (1,), dtype, device, persistent=True)
    if zero:
        t.zero_()
    return t


def check_guards():
    """Synchronise, verify the guard zones of every block handed out since the last call, release them.  Raises on the first
    damaged zone with the allocation site."""
    global _held_bytes
    live = []
    for ref, g, site in _persist:
        b = ref()
        if b is not None:
            live.append((b, g, site))
    _persist[:] = [(weakref.ref(b), g, site) for b, g, site in live]
    blocks = _held + live
    if not blocks:
        return 0
    torch.cuda.synchronize()
    bad = []
    # one reduction per block and side would be thousands of tiny launches + syncs: gather the zones first
    zones = []
    for base, g, site in blocks:
        u = base.view(torch.uint8)
        zones.append(u[:GUARD_BYTES])
        zones.append(u[-GUARD_BYTES:])
    CH = 4096
    for c0 in range(0, len(zones), CH):
        z = torch.stack(zones[c0:c0 + CH])                    # (k, GUARD_BYTES)
        ok = (z == 0xFF).all(dim=1).cpu()
        for j in torch.nonzero(~ok).flatten().tolist():
            base, g, site = blocks[(c0 + j) // 2]
            side = 'front' if (c0 + j) % 2 == 0 else 'back'
            zz = zones[c0 + j].cpu()
            first = int(torch.nonzero(zz != 0xFF)[0])
            bad.append('%s guard of a %d-element %s block allocated at %s (first damaged byte %d)'
                       % (side, base.numel() - 2 * g, str(base.dtype).replace('torch.', ''), site, first))
    n = len(blocks)
    del blocks, live
    stats['checked'] += n
    stats['checks'] += 1
    _held.clear()
    _held_bytes = 0
    if bad:
        raise AssertionError('out-of-range device stores: ' + '; '.join(bad[:8]))
    return n


if POISON:
    empty, empty_like, empty_strided = _p_empty, _p_empty_like, _p_empty_strided
else:
    empty, empty_like, empty_strided = torch.empty, torch.empty_like, torch.empty_strided
