"""Detection-loss device ops over the C ABI (utils/loss.py:113-245 of the reference)."""
import torch

from . import alloc, lib
from .ops import _stream

_grid_cache = {}


def _grids_dev(grids, device):
    key = (tuple(grids), device)
    t = _grid_cache.get(key)
    if t is None:
        t = torch.tensor(grids, dtype=torch.int32, device=device).contiguous()
        _grid_cache[key] = t
    return t


def build_targets_raw(targets, anchors, grids, anchor_t):
    """Fixed-capacity device outputs + device counts (no host sync).  Returns (idx, tcls, tbox, anch, counts, cap):
    idx (nl,4,cap) int64, tcls (nl,cap) int64, tbox (nl,cap,4), anch (nl,cap,2), counts (nl,) int32."""
    assert targets.is_cuda and targets.dtype == torch.float32
    targets = targets.contiguous()
    anchors = anchors.contiguous()
    nl, na = anchors.shape[0], anchors.shape[1]
    nt = targets.shape[0]
    cap = 5 * na * nt
    dev = targets.device
    idx = alloc.empty((nl, 4, max(cap, 1)), dtype=torch.int64, device=dev)
    tcls = alloc.empty((nl, max(cap, 1)), dtype=torch.int64, device=dev)
    tbox = alloc.empty((nl, max(cap, 1), 4), dtype=torch.float32, device=dev)
    anch = alloc.empty((nl, max(cap, 1), 2), dtype=torch.float32, device=dev)
    counts = alloc.empty((nl,), dtype=torch.int32, device=dev)
    if cap == 0:  # capacity-1 dummies keep the pointer arithmetic valid; the kernel derives strides from nt
        idx, tcls, tbox, anch = idx[:, :, :0], tcls[:, :0], tbox[:, :0], anch[:, :0]
    lib.build_targets(targets.data_ptr(), nt, anchors.data_ptr(), nl, na, _grids_dev(grids, dev).data_ptr(),
                      float(anchor_t), idx.data_ptr(), tcls.data_ptr(), tbox.data_ptr(), anch.data_ptr(),
                      counts.data_ptr(), _stream())
    return idx, tcls, tbox, anch, counts, cap


def build_targets(targets, anchors, grids, anchor_t):
    """Reference-shaped result (lists of exact-length tensors); reading the counts is the one host sync, exactly where
    the reference syncs at its first boolean mask."""
    idx, tcls, tbox, anch, counts, cap = build_targets_raw(targets, anchors, grids, anchor_t)
    n = counts.tolist()
    nl = len(n)
    return ([tcls[l, :n[l]] for l in range(nl)], [tbox[l, :n[l]] for l in range(nl)],
            [tuple(idx[l, k, :n[l]] for k in range(4)) for l in range(nl)], [anch[l, :n[l]] for l in range(nl)])
