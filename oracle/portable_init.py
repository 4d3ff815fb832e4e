"""Oracle helper (test infrastructure): portable deterministic weights.

A state_dict is filled from an integer hash of (tensor name, flat index) -> fp32, so the build container
(where the reference is importable) and the GPU box (where it is not) regenerate bit-identical weights
without relying on torch RNG streams.  207.9 M parameters are therefore never committed.
"""
import re
import zlib

import numpy as np
import torch

_SKIP = ('sobel_weight', 'anchors', 'anchor_grid', 'num_batches_tracked')


def _u01(name, n):
    """splitmix64 of (crc32(name) << 32 | index) -> uniform [0,1) with 24 random bits (exact in fp32)."""
    with np.errstate(over='ignore'):
        z = (np.uint64(zlib.crc32(name.encode())) << np.uint64(32)) + np.arange(n, dtype=np.uint64)
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32)


def fill_(state_dict):
    """In-place portable init of every floating tensor of a (reference-keyed) state_dict. Returns it."""
    for name, t in state_dict.items():
        if not t.dtype.is_floating_point or any(s in name for s in _SKIP):
            continue
        u = torch.from_numpy(_u01(name, t.numel())).view(t.shape)
        leaf = name.rsplit('.', 1)[-1]
        if 'running_var' in name:
            v = 0.5 + u                                    # (0.5, 1.5)
        elif 'running_mean' in name:
            v = (u - 0.5) * 0.2
        elif t.dim() == 1 and leaf == 'weight':           # BN / LayerNorm scale
            v = 0.5 + u
        elif leaf == 'bias':
            if re.match(r'^model\.\d+\.m\.\d+\.bias$', name):   # Detect head bias: hash jitter around a fixed prior
                v = ((u - 0.5) * 0.2).view(3, -1)           # na=3 rows of (x,y,w,h,obj,cls...)
                v[:, 4] -= 4.5                              # objectness / class priors like yolo_test.py:280-290
                v[:, 5:] -= 2.0
                v = v.reshape(-1)
            else:
                v = (u - 0.5) * 0.2
        elif leaf == 'sobel_factor':
            v = 0.5 + u
        elif leaf == 'pos_emb':
            v = (u - 0.5) * 0.2
        elif t.dim() == 4:                                 # conv weight: U(+-sqrt(3/fan_in)) -> unit-variance-preserving
            fan_in = t.shape[1] * t.shape[2] * t.shape[3]
            v = (u - 0.5) * 2 * (3.0 / fan_in) ** 0.5
        elif t.dim() == 2:                                 # Linear weight
            v = (u - 0.5) * 2 * (1.0 / t.shape[1]) ** 0.5
        else:
            v = (u - 0.5) * 0.2
        t.copy_(v)
    return state_dict


def synth_batch(bs, size, nc, per_image=8, seed=0):
    """Synthetic paired batch in the loader's wire format: uint8 (B,6,S,S) and
    targets (nT,6)=[img,cls,xc,yc,w,h]; hash-generated so it is identical on every machine."""
    img = (_u01('imgs:%d' % seed, bs * 6 * size * size) * 256).astype(np.uint8).reshape(bs, 6, size, size)
    nt = bs * per_image
    u = _u01('targets:%d' % seed, nt * 5).reshape(nt, 5)
    t = np.zeros((nt, 6), np.float32)
    t[:, 0] = np.repeat(np.arange(bs), per_image)
    t[:, 1] = np.floor(u[:, 0] * nc)
    t[:, 2:4] = 0.1 + 0.8 * u[:, 1:3]
    t[:, 4:6] = 0.02 + 0.30 * u[:, 3:5]
    return torch.from_numpy(img), torch.from_numpy(t)


def signs(n, name):
    """A fixed +-1 vector (float64 torch tensor) from an integer hash of (name, index): random projections of gradient
    tensors that both sides of a fixture can regenerate (oracle/gen_fp64_fullsize.py)."""
    with np.errstate(over='ignore'):
        z = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(zlib.crc32(name.encode()) * 0x10001 + 12345)
        z = (z ^ (z >> np.uint64(29))) * np.uint64(0xBF58476D1CE4E5B9)
        bit = (z >> np.uint64(40)) & np.uint64(1)
    return torch.from_numpy(bit.astype(np.float64) * 2.0 - 1.0)
