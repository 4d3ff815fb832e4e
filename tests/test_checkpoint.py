"""Checkpoint compatibility, CPU: whole-object pickles as train.py:881-899 writes them, loaded back
through models/experimental.py::attempt_load; a checkpoint written by the REAL reference when it is around (build
container only -- the GPU box has no /root/reference and skips that case)."""
import contextlib
import io
import os
import sys
from copy import deepcopy

import pytest
import torch

from conftest import tiny_cfg


def _native(kind='fourier'):
    from models.yolo_test import Model
    from oracle import portable_init
    m = Model(tiny_cfg(kind))
    m.load_state_dict(portable_init.fill_(m.state_dict()))
    m.names = ['c%d' % i for i in range(m.yaml['nc'])]
    return m


def test_native_checkpoint_round_trip(tmp_path):
    from models.experimental import attempt_load
    from utils.torch_utils import ModelEMA, intersect_dicts
    m = _native()
    ema = ModelEMA(m)
    f = str(tmp_path / 'last.pt')
    torch.save({'epoch': 3, 'model': deepcopy(m).half(), 'ema': deepcopy(ema.ema).half(), 'updates': ema.updates,
                'optimizer': None}, f)                                   # train.py:881-899
    ckpt = torch.load(f, map_location='cpu', weights_only=False)
    assert type(ckpt['model']).__module__ == 'models.yolo_test'
    # resume path of train.py:521-531: fresh model from the stored yaml, intersecting state dicts
    from models.yolo_test import Model
    fresh = Model(ckpt['model'].yaml)
    sd = intersect_dicts(ckpt['model'].float().state_dict(), fresh.state_dict(), exclude=['anchor'])
    assert len(sd) == len([k for k in fresh.state_dict() if 'anchor' not in k])
    fresh.load_state_dict(sd, strict=False)
    # inference path of test.py:60 / detect_twostream.py:33
    fused = attempt_load(f)
    assert not fused.training and not any(hasattr(mod, 'bn') for mod in fused.model.modules() if type(mod).__name__ == 'Conv')
    loaded = attempt_load(f, fuse=False)
    assert not loaded.training and loaded.names == m.names
    ref = {k: v.half().float() for k, v in ema.ema.state_dict().items()}
    for k, v in loaded.state_dict().items():
        if v.dtype.is_floating_point:
            assert torch.equal(v, ref[k]), k
    ens = attempt_load([f, f])
    assert len(ens) == 2 and ens.names == m.names


@pytest.mark.skipif(not os.path.isdir('/root/reference/models'), reason='needs the reference checkout (build container only)')
def test_reference_written_checkpoint_loads(tmp_path):
    """A checkpoint pickled by the reference's own classes, unpickled against this package's classes of the same names."""
    import subprocess
    f = str(tmp_path / 'ref.pt')
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = '''
import sys, contextlib, io
sys.path.insert(0, %r)
from oracle.gen_golden import import_reference, tiny_cfg
from oracle import portable_init
import torch
from copy import deepcopy
Model = import_reference()[0]
with contextlib.redirect_stdout(io.StringIO()):
    m = Model(deepcopy(tiny_cfg('fourier')))
m.load_state_dict(portable_init.fill_(m.state_dict()))
m.names = ['person', 'car', 'bus', 'lamp', 'motorcycle', 'truck']
torch.save({'model': deepcopy(m).half(), 'ema': None}, %r)
''' % (repo, f)
    subprocess.run([sys.executable, '-c', code], check=True, capture_output=True, timeout=600)
    from models.experimental import attempt_load
    with contextlib.redirect_stdout(io.StringIO()):
        loaded = attempt_load(f, fuse=False)
    assert type(loaded).__module__ == 'models.yolo_test' and loaded.names[1] == 'car'
    from oracle import portable_init
    want = portable_init.fill_({k: v.clone() for k, v in loaded.state_dict().items()})
    for k, v in loaded.state_dict().items():
        if v.dtype.is_floating_point and 'anchor' not in k:
            assert torch.equal(v, want[k].half().float()), k
