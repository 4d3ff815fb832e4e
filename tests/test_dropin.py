"""The drop-in itself (BASELINE.json north_star "drops into the existing train.py and test.py"), CPU, build
container only (needs the reference checkout; nothing of it travels): the reference's OWN train.py / test.py /
detect_twostream.py are imported and launched on the overlay packages (mmidet_hip/overlay.py) and must end up bound to the
native hot path, while every module outside the path still comes from the reference.  The packages the reference imports
but this image lacks (cv2, torchvision, seaborn, thop, torchsummary, tensorboard, wandb, pycocotools) are
inert stubs seeded by a sitecustomize.py the test writes."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'mmi-det_amd')
REF = '/root/reference'
needs_ref = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, 'train.py')),
                               reason='needs the reference checkout (build container only)')

STUBS = '''
import importlib.machinery, sys
from unittest import mock
sys.dont_write_bytecode = True
for name in ('cv2', 'torchvision', 'torchvision.ops', 'torchvision.models', 'seaborn', 'thop', 'torchsummary',
             'torch.utils.tensorboard', 'wandb', 'pycocotools', 'pycocotools.coco', 'pycocotools.cocoeval'):
    m = mock.MagicMock()
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    sys.modules[name] = m
'''

PROBE = '''
import json, os, sys
from mmidet_hip import overlay
overlay.activate(%r)
import train, test, detect_twostream            # the reference's scripts, as modules
import models.common, models.experimental, models.yolo, models.yolo_test
import utils.datasets, utils.general, utils.loss, utils.plots, utils.metrics, utils.torch_utils, utils.autoanchor
def src(obj):
    return os.path.realpath(sys.modules[obj.__module__].__file__) if hasattr(obj, '__module__') else os.path.realpath(obj.__file__)
out = {
    'train.Model': src(train.Model), 'train.ComputeLoss': src(train.ComputeLoss), 'train.ModelEMA': src(train.ModelEMA),
    'train.attempt_load': src(train.attempt_load), 'test.attempt_load': src(test.attempt_load),
    'detect.attempt_load': src(detect_twostream.attempt_load),
    'test.non_max_suppression': src(test.non_max_suppression), 'detect.non_max_suppression': src(detect_twostream.non_max_suppression),
    'models.yolo.Model is models.yolo_test.Model': models.yolo.Model is models.yolo_test.Model,
    'models.common.Conv': src(models.common.Conv),
    # outside the hot path: still the reference's
    'utils.datasets': src(utils.datasets), 'utils.plots': src(utils.plots), 'utils.metrics': src(utils.metrics),
    'train.create_dataloader_rgb_ir': src(train.create_dataloader_rgb_ir),
    'train.labels_to_class_weights': os.path.realpath(train.labels_to_class_weights.__code__.co_filename),
    'train.select_device': os.path.realpath(train.select_device.__code__.co_filename),
    'train.check_anchors': os.path.realpath(train.check_anchors.__code__.co_filename),
    'test.scale_coords': os.path.realpath(test.scale_coords.__code__.co_filename),
    'models.common.autoShape': os.path.realpath(sys.modules[models.common.autoShape.__module__].__file__),
    'test module': os.path.realpath(test.__file__),
}
print('RESULT ' + json.dumps(out))
'''


def _env(tmp_path):
    stubs = tmp_path / 'stubs'
    stubs.mkdir()
    (stubs / 'sitecustomize.py').write_text(STUBS)
    env = dict(os.environ)
    env['PYTHONPATH'] = os.pathsep.join([str(stubs), PKG])
    env.pop('MMIDET_REFERENCE_ROOT', None)
    return env


@needs_ref
def test_reference_scripts_bind_to_the_native_hot_path(tmp_path):
    r = subprocess.run([sys.executable, '-W', 'ignore', '-c', PROBE % REF], env=_env(tmp_path), cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('RESULT ')][-1][7:])
    native = ['train.Model', 'train.ComputeLoss', 'train.ModelEMA', 'train.attempt_load', 'test.attempt_load',
              'detect.attempt_load', 'test.non_max_suppression', 'detect.non_max_suppression', 'models.common.Conv']
    for k in native:
        assert out[k].startswith(os.path.realpath(PKG) + os.sep), (k, out[k])
    assert out['models.yolo.Model is models.yolo_test.Model'] is True
    theirs = ['utils.datasets', 'utils.plots', 'utils.metrics', 'train.create_dataloader_rgb_ir',
              'train.labels_to_class_weights', 'train.select_device', 'train.check_anchors', 'test.scale_coords',
              'models.common.autoShape', 'test module']
    for k in theirs:
        assert out[k].startswith(os.path.realpath(REF) + os.sep), (k, out[k])


@needs_ref
@pytest.mark.parametrize('script', ['train.py', 'test.py', 'detect_twostream.py'])
def test_launcher_runs_the_reference_scripts(tmp_path, script):
    """INTEGRATION.md §1, verbatim: `PYTHONPATH=<repo>/mmi-det_amd python -m mmidet_hip.overlay <script> --help` from the
    reference checkout.  --help makes argparse exit right after the script's whole import block has run."""
    r = subprocess.run([sys.executable, '-W', 'ignore', '-m', 'mmidet_hip.overlay', script, '--help'], env=_env(tmp_path),
                       cwd=REF, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'usage:' in r.stdout and '--weights' in r.stdout


def test_overlay_is_inert_without_a_reference(tmp_path):
    """The GPU box: no checkout anywhere.  The native modules import and unknown names fail the ordinary way."""
    code = ('import sys; sys.path.insert(0, %r)\n'
            'import utils.general, models.common\n'
            'from mmidet_hip import overlay\n'
            'assert overlay.reference_root() is None\n'
            'try:\n    utils.general.labels_to_class_weights\n    raise SystemExit(3)\nexcept AttributeError:\n    pass\n'
            'try:\n    import utils.datasets\n    raise SystemExit(4)\nexcept ImportError:\n    pass\n' % PKG)
    env = dict(os.environ)
    env.pop('MMIDET_REFERENCE_ROOT', None)
    env.pop('PYTHONPATH', None)
    r = subprocess.run([sys.executable, '-c', code], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]


def test_model_half_is_io_only():
    """train.py:680 `model.half().float()` rounds the weights through fp16; test.py:66-68 runs inference 'in fp16'.  The
    native model keeps fp32 storage (the kernels are fp32) with the same values."""
    import torch
    sys.path.insert(0, PKG)
    from conftest import tiny_cfg
    from models.yolo_test import Model
    m = Model(tiny_cfg('add'))
    w0 = m.model[0].conv.conv.weight.detach().clone()
    m.half()
    w1 = m.model[0].conv.conv.weight
    assert w1.dtype == torch.float32 and torch.equal(w1, w0.half().float()) and m._io_half
    assert w1.is_contiguous(memory_format=torch.channels_last)
    m.float()
    assert not m._io_half and torch.equal(m.model[0].conv.conv.weight, w0.half().float())
