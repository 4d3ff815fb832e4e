import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'mmi-det_amd')
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, 'tests', 'golden')
CFG_DIR = os.path.join(PKG, 'models', 'transformer')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def tiny_cfg(kind):
    """Same tiny graphs as oracle/gen_golden.py::tiny_cfg, from the YAMLs shipped with the package."""
    import yaml
    if kind == 'fourier':
        with open(os.path.join(CFG_DIR, 'yolov5l_fusion_transformer_M3FD_fuse3_fourier.yaml')) as f:
            d = yaml.safe_load(f)
        d['depth_multiple'], d['width_multiple'] = 0.33, 0.25
        d['backbone'][6][3] = [32]
    else:
        with open(os.path.join(CFG_DIR, 'yolov5s_fusion_add_vedai.yaml')) as f:
            d = yaml.safe_load(f)
        d['width_multiple'] = 0.25
    return d


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
