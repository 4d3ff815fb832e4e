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
    # Poison mode for the GPU suite (mmidet_hip/alloc.py): every output / workspace is born as 0xFF bytes (NaN / -1) between
    # guard zones, so an element no kernel stored, or a store outside an output, fails the test that caused it instead of
    # hiding behind whatever a recycled block held.  Must be set before mmidet_hip is imported.  MMIDET_POISON=0 switches off.
    expr = config.getoption('markexpr', '') or ''
    if 'gpu' in expr and 'not gpu' not in expr:
        os.environ.setdefault('MMIDET_POISON', '1')


@pytest.fixture(autouse=True)
def _poison_checks(request):
    """After every GPU test: the guard zones of everything allocated during the test are intact (no out-of-range store) and
    every workspace's arrival counters are back at zero (no last-arriver election left half done)."""
    yield
    if request.node.get_closest_marker('gpu') is None or 'mmidet_hip.alloc' not in sys.modules:
        return
    from mmidet_hip import alloc, ops
    if alloc.POISON:
        alloc.check_guards()
        ops.check_counters()


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


def own_process(fn):
    """Run an argument-less GPU test in a pytest child process of its own.  For the tests that capture a hipGraph while a
    torch.distributed (NCCL) process group's watchdog thread is alive: one of round 4's full runs died with SIGABRT inside such a
    capture, and an abort in the main session would take every later test with it.  In a child it is one
    failed test with the child's output attached."""
    import functools
    import subprocess

    @functools.wraps(fn)
    def wrapper():
        if os.environ.get('MMIDET_TEST_CHILD') == '1':
            return fn()
        node = '%s::%s' % (os.path.abspath(sys.modules[fn.__module__].__file__), fn.__name__)
        env = dict(os.environ, MMIDET_TEST_CHILD='1')
        for k in ('MASTER_PORT', 'MASTER_ADDR', 'RANK', 'WORLD_SIZE', 'LOCAL_RANK'):      # (rendezvous settings an earlier test of this session left:
            env.pop(k, None)                                                              #  its store may still hold the port)
        r = subprocess.run([sys.executable, '-m', 'pytest', node, '-q', '-m', 'gpu', '-x', '-p', 'no:cacheprovider'],
                           env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, 'child pytest of %s failed (exit code %d):\n%s' % (fn.__name__, r.returncode, (r.stdout + r.stderr)[-4000:])
    return wrapper
