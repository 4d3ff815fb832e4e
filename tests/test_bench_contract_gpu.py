"""bench.py's output contract (one JSON line on stdout; metric/value/unit/n_gpus/steps/warmup/ms_per_step/... plus the
`roofline` object) on a small workload, run as the driver runs it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--workload', 's_add', '--steps', '3', '--warmup', '2',
                        '--no-cpu-baseline'], capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, 'stdout must hold exactly the JSON line: %r' % lines[:3]
    j = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert k in j, k
    assert j['n_gpus'] == 1 and j['steps'] == 3 and j['warmup'] == 2 and j['scaling'] == 'weak' and j['vs_baseline'] is None
    assert j['higher_is_better'] is True and j['dtype'] == 'f32' and j['data'] == 'synthetic' and 'workload' in j['config']
    assert abs(j['value'] - j['config']['global_batch'] / (j['ms_per_step'] * 1e-3)) / j['value'] < 1e-3
    rf = j['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in rf, k
    assert rf['bound'] == 'mfma' and rf['unit'] == 'TFLOP/s' and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-3
