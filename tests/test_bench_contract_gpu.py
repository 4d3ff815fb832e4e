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


def test_long_unsynchronised_run_stays_finite():
    """80 eager steps of the BASELINE workload with no host synchronisation in between: the host enqueues a step in half the
    time the GPU runs it and gets many steps ahead.  Regression test for the optimizer's gradient-pointer table, which used to
    travel through ONE pinned buffer that the host rewrote before the GPU had read the previous step's copy (training then
    diverged to NaN within ~100 steps; mmidet_hip/optim.py::_Staging)."""
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--steps', '80', '--warmup', '2', '--mode', 'eager',
                        '--no-cpu-baseline', '--no-roofline', '--no-split-probe'], capture_output=True, text=True, timeout=900,
                       cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    loss = j['config']['loss']
    assert all(v == v and abs(v) < 1e3 for v in loss), loss           # finite (bench asserts it too) and sane
    assert loss[3] < 0.25, loss                                       # the detection loss has come down from ~0.32 / image


def test_self_spawned_rank_with_the_native_transport_world1():
    """`python bench.py --gpus N` starts its own ranks (torch.distributed.run, before anything touches the GPU) and relays rank
    0's JSON line: the launcher path of the N > 1 runs, rehearsed at world size 1 with the library's own RCCL communicator
    (MMIDET_COMM=native) and the data-parallel code path (--ddp)."""
    env = dict(os.environ, MMIDET_COMM='native')
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '1', '--spawn', '--ddp', '--workload', 's_add',
                        '--steps', '3', '--warmup', '2', '--no-cpu-baseline', '--no-roofline', '--no-split-probe'],
                       capture_output=True, text=True, timeout=600, cwd=REPO, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, r.stdout[-500:]
    j = json.loads(lines[0])
    assert j['n_gpus'] == 1 and 'mmi_allreduce_bucket' in (j['config']['gradient_transport'] or '')
    assert j['value'] > 0 and all(v == v for v in j['config']['loss'])


@pytest.mark.parametrize('mode,ranks', [('eager', 2), ('auto', 2), ('auto', 4)])
def test_two_ranks_rehearsal_on_one_gpu(mode, ranks):
    """`python bench.py --gpus 2` end to end on a one-GPU box: the self-spawn launcher, two ranks (both on GPU 0, collectives over
    gloo: MMIDET_COMM=gloo, a rehearsal transport), broadcast of the initial weights, warm-up, the launch-mode probe with its
    cross-rank decision, the barrier-bracketed timed region with the max over ranks, the per-rank host-time gather and rank 0's
    single JSON line with the whole-job aggregate -- every line of bench.py the driver's N > 1 scaling run executes except RCCL.
    Four ranks is as far as a one-GPU box goes (at most 6 processes may hold the card, this one included); each rank pins itself
    to its own block of cores before touching the GPU and the line reports the ranks seen and their core sets."""
    env = dict(os.environ, MMIDET_COMM='gloo', MMIDET_POISON='0')
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', str(ranks), '--workload', 's_add', '--steps', '3', '--warmup', '2',
                        '--mode', mode, '--no-cpu-baseline', '--no-split-probe'] + (['--no-roofline'] if ranks > 2 else []),
                       capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, 'rank 0 alone prints the JSON line: %r' % r.stdout[-500:]
    j = json.loads(lines[0])
    c = j['config']
    assert j['n_gpus'] == ranks and j['scaling'] == 'weak' and c['parallelism'] == 'dp%d' % ranks and c['global_batch'] == ranks * c['batch_per_gpu']
    assert abs(j['value'] - c['global_batch'] / (j['ms_per_step'] * 1e-3)) / j['value'] < 1e-3      # whole-job aggregate
    assert len(c['host_enqueue_ms_per_step_per_rank']) == ranks and 'REHEARSAL' in c['gradient_transport']
    assert c['ranks_seen'] == ranks and len(c['cpu_affinity_per_rank']) == ranks
    sets = [v for v in c['cpu_affinity_per_rank'] if v is not None]
    assert len(set(sets)) == len(sets), 'ranks share a core set: %r' % (c['cpu_affinity_per_rank'],)   # (None: fewer cores than ranks)
    assert all(v == v and abs(v) < 1e3 for v in c['loss']), c['loss']
    assert ('roofline' in j) == (ranks <= 2) and 'cpu_baseline' not in j
