import os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mmi-det_amd'), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests')]
import torch
import torch.distributed as dist
from test_step_gpu import make, batch
from test_ops_gpu import dev
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev())
from mmidet_hip.ddp import GradReducer
res = {}
for name, ddp, graph in [('plainA', 0, 1), ('plainB', 0, 1), ('ddpA', 1, 1), ('ddpB', 1, 1), ('eagerA', 0, 0), ('ddp_eager', 1, 0)]:
    m, ts, cfg = make(graph=bool(graph))
    if ddp:
        ts.reducer = GradReducer(list(m.parameters()))
    ls = []
    if not graph:
        for _ in range(2):
            ts.step(*batch(cfg, 30))
    for i in range(3):
        l, it = ts.step(*batch(cfg, 30 + i))
        ls.append(float(l))
    res[name] = ls
    print(name, ['%.6f' % v for v in ls], flush=True)
dist.destroy_process_group()
