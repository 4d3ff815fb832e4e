"""How long does dist.barrier() take on an idle and on a busy stream (world size 1, RCCL)?"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29577')
dev = torch.device('cuda:0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
x = torch.randn(8192, 8192, device=dev)
for i in range(5):
    torch.cuda.synchronize()
    t = time.perf_counter()
    dist.barrier()
    torch.cuda.synchronize()
    print('idle barrier %d: %.3f ms' % (i, (time.perf_counter() - t) * 1e3), flush=True)
for i in range(3):
    for _ in range(20):
        y = x @ x
    t = time.perf_counter()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(20):
        y = x @ x
    t2 = time.perf_counter()
    dist.barrier()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print('20 matmuls: sync-only %.2f ms; with barrier %.2f ms' % ((t1 - t) * 1e3, (t3 - t2) * 1e3), flush=True)
dist.destroy_process_group()
