"""Shader clock while the 3x3 conv kernel runs back to back (rocm-smi sampled from a side thread) vs idle."""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mmi-det_amd'))
from mmidet_hip import lib, ops  # noqa: E402


def sample(tag):
    out = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True).stdout
    keep = [l.strip() for l in out.splitlines() if 'sclk' in l or 'Power' in l or 'mclk' in l or 'fclk' in l]
    print(tag, ' | '.join(keep)[:400], flush=True)


d = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream
C, H, W = 256, 1536, 128
x = torch.randn(1, H, W, C, device=d)
w = torch.randn(C, 3, 3, C, device=d) * 0.05
desc = ops._desc((1, H, W, C), C, 3, 1, C, C)
y = torch.empty(1, H, W, C, device=d)
part = torch.empty((lib.conv_fwd_row_blocks(desc) + 64) * 2 * C, device=d)
sample('idle   ')
stop = False


def sampler():
    time.sleep(1.0)
    for i in range(3):
        sample('loaded ')
        time.sleep(0.7)


th = threading.Thread(target=sampler)
th.start()
t0 = time.time()
n = 0
while time.time() - t0 < 4.0:
    for _ in range(50):
        ops.conv_fwd(x, w, None, y, part, desc, st)
    torch.cuda.synchronize()
    n += 50
dt = time.time() - t0
th.join()
print('%.1f TFLOP/s over %.1f s' % (n * 2.0 * H * W * C * C * 9 / dt / 1e12, dt))
