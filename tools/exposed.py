"""Where a training step's time goes when no GEMM is on the chip: from a rocprofv3 --kernel-trace CSV, the last full step (between
the last two sgd_ema_kernel launches) is cut into intervals by what is running -- a GEMM-family kernel (igemm / wgrad), only other
kernels (attributed in equal shares to the kernel names running), or nothing (idle, attributed to the kernel that ends the gap).

usage: python tools/exposed.py <kernel_trace.csv> [steps_back]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([\w:]+)', name)
    return (m.group(1) if m else name)[:48]


rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 0
opt = sorted(int(r['End_Timestamp']) for r in rows if 'sgd_ema_kernel' in r['Kernel_Name'])
t0, t1 = opt[-2 - back], opt[-1 - back]
ev = []
for r in rows:
    a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if b <= t0 or a >= t1:
        continue
    k = short(r['Kernel_Name'])
    gemm = 'igemm_kernel' in k or 'wgrad_kernel' in k
    ev.append((max(a, t0), 1, k, gemm))
    ev.append((min(b, t1), -1, k, gemm))
ev.sort(key=lambda e: (e[0], e[1]))
running = defaultdict(int)
ngemm = 0
prev = t0
gemm_t = idle_t = 0
other = defaultdict(float)
idle_by = defaultdict(float)
for t, d, k, g in ev:
    dt = t - prev
    if dt > 0:
        names = [n for n, c in running.items() if c > 0]
        if ngemm > 0:
            gemm_t += dt
        elif names:
            for n in names:
                other[n] += dt / len(names)
        else:
            idle_t += dt
            if d > 0:
                idle_by[k] += dt
    prev = t
    running[k] += d
    if g:
        ngemm += d
step = (t1 - t0) / 1e6
print('step %.2f ms: a GEMM running %.2f ms, only other kernels %.2f ms, idle %.2f ms' % (step, gemm_t / 1e6, sum(other.values()) / 1e6, idle_t / 1e6))
print('-- no GEMM on the chip, by the kernels running (ms)')
for n, v in sorted(other.items(), key=lambda kv: -kv[1])[:25]:
    print('   %-48s %7.3f' % (n, v / 1e6))
print('-- idle, by the kernel that ended the gap (ms)')
for n, v in sorted(idle_by.items(), key=lambda kv: -kv[1])[:15]:
    print('   %-48s %7.3f' % (n, v / 1e6))
