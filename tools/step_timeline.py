"""Per-stream timeline of one training step from a rocprofv3 --kernel-trace CSV: the step between the last two
sgd_ema_kernel launches; per HIP stream its first start, last end, busy time and launch count, and a coarse (5 ms bins)
occupancy chart, to see which stream carries the critical path and where the device waits.

usage: python tools/step_timeline.py <kernel_trace.csv> [bin_ms] [steps_back]   (steps_back=1: the step before the last one)"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
binw = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 5e6
opt = [int(r['End_Timestamp']) for r in rows if 'sgd_ema_kernel' in r['Kernel_Name']]
assert len(opt) >= 2, 'need two optimizer steps in the trace'
back = int(sys.argv[3]) if len(sys.argv) > 3 else 0
t0, t1 = opt[-2 - back], opt[-1 - back]
step = [r for r in rows if t0 <= int(r['Start_Timestamp']) < t1]
print('step %.2f ms, %d launches' % ((t1 - t0) / 1e6, len(step)))
by = defaultdict(list)
for r in step:
    by[(r['Queue_Id'], r['Stream_Id'])].append((int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0, r['Kernel_Name']))
nb = int((t1 - t0) / binw) + 1
print('%-10s %6s %9s %9s %9s   busy per %.0f ms bin (%%)' % ('q/stream', 'n', 'first ms', 'last ms', 'busy ms', binw / 1e6))
for key, iv in sorted(by.items(), key=lambda kv: kv[1][0][0]):
    iv.sort()
    busy = sum(b - a for a, b, _ in iv)
    bins = [0.0] * nb
    for a, b, _ in iv:
        i = int(a / binw)
        while a < b and i < nb:
            e = min(b, (i + 1) * binw)
            bins[i] += e - a
            a = e
            i += 1
    chart = ' '.join('%3d' % min(999, round(100 * x / binw)) for x in bins)
    print('%-10s %6d %9.2f %9.2f %9.2f   %s' % ('%s/%s' % key, len(iv), iv[0][0] / 1e6, max(b for _, b, _ in iv) / 1e6, busy / 1e6, chart))
last = sorted(step, key=lambda r: int(r['End_Timestamp']))[-12:]
print('\nlast kernels of the step:')
for r in last:
    print('  %8.2f..%8.2f ms  q%s/s%s  %s' % ((int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - t0) / 1e6,
                                            r['Queue_Id'], r['Stream_Id'], r['Kernel_Name'][:80]))
