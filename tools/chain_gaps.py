"""Forward and backward of one training step as the GPU saw them (rocprofv3 --kernel-trace CSV, the step between the last two
sgd_ema_kernel launches, steps_back earlier): wall time, kernel-busy time (union over streams), launch count, and the total of the
gaps in which NO kernel runs, split at the first loss kernel.  The gaps are what dependent launches cost beyond their kernels.

usage: python tools/chain_gaps.py <kernel_trace.csv> [steps_back]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 0
opt = sorted(int(r['End_Timestamp']) for r in rows if 'sgd_ema_kernel' in r['Kernel_Name'])
t0, t1 = opt[-2 - back], opt[-1 - back]
step = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows
               if t0 <= int(r['Start_Timestamp']) < t1), key=lambda e: e[0])
tl = next(a for a, b, k in step if 'loss_records' in k or 'build_targets' in k)


def seg(name, lo, hi):
    ev = [(max(a, lo), min(b, hi), k) for a, b, k in step if b > lo and a < hi]
    busy, gaps, cur_end, ngap, big = 0, 0, lo, 0, []
    for a, b, k in ev:
        if a > cur_end:
            gaps += a - cur_end
            ngap += 1
            big.append((a - cur_end, k))
            cur_end = a
        if b > cur_end:
            busy += b - cur_end
            cur_end = b
    if hi > cur_end:
        gaps += hi - cur_end
    big.sort(reverse=True)
    print('%-9s wall %7.2f ms  kernels busy %7.2f ms  %5d launches  idle %6.2f ms in %d gaps (mean %.1f us); largest: %s'
          % (name, (hi - lo) / 1e6, busy / 1e6, len(ev), gaps / 1e6, ngap, gaps / max(ngap, 1) / 1e3,
             ', '.join('%.0f us before %s' % (g / 1e3, k.split('(')[0][-40:]) for g, k in big[:4])))


seg('forward', t0, tl)
seg('backward', tl, t1)
