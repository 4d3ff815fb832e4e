"""The token chains of one training step as the GPU ran them (rocprofv3 --kernel-trace CSV): for the forward, every kernel between
an avgpool8_fwd (pool to tokens) and the upsample_add_fwd that ends the same fusion block, with start offset, duration and gap to
the previous kernel's end; totals per kernel name for forward chains and for the backward's mirror (upsample_add_bwd .. avgpool8_bwd).

usage: python tools/token_chain.py <kernel_trace.csv> [detail_block_index]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
detail = int(sys.argv[2]) if len(sys.argv) > 2 else 0
opt = sorted(int(r['End_Timestamp']) for r in rows if 'sgd_ema_kernel' in r['Kernel_Name'])
t0, t1 = opt[-2], opt[-1]
step = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][-60:])
               for r in rows if t0 <= int(r['Start_Timestamp']) < t1), key=lambda e: e[0])


def chains(first, last):
    out, cur = [], None
    for a, b, k in step:
        if cur is None and first in k:
            cur = []
        if cur is not None:
            cur.append((a, b, k))
            if last in k:
                out.append(cur)
                cur = None
    return out


for name, first, last in (('forward', 'avgpool8_fwd', 'upsample_add_fwd'), ('backward', 'upsample_add_bwd', 'avgpool8_bwd')):
    cs = chains(first, last)
    tot = collections.Counter()
    cnt = collections.Counter()
    wall = gaps = 0
    for c in cs:
        wall += c[-1][1] - c[0][0]
        end = c[0][0]
        for a, b, k in c:
            tot[k] += b - a
            cnt[k] += 1
            if a > end:
                gaps += a - end
            end = max(end, b)
    print('== %s: %d chains, wall %.2f ms, idle inside %.2f ms, %d launches' % (name, len(cs), wall / 1e6, gaps / 1e6, sum(cnt.values())))
    for k, v in tot.most_common(24):
        print('   %-62s %4d x  %8.3f ms' % (k, cnt[k], v / 1e6))
    if cs and detail < len(cs):
        c = cs[detail]
        print('-- chain %d in order (offset us, duration us, gap us)' % detail)
        end = c[0][0]
        for a, b, k in c:
            print('   %9.1f %8.1f %7.1f  %s' % ((a - c[0][0]) / 1e3, (b - a) / 1e3, (a - end) / 1e3, k))
            end = max(end, b)
