"""Condense a rocprofv3 --kernel-trace CSV: per (kernel, grid) launch count, total and mean duration, plus the union of
busy intervals (kernels of the two backbone lanes overlap, so the plain sum over-counts).

usage: python tools/agg_trace.py <..._kernel_trace.csv> [top] [tail_ms]
(tail_ms: only the kernels that start in the last tail_ms of the trace, e.g. the graph-replayed timed steps)

Also prints where the device is idle or nearly so: idle time attributed to the kernel that ends the gap (what the
device was waiting to start), and "thin" time -- intervals in which all running kernels together have fewer than
256 workgroups (one per CU) -- attributed to the kernels running then."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([\w:]+(<[^(]*>)?)', name)
    s = m.group(1) if m else name
    return s[:70]


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    rows = list(csv.DictReader(open(path)))
    if len(sys.argv) > 3:
        end = max(int(r['End_Timestamp']) for r in rows)
        cut = end - float(sys.argv[3]) * 1e6
        rows = [r for r in rows if int(r['Start_Timestamp']) >= cut]
    agg = defaultdict(lambda: [0, 0])
    byname = defaultdict(lambda: [0, 0])
    iv = []
    for r in rows:
        t0, t1 = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        k = short(r['Kernel_Name'])
        grid = '%sx%sx%s' % (r.get('Grid_Size_X', '?'), r.get('Grid_Size_Y', '?'), r.get('Grid_Size_Z', '?'))
        a = agg[(k, grid)]
        a[0] += 1
        a[1] += t1 - t0
        b = byname[k]
        b[0] += 1
        b[1] += t1 - t0
        iv.append((t0, t1))
    iv.sort()
    busy, cur0, cur1 = 0, None, None
    for a, b in iv:
        if cur1 is None or a > cur1:
            if cur1 is not None:
                busy += cur1 - cur0
            cur0, cur1 = a, b
        else:
            cur1 = max(cur1, b)
    if cur1 is not None:
        busy += cur1 - cur0
    total = sum(v[1] for v in byname.values())
    span = iv[-1][1] - iv[0][0] if iv else 0
    print('launches %d   sum %.2f ms   union-busy %.2f ms   span %.2f ms' % (len(rows), total / 1e6, busy / 1e6, span / 1e6))
    print('\n== by kernel ==')
    for k, (n, t) in sorted(byname.items(), key=lambda kv: -kv[1][1])[:top]:
        print('%-72s x%-6d %9.2f ms  %5.1f%%  %8.1f us' % (k, n, t / 1e6, 100.0 * t / total, t / n / 1e3))
    print('\n== by kernel and grid ==')
    for (k, g), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print('%-60s %-16s x%-6d %9.2f ms  %8.1f us' % (k[:60], g, n, t / 1e6, t / n / 1e3))
    gaps(rows, min(top, 30))


def gaps(rows, top):
    ev = []
    for i, r in enumerate(rows):
        t0, t1 = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        wg = 1
        for ax in 'XYZ':
            g, w = int(r.get('Grid_Size_' + ax, 1) or 1), int(r.get('Workgroup_Size_' + ax, 1) or 1)
            wg *= max(1, g // max(1, w))
        ev.append((t0, 1, i, wg))
        ev.append((t1, 0, i, wg))
    ev.sort()
    running = {}
    idle_by, thin_by = defaultdict(lambda: [0, 0]), defaultdict(lambda: [0, 0])
    idle = thin = 0
    last = None
    for t, kind, i, wg in ev:
        if last is not None and t > last:
            dt = t - last
            if not running:
                if kind == 1 and dt < 2e6:  # gaps above 2 ms are host stalls (step boundaries), not launch gaps
                    idle += dt
                    a = idle_by[short(rows[i]['Kernel_Name'])]
                    a[0] += 1
                    a[1] += dt
            elif sum(running.values()) < 256:
                thin += dt
                for j in running:
                    a = thin_by[short(rows[j]['Kernel_Name'])]
                    a[0] += 1
                    a[1] += dt
        if kind == 1:
            running[i] = wg
        else:
            running.pop(i, None)
        last = t
    print('\n== idle (no kernel running; gaps < 2 ms) %.2f ms, by the kernel that ends the gap ==' % (idle / 1e6))
    for k, (n, t) in sorted(idle_by.items(), key=lambda kv: -kv[1][1])[:top]:
        print('%-72s x%-6d %9.2f ms  %8.1f us' % (k, n, t / 1e6, t / n / 1e3))
    print('\n== thin (< 256 workgroups running in total) %.2f ms, by running kernel ==' % (thin / 1e6))
    for k, (n, t) in sorted(thin_by.items(), key=lambda kv: -kv[1][1])[:top]:
        print('%-72s x%-6d %9.2f ms  %8.1f us' % (k, n, t / 1e6, t / n / 1e3))


if __name__ == '__main__':
    main()
