"""Condense the four `rocprofv3 --pmc` passes of tools/run_pmc.sh into one table per kernel (mean per launch).
FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950 wide streaming reads;
FETCH_SIZE/WRITE_SIZE are reported by rocprofv3 in KiB.

usage: python tools/pmc_summary.py gpurun_out/pmc_<tag>"""
import csv
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    m = re.match(r'(?:void )?([\w:]+(<[^(]*>)?)', name)
    return (m.group(1) if m else name)[:60]


def main():
    base = sys.argv[1]
    vals = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for i in range(1, 6):
        try:
            f = open('%s_%d/p_counter_collection.csv' % (base, i))
        except OSError:
            continue
        for r in csv.DictReader(f):
            k = short(r['Kernel_Name'])
            if not any(s in k for s in ('igemm_kernel', 'wgrad_kernel', 'slab_reduce')):
                continue
            vals[k][r['Counter_Name']].append(float(r['Counter_Value']))
            if i == 1 and r['Counter_Name'] == 'SQ_WAVE_CYCLES':
                dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k in sorted(vals):
        print('== %s' % k)
        if dur[k]:
            print('   %-28s %14.1f us (under counter collection)' % ('duration', sum(dur[k]) / len(dur[k])))
        for c in sorted(vals[k]):
            v = sum(vals[k][c]) / len(vals[k][c])
            if c == 'FETCH_SIZE':
                print('   %-28s %14.1f MB  (= 2 x %.1f MB reported: gfx950 correction)' % ('HBM/fabric read', 2 * v * 1024 / 1e6, v * 1024 / 1e6))
            elif c == 'WRITE_SIZE':
                print('   %-28s %14.1f MB' % ('HBM/fabric write', v * 1024 / 1e6))
            else:
                print('   %-28s %14.0f' % (c, v))
        c = vals[k]
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'SQ_BUSY_CYCLES' in c:
            mf = sum(c['SQ_VALU_MFMA_BUSY_CYCLES']) / len(c['SQ_VALU_MFMA_BUSY_CYCLES'])
            wc = sum(c['SQ_WAVE_CYCLES']) / len(c['SQ_WAVE_CYCLES'])
            print('   %-28s %14.3f  (MFMA busy cycles / (4 x wave quad-cycles): share of its life a wave keeps the pipe busy)'
                  % ('mfma_busy / wave_cycles', mf / (4 * wc)))


if __name__ == '__main__':
    main()
