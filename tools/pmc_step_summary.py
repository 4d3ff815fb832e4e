"""Per-step HBM traffic by kernel family from two `rocprofv3 --pmc` passes over bench.py (tools/pmc_step.sh).
FETCH_SIZE / WRITE_SIZE come in KiB per dispatch; FETCH_SIZE is doubled, as /opt/skills/guides/MI355X_MICROARCH.md (HBM
section) prescribes for gfx950 wide streaming reads (128-byte requests are tallied at 64 bytes).  Output: one JSON object,
MB per training step (dispatches of all steps / number of optimizer launches)."""
import csv
import glob
import json
import re
import sys

csv.field_size_limit(1 << 30)

FAMILIES = [('igemm', r'igemm_kernel|wgrad_kernel|slab_reduce'), ('bn', r'bn_act_fwd|bn_bwd|bn_finalize|stat_fold|pair_fold|pair_finalize'),
            ('cem', r'smallconv|sobel|chansum|cem_'), ('tokens', r'layernorm|attn_|gelu|dropout|sigmoid|mul_kernel|scale_kernel'),
            ('fusion', r'avgpool8|upsample_add|fusion_stats|ffm_|separation'), ('spp', r'spp_'), ('optimizer', r'sgd_ema'),
            ('aten', r'at::native|rocclr'), ('other', r'.')]


def family(name):
    for f, pat in FAMILIES:
        if re.search(pat, name):
            return f
    return 'other'


def load(d, counter):
    per = {}
    steps = 0
    files = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    for fn in files:
        for r in csv.DictReader(open(fn)):
            if r['Counter_Name'] != counter:
                continue
            n = r['Kernel_Name']
            if 'sgd_ema' in n:
                steps += 1
            f = family(n)
            per[f] = per.get(f, 0.0) + float(r['Counter_Value'])
    return per, steps


def main():
    rd, s1 = load(sys.argv[1], 'FETCH_SIZE')
    wr, s2 = load(sys.argv[2], 'WRITE_SIZE')
    steps = max(s1, 1)
    out = {'workload': sys.argv[3] if len(sys.argv) > 3 else 'l_fourier', 'steps_profiled': s1,
           'method': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --mode eager; KiB per dispatch summed per '
                     'family and step; FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md)',
           'read_MB_per_step': {k: round(2 * v * 1024 / 1e6 / steps, 1) for k, v in sorted(rd.items())},
           'write_MB_per_step': {k: round(v * 1024 / 1e6 / max(s2, 1), 1) for k, v in sorted(wr.items())}}
    out['total_MB_per_step'] = round(sum(out['read_MB_per_step'].values()) + sum(out['write_MB_per_step'].values()), 1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
