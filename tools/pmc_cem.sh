#!/bin/bash
# usage: tools/pmc_cem.sh <tag>   (run on the GPU box; writes gpurun_out/pmccem_<tag>_*/ and prints a per-kernel table)
set -u
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
            "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmccem_${tag}_$i -o p -- python3 $R/tools/pmc_cem.py > $R/gpurun_out/pmccem_${tag}_$i.log 2>&1 || { echo "pass $i failed"; tail -n 5 $R/gpurun_out/pmccem_${tag}_$i.log; }
done
python3 - <<PY
import csv, re, glob
from collections import defaultdict
vals = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
csv.field_size_limit(1 << 30)
for i in range(1, 6):
    for f in glob.glob('$R/gpurun_out/pmccem_${tag}_%d/*counter_collection.csv' % i):
        for r in csv.DictReader(open(f)):
            k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); k = re.sub(r'^void ', '', k); k = re.match(r'[\w:]+(<[^(]*>)?', k).group(0)[:50]
            vals[k][r['Counter_Name']].append(float(r['Counter_Value']))
            if i == 1 and r['Counter_Name'] == 'SQ_WAVE_CYCLES': dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k in sorted(vals):
    print('==', k, ' %.1f us' % (sum(dur[k]) / max(len(dur[k]), 1)))
    for c in sorted(vals[k]):
        v = sum(vals[k][c]) / len(vals[k][c])
        print('   %-28s %16.0f' % (c, v))
PY
