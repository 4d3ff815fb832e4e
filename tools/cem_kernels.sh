#!/bin/bash
# Per-kernel durations of the CEM module at the step's size (rocprofv3 kernel trace of tools/bench_cem.py, fused rows only)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4/cemprof_${1:-x}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH_CEM_FUSED_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o cem -- python3 $R/tools/bench_cem.py > $O/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/**/*kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
with open("$O/summary.txt", "w") as out:
    for r in rows:
        line = "%-100s %6s calls %10.1f us avg" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3)
        print(line)
        out.write(line + "\n")
PY
