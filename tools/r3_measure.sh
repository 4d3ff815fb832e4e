#!/bin/bash
# Round-3 measurement batch on the GPU box: default bench, kernel-trace stats, serial profile, PMC traffic, the other workloads.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { echo "default bench failed"; tail -n 20 $O/bench_default.err; exit 1; }
cut -c1-400 $O/bench_default.json
python3 bench.py --dump-gemm $O/gemm_by_shape.txt --no-cpu-baseline --no-split-probe --steps 6 --warmup 2 > $O/bench_shape.json 2> $O/bench_shape.err || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kstats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats -o s -- python3 $R/bench.py --steps 8 --warmup 2 --mode eager --no-cpu-baseline --no-split-probe --no-roofline > $O/kstats.log 2>&1 \
  || { echo "kernel-trace run failed"; tail -n 20 $O/kstats.log; exit 1; }
cd $R
MMIDET_TWIN=1 bash tools/serial_profile.sh twin || exit 1
bash tools/pmc_step.sh l_fourier > $O/pmc.log 2>&1 || { echo "pmc failed"; tail -n 20 $O/pmc.log; exit 1; }
for w in s_add s_fourier x_1280; do
  python3 bench.py --workload $w --no-cpu-baseline --no-split-probe --steps 8 --warmup 3 > $O/bench_$w.json 2> $O/bench_$w.err || { echo "bench $w failed"; tail -n 20 $O/bench_$w.err; exit 1; }
  python3 -c "import json,sys; j=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); print('$w', j['value'], j['ms_per_step'], j['roofline']['frac'], j['config']['launch_mode'])"
done
