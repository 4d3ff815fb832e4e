#!/bin/bash
# Round-4 measurement batch on the GPU box: the driver's bench command, kernel-trace stats of the same command (eager), the serial
# (nothing overlapped) profile, PMC step traffic, the other workloads.  Trace CSVs are deleted after the summaries are made.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4m
mkdir -p $O
cd $R
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { echo "default bench failed"; tail -n 20 $O/bench_default.err; exit 1; }
cut -c1-300 $O/bench_default.json
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kstats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats -o s -- python3 $R/bench.py --steps 8 --warmup 2 --mode eager --no-cpu-baseline --no-split-probe --no-roofline > $O/kstats.log 2>&1 \
  || { echo "kernel-trace run failed"; tail -n 20 $O/kstats.log; exit 1; }
cd $R
f=$(find $O/kstats -name "*kernel_trace.csv" | head -1)
python3 tools/exposed.py $f 1 > $O/exposed.txt 2>&1
rm -f $f
MMIDET_TWIN=1 bash tools/serial_profile.sh r4 > $O/serial.log 2>&1 || { echo "serial profile failed"; tail -n 5 $O/serial.log; }
find $R/gpurun_out/serial_stats_r4 -name "*kernel_trace.csv" -delete
bash tools/pmc_step.sh l_fourier > $O/pmc.log 2>&1 || { echo "pmc failed"; tail -n 20 $O/pmc.log; }
find $R/gpurun_out/pmc_step_FETCH_SIZE $R/gpurun_out/pmc_step_WRITE_SIZE -name "*.csv" -size +20M -delete 2>/dev/null
for w in s_add s_fourier x_1280; do
  python3 bench.py --workload $w --no-cpu-baseline --no-split-probe --steps 8 --warmup 3 > $O/bench_$w.json 2> $O/bench_$w.err || { echo "bench $w failed"; tail -n 20 $O/bench_$w.err; continue; }
  python3 -c "import json,sys; j=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); print('$w', j['value'], j['ms_per_step'], j['roofline']['frac'], j['config']['launch_mode'], j['config']['launch_mode_probe_ms'])"
done
python3 bench.py --storage bf16 --no-cpu-baseline --no-split-probe --steps 10 --warmup 3 > $O/bench_bf16.json 2> $O/bench_bf16.err && python3 -c "import json; j=json.load(open('$O/bench_bf16.json')); print('bf16 storage', j['value'], j['ms_per_step'])"
