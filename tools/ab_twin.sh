#!/bin/bash
# twin launches vs backbone lanes, eager vs captured step, on the BASELINE workloads:  bash tools/ab_twin.sh [workloads...]
cd "$(dirname "$0")/.."
AB_FMT="'%.2f ms/step  %.1f img/s  host enqueue %.1f ms  %s' % (j['ms_per_step'], j['value'], j['config']['host_enqueue_ms_per_step'], j['config']['launch_mode'])"
. tools/ab_common.sh
for w in "${@:-l_fourier}"; do
  for mode in eager graph; do
    AB_BENCH_ARGS="--workload $w --steps 12 --warmup 3 --mode $mode --no-cpu-baseline --no-split-probe --no-roofline"
    run MMIDET_TWIN=1
    run MMIDET_TWIN=0
  done
done
