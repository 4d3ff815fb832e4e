#!/bin/bash
# HBM traffic of the whole training step by kernel family, from the PMC counters (run on the GPU box):
#   bash tools/pmc_step.sh [workload]     -> gpurun_out/pmc_step_traffic.json  (copy to profiles/ to have bench.py report it)
# Two rocprofv3 passes (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"), counters only
# with --kernel-trace; the program itself comes right after `--` (no wrapper that would re-exec under the preloaded tool).
set -u
W=${1:-l_fourier}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_step_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_step_$c -o p -- \
    python3 $R/bench.py --workload $W --steps 2 --warmup 1 --mode eager --no-cpu-baseline --no-split-probe --no-roofline \
    > $R/gpurun_out/pmc_step_$c.log 2>&1 || { echo "pass $c failed: stopping (see gpurun_out/pmc_step_$c.log)"; tail -n 20 $R/gpurun_out/pmc_step_$c.log; exit 1; }
done
python3 $R/tools/pmc_step_summary.py $R/gpurun_out/pmc_step_FETCH_SIZE $R/gpurun_out/pmc_step_WRITE_SIZE $W > $R/gpurun_out/pmc_step_traffic.json
cat $R/gpurun_out/pmc_step_traffic.json
