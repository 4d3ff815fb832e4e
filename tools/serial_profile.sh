#!/bin/bash
# Kernel time by family with NOTHING overlapped (one lane, wgrad on the lane's own stream): the undiluted cost of each kernel
# family of the step, to set against the overlapped wall clock.   bash tools/serial_profile.sh [tag]  -> gpurun_out/serial_stats_<tag>/
# (MMIDET_TWIN from the environment decides between twin launches and the per-lane form)
set -u
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
export MMIDET_TWO_STREAMS=0 MMIDET_OVERLAP_WGRAD=0
rm -rf $R/gpurun_out/serial_stats_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/serial_stats_$TAG -o s -- \
  python3 $R/bench.py --steps 8 --warmup 2 --mode eager --no-cpu-baseline --no-split-probe --no-roofline > $R/gpurun_out/serial_stats_$TAG.log 2>&1 \
  || { echo "profile run failed: see gpurun_out/serial_stats_$TAG.log"; tail -n 20 $R/gpurun_out/serial_stats_$TAG.log; exit 1; }
find $R/gpurun_out/serial_stats_$TAG -name "*kernel_stats.csv" | head -1
