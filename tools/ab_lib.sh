#!/bin/bash
# A/B of two builds of the kernel library: conv micro-bench and the full step, interleaved.  bash tools/ab_lib.sh /path/to/other.so
cd "$(dirname "$0")/.."
. tools/ab_common.sh
OTHER=$1
echo "== conv micro-bench, this build"; ab_run_cmd "bench_conv" python tools/bench_conv.py
echo "== conv micro-bench, $OTHER"; ab_run_cmd "bench_conv other" env MMIDET_HIP_LIB=$OTHER python tools/bench_conv.py
for i in 1 2 3; do
  run X=1
  run MMIDET_HIP_LIB=$OTHER
done
