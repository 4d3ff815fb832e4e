#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4/toktrace
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 $R/bench.py --steps 4 --warmup 2 --mode eager --no-cpu-baseline --no-split-probe --no-roofline > $O/run.log 2>&1
f=$(find $O -name '*kernel_trace.csv' | head -1)
python3 $R/tools/token_chain.py $f 0 > $R/gpurun_out/r4/token_chain.txt 2>&1
rm -f $f
tail -5 $R/gpurun_out/r4/token_chain.txt
