#!/bin/bash
# Where the native transport's +3 ms at world size 1 come from: the plain single-GPU step (no reducer, no collective) with only
# parts of a transport's SET-UP present in the process (bench.py MMIDET_COMM_BISECT), interleaved with the plain run.
cd "$(dirname "$0")/.."
. tools/ab_common.sh
run X=1
run MMIDET_COMM_BISECT=gloo
run MMIDET_COMM_BISECT=rccl
run MMIDET_COMM_BISECT=nccl
run X=1
run MMIDET_COMM_BISECT=gloo+rccl
run MMIDET_COMM_BISECT=rccl
run MMIDET_COMM_BISECT=gloo
