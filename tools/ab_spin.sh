#!/bin/bash
# host sensitivity of the step: a busy-wait of n microseconds after every library call (about 3000 per step)
cd "$(dirname "$0")/.."
. tools/ab_common.sh
run MMIDET_HOST_SPIN_US=0
run MMIDET_HOST_SPIN_US=2
run MMIDET_HOST_SPIN_US=4
run MMIDET_HOST_SPIN_US=0
run MMIDET_HOST_SPIN_US=6
