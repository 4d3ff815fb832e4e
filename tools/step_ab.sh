#!/bin/bash
# In-box A/B of two environments on the whole step (config 3, eager): tools/step_ab.sh "VAR=a ..." "VAR=b ..." [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { env $1 python $R/bench.py --steps 12 --warmup 4 --mode eager --no-cpu-baseline --no-split-probe --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in $(seq 1 ${3:-3}); do
  echo -n "A [$1]: "; run "$1"
  echo -n "B [$2]: "; run "$2"
done
