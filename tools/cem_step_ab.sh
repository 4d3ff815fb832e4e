#!/bin/bash
# In-box A/B of the round-4 CEM work on the whole step: old = rounds 2-3 forms via the switches, new = defaults
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { python $R/bench.py --steps 12 --warmup 4 --mode eager --no-cpu-baseline --no-split-probe --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in 1 2 3; do
  echo -n "old: "; MMIDET_CEM_TWO_PASS=0 MMIDET_CEM_WGRAD_BN=0 MMIDET_CEM_FORM=0 MMIDET_CEM_MID_BLOCKS=1024 run
  echo -n "new: "; run
done
