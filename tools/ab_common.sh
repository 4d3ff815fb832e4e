#!/bin/bash
# Shared runner of the A/B scripts (sourced).  One bench.py run per call; stderr is KEPT (gpurun_out/ab_logs/), the exit
# status is checked, and the first run that fails or whose log shows a GPU fault stops the whole script: after a fault
# nothing further is launched on the box, and the evidence is the log tail printed here.
set -o pipefail
AB_LOG_DIR=${AB_LOG_DIR:-gpurun_out/ab_logs}
mkdir -p "$AB_LOG_DIR"
AB_RUN_NO=0
AB_BENCH_ARGS=${AB_BENCH_ARGS:---steps 16 --warmup 4 --mode eager --no-cpu-baseline --no-split-probe --no-roofline}
AB_FMT=${AB_FMT:-"'%.2f ms/step  %.1f img/s  host enqueue %.1f ms' % (j['ms_per_step'], j['value'], j['config']['host_enqueue_ms_per_step'])"}
ab_fail() {
  echo "!! $1 -- stopping (no further GPU runs).  Last lines of $2:" >&2
  tail -n 25 "$2" >&2
  exit 1
}
ab_run_cmd() {   # ab_run_cmd <label> <command...>: stdout of the command is returned on stdout
  AB_RUN_NO=$((AB_RUN_NO + 1))
  local log="$AB_LOG_DIR/run_${AB_RUN_NO}.err" out="$AB_LOG_DIR/run_${AB_RUN_NO}.out"
  "${@:2}" >"$out" 2>"$log"
  local rc=$?
  if grep -Eq 'Memory access fault|HSA_STATUS|Aborted|core dumped|hipError|Segmentation' "$log"; then ab_fail "GPU fault in '$1'" "$log"; fi
  if [ $rc -ne 0 ]; then ab_fail "'$1' exited with status $rc" "$log"; fi
  cat "$out"
}
run() {          # run VAR=VALUE ... : one bench.py step measurement under these environment settings
  echo "== $*"
  ab_run_cmd "$*" env "$@" python bench.py $AB_BENCH_ARGS |
    python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($AB_FMT)" || exit 1
}
