#!/bin/bash
# usage: tools/run_pmc.sh <tag> <conv shape args...>   (run on the GPU box; writes gpurun_out/pmc_<tag>_*.csv)
set -u
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_$i -o p -- python3 $R/tools/pmc_conv.py "$@" > $R/gpurun_out/pmc_${tag}_$i.log 2>&1 || { echo "pass $i failed: stopping (see gpurun_out/pmc_${tag}_$i.log)"; tail -n 20 $R/gpurun_out/pmc_${tag}_$i.log; exit 1; }
done
ls $R/gpurun_out/pmc_${tag}_1 | head
