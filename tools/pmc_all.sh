#!/bin/bash
# SQ / cache counters of the two headline kernels (separate rocprofv3 --pmc passes, kernel trace only beside them):
#   tools/pmc_all.sh r02b  ->  gpurun_out/r02b_sq_counters.txt     (copy into profiles/)
tag=${1:-rXX}
o=gpurun_out/${tag}_sq_counters.txt
: > $o
sets=("SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE")
# LDS=1: only the LDS counters (a second call: five sets do not fit one gpurun limit)
if [ "$LDS" = "1" ]; then sets=("SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"); o=gpurun_out/${tag}_lds_counters.txt; : > $o; fi
for set in "${sets[@]}"; do
  echo "## pagerank sweep / probe: $set" >> $o
  bash tools/pmc_pr2.sh "$set" ${tag}a >> $o 2>&1 || exit 1
  echo "## scoring kernels: $set" >> $o
  bash tools/pmc_score.sh "$set" ${tag}b >> $o 2>&1 || exit 1
done
