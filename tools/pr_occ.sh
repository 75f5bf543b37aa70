#!/bin/bash
# sweep time per work class at different residencies (experiment builds, see tools/pr_kmask.sh)
for b in ${BPC:-1 2 3 4}; do echo "blocks_per_cu=$b" >> ${1:-gpurun_out/pr_occ.log}; SS_PR_BLOCKS_PER_CU=$b LIBS=${LIBS:-kmask} MASKS="${MASKS:-0x100 0x200 0xFFFFFFFF}" bash tools/pr_kmask.sh ${1:-gpurun_out/pr_occ.log}; done
