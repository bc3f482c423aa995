#!/bin/bash
# Collects the PMC passes for the bench kernels (separate runs, counters only with --kernel-trace, as the guide
# prescribes). Usage on the GPU box: bash profiles/run_pmc.sh <tag> [bench args...]
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
cd /tmp
i=0
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/pmc_${TAG}_$i.log 2> $R/gpurun_out/pmc_${TAG}_$i.err
  echo "pass $i done"
done
