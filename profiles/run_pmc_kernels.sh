#!/bin/bash
# PMC passes for the kernel probe incl. the out-of-cache scene: bash profiles/run_pmc_kernels.sh <tag>
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1
cd /tmp
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $R/gpurun_out/pmck_${TAG}_$i -- python3 $R/bench_kernels.py --big 512 --no-oracle --rays 8388608 --reps 2 > $R/gpurun_out/pmck_${TAG}_$i.log 2> $R/gpurun_out/pmck_${TAG}_$i.err
  echo "pass $i done"
done
