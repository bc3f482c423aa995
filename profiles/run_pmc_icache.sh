#!/bin/bash
# Instruction-cache counters for the bench kernels: bash profiles/run_pmc_icache.sh <tag> [bench args...]
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
cd /tmp
i=0
for PMC in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $R/gpurun_out/pmci_${TAG}_$i -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/pmci_${TAG}_$i.log 2> $R/gpurun_out/pmci_${TAG}_$i.err || echo "pass $i failed"
  echo "pass $i done"
done
