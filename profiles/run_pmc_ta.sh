#!/bin/bash
# Texture-addresser / L1 (TCP) passes: is the traversal bound by the vector-memory front end (64 divergent 128-byte
# node fetches per wave and step) rather than by latency?   bash profiles/run_pmc_ta.sh <tag> <program.py> [args...]
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
PROG=$1; shift
cd /tmp
i=0
# (TA_BUSY_* and TA_*_STALLED_BY_* were tried first: those passes never finished on this pool — not collected.)
for PMC in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN2_sum TCP_TAGRAM0_REQ_sum" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python3 $R/$PROG "$@" > $R/gpurun_out/pmc_${TAG}_$i.log 2> $R/gpurun_out/pmc_${TAG}_$i.err || { echo "pass $i failed: stopping"; exit 1; }
  echo "pass $i done"
done
