#!/bin/bash
# Round-2 PMC collection: separate rocprofv3 passes, counters only with --kernel-trace (MI355X_MICROARCH.md).
# HBM-side traffic comes from the L2's fabric request counters in 32-byte units (TCC_EA0_RDREQ_DRAM_32B_sum,
# TCC_EA0_WRREQ_WRITE_DRAM_32B_sum: "1 64-byte request counted as 2, 128-byte as 4" per rocprofv3 -L), which need no
# access-width correction; FETCH_SIZE / WRITE_SIZE are collected next to them for comparison with round 1.
# Usage on the GPU box:  bash profiles/run_pmc_r02.sh <tag> <program.py> [args...]
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
PROG=$1; shift
cd /tmp
i=0
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_DRAM_sum" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python3 $R/$PROG "$@" > $R/gpurun_out/pmc_${TAG}_$i.log 2> $R/gpurun_out/pmc_${TAG}_$i.err
  echo "pass $i done"
done
