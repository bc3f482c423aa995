#!/bin/bash
# Round-4 PMC collection: the separate rocprofv3 --kernel-trace --pmc passes of run_pmc_r02.sh, with a choice of passes
# (PASSES="1 3 4": wave time + VALU, DRAM reads + L2 hit, DRAM writes — what hbm_measured_frac / l2_hit need; default all seven).
#   bash profiles/run_pmc_r04.sh <tag> <program.py> [args...]
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
PROG=$1; shift
PASSES=${PASSES:-"1 2 3 4 5 6 7"}
cd /tmp
P[1]="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD"
P[2]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
P[3]="TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum"
P[4]="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_DRAM_sum"
P[5]="FETCH_SIZE"
P[6]="WRITE_SIZE"
P[7]="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
for i in $PASSES; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc ${P[$i]} --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python3 $R/$PROG "$@" > $R/gpurun_out/pmc_${TAG}_$i.log 2> $R/gpurun_out/pmc_${TAG}_$i.err
  echo "pass $i done"
done
