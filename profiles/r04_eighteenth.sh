#!/bin/bash
# Round 4, eighteenth call: a longer fuzz soak on the final build, seeds no earlier run saw (CRT_FUZZ_BASE=20000).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rm -f gpurun_out/r04_fuzz_census2.txt
soak() { local n=$1; shift; env "$@" CRT_FUZZ_BASE=20000 CRT_FUZZ_CENSUS=gpurun_out/r04_fuzz_census2.txt timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider $KSEL > gpurun_out/r04_fuzz_soak2_$n.log 2>&1 || { tail -5 gpurun_out/r04_fuzz_soak2_$n.log; exit 1; }; echo "$n: $(tail -1 gpurun_out/r04_fuzz_soak2_$n.log)"; }
KSEL=""; soak default CRT_FUZZ_EXTRA=8000
KSEL="-k world"; soak wide1 CRT_FUZZ_EXTRA=3000 CRT_WIDE=1 CRT_FUSED=0 CRT_STAGE_MIN_PATHS=1
KSEL="-k world"; soak wide2_direct CRT_FUZZ_EXTRA=3000 CRT_DIRECT_LEAVES=1 CRT_WIDE=2 CRT_FUSED=0 CRT_STAGE_MIN_PATHS=1
KSEL="-k world"; soak three_wave_stage CRT_FUZZ_EXTRA=3000 CRT_WIDE=0 CRT_FUSED=0 CRT_STAGE_MIN_PATHS=1
KSEL="-k world"; soak fused CRT_FUZZ_EXTRA=3000 CRT_FUSED=1
cat gpurun_out/r04_fuzz_census2.txt
echo eighteenth done
