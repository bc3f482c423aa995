#!/bin/bash
# Round 4, twenty-second call: fuzz soak on the final build (hash 9145f31e105e9c04: the kernels of 7f83b5c40a871245 + exception guards
# in host code), seeds no earlier run saw (CRT_FUZZ_BASE=40000).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rm -f gpurun_out/r04_fuzz_census3.txt
soak() { local n=$1; shift; env "$@" CRT_FUZZ_BASE=40000 CRT_FUZZ_CENSUS=gpurun_out/r04_fuzz_census3.txt timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider $KSEL > gpurun_out/r04_fuzz_soak3_$n.log 2>&1 || { tail -5 gpurun_out/r04_fuzz_soak3_$n.log; exit 1; }; echo "$n: $(tail -1 gpurun_out/r04_fuzz_soak3_$n.log)"; }
KSEL=""; soak default CRT_FUZZ_EXTRA=4000
KSEL="-k world"; soak wide1 CRT_FUZZ_EXTRA=1500 CRT_WIDE=1 CRT_FUSED=0 CRT_STAGE_MIN_PATHS=1
KSEL="-k world"; soak wide2_direct CRT_FUZZ_EXTRA=1500 CRT_DIRECT_LEAVES=1 CRT_WIDE=2 CRT_FUSED=0 CRT_STAGE_MIN_PATHS=1
cat gpurun_out/r04_fuzz_census3.txt
echo twentysecond done
