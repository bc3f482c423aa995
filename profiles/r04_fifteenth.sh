#!/bin/bash
# Round 4, fifteenth call: fuzz worlds with direct leaf words forced into every image (CRT_DIRECT_LEAVES=1), so that each one runs the
# four-wave DIRECT-engine kernels (CRT_WIDE=2, one launch per stage) — the fourteenth call's wide2 run only met them on worlds whose
# images carry direct words by themselves.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rm -f gpurun_out/r04_fuzz_census.txt
CRT_FUZZ_CENSUS=gpurun_out/r04_fuzz_census.txt CRT_FUZZ_BASE=10000 CRT_FUZZ_EXTRA=1500 CRT_DIRECT_LEAVES=1 CRT_WIDE=2 CRT_FUSED=0 CRT_STAGE_MIN_PATHS=1 timeout -k 10 500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider -k world > gpurun_out/r04_fuzz_soak_wide2_direct.log 2>&1 || { tail -5 gpurun_out/r04_fuzz_soak_wide2_direct.log; exit 1; }
tail -1 gpurun_out/r04_fuzz_soak_wide2_direct.log
cat gpurun_out/r04_fuzz_census.txt
echo fifteenth done
