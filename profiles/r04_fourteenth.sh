#!/bin/bash
# Round 4, fourteenth call, on the final build: rocprofv3 stats of the other_configs' kernels, smoke(), fuzz soak (default pipelines; the
# four-wave per-stage kernels forced: flat engine copy = CRT_WIDE=1, direct engine copy = CRT_WIDE=2; the fused kernel).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash profiles/r04_stats_extra.sh > gpurun_out/r04_stats_extra.log 2>&1 || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_smoke.log 2>&1 || { tail -5 gpurun_out/r04_smoke.log; exit 1; }
soak() { local n=$1; shift; env "$@" timeout -k 10 500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider $KSEL > gpurun_out/r04_fuzz_soak_$n.log 2>&1 || { tail -5 gpurun_out/r04_fuzz_soak_$n.log; exit 1; }; tail -1 gpurun_out/r04_fuzz_soak_$n.log; }
KSEL=""; soak default CRT_FUZZ_BASE=10000 CRT_FUZZ_EXTRA=2000
KSEL="-k world"; soak wide1 CRT_FUZZ_BASE=10000 CRT_FUZZ_EXTRA=1000 CRT_WIDE=1 CRT_FUSED=0 CRT_STAGE_MIN_PATHS=1
KSEL="-k world"; soak wide2 CRT_FUZZ_BASE=10000 CRT_FUZZ_EXTRA=1000 CRT_WIDE=2 CRT_FUSED=0 CRT_STAGE_MIN_PATHS=1
KSEL="-k world"; soak fused CRT_FUZZ_BASE=10000 CRT_FUZZ_EXTRA=500 CRT_FUSED=1
echo fourteenth done
