#!/bin/bash
# Round 4, sixteenth call: the whole GPU suite on the final tree (with the integrator-on-the-seams tests); the bench line once more on
# whichever box of the pool this is.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r04_tests.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r04_tests.log
grep -q " failed" gpurun_out/r04_tests.log && exit 1
python bench.py > gpurun_out/r04_bench_default_run2.json 2> gpurun_out/r04_bench_run2.err || exit 1
python -c "
import json; d=json.loads(open('gpurun_out/r04_bench_default_run2.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], [(o['workload'].split()[0], o['value']) for o in d['other_configs']], d['roofline']['pmc_source'])"
echo sixteenth done
