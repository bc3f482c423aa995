#!/bin/bash
# Round 4, seventeenth call: the shade stage writes a shadow request's ray (1) / ray and contribution (2) to its queue slot before it
# samples the BSDF — parity of both on the lit scenes, then A/B against the form that carries them (0).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; T=${1:-r04s}
for v in early1 early2; do
  CRT_AMD_LIB=$PWD/variants/$v.so timeout -k 10 500 python -m pytest tests/test_gpu_render.py tests/test_gpu_fuzz.py tests/test_gpu_stress.py -m gpu -x -q -p no:cacheprovider -k "image_and_counters or random_world or lights_at_infinity or stress or pipelines" > gpurun_out/${T}_tests_$v.log 2>&1; echo "$v pytest rc=$?"; tail -1 gpurun_out/${T}_tests_$v.log
  grep -q " failed\| error" gpurun_out/${T}_tests_$v.log && exit 1
done
: > gpurun_out/${T}_ab.txt
bash profiles/ab_env.sh "early0 early1 early2" --scene veach_mis --spp 512 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "early0 early1 early2" --scene stress --spp 256 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "early0 early1 early2" --scene openpbr_showcase --spp 512 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "early0 early1 early2" --scene synthetic:big --spp 128 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
cat gpurun_out/${T}_ab.txt
echo seventeenth done
