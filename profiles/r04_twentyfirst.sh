#!/bin/bash
# Round 4, twenty-first call: the backend's other instruction-scheduling strategies for the kernels (-mllvm -amdgpu-sched-strategy=
# max-ilp / max-memory-clause) against the default, four workloads.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; T=${1:-r04ab}
: > gpurun_out/${T}_ab.txt
bash profiles/ab_env.sh "cur maxilp maxclause" --scene cornellbox --spp 512 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "cur maxilp maxclause" --scene veach_mis --spp 512 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "cur maxilp maxclause" --scene stress --spp 256 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "cur maxilp maxclause" --scene PointInstancedMedCity --width 3840 --height 2160 --spp 128 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
cat gpurun_out/${T}_ab.txt
echo twentyfirst done
