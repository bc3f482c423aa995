#!/bin/bash
# Round 4, eighth call: whole-packet fetch in the four-wave kernels on large trees; grid and lane counts on the large workloads.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; T=${1:-r04i}
: > gpurun_out/${T}_ab.txt
bash profiles/ab_env.sh "cur lean0" --scene stress --spp 256 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "cur lean0" --scene synthetic:big --spp 128 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "cur lean0" --scene cornellbox --spp 256 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "cur@CRT_GRID_MULT=8 cur@CRT_GRID_MULT=16 cur@CRT_GRID_MULT=32 cur@CRT_GRID_MULT=64 cur" --scene stress --spp 256 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "cur@CRT_GRID_MULT=3 cur@CRT_GRID_MULT=6 cur@CRT_GRID_MULT=12 cur@CRT_GRID_MULT=24 cur" --scene PointInstancedMedCity --width 3840 --height 2160 --spp 128 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
cat gpurun_out/${T}_ab.txt
echo eighth done
