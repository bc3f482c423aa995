#!/bin/bash
# Round 4, tenth call: host-side choices on the lit per-stage scenes (three-wave shade beside four-wave traversal; lane counts; batch size).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; T=${1:-r04k}
: > gpurun_out/${T}_ab.txt
bash profiles/ab_env.sh "cur cur@CRT_SHADE_WIDE=0 cur@CRT_LANES=2 cur@CRT_LANES=3" --scene veach_mis --spp 512 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "cur cur@CRT_SHADE_WIDE=0 cur@CRT_LANES=2" --scene stress --spp 256 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "cur cur@CRT_SHADE_WIDE=0" --scene cornellbox_guided --spp 512 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "cur cur@CRT_LANES=2 cur@CRT_LANES=3" --scene openpbr_showcase --spp 512 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "cur" --scene PointInstancedMedCity --width 3840 --height 2160 --spp 160 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "cur" --scene stress --spp 384 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
cat gpurun_out/${T}_ab.txt
echo tenth done
