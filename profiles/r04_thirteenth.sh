#!/bin/bash
# Round 4, thirteenth call: the general-material shade kernel at TWO waves per SIMD (256 VGPRs, no spills) against three (168, 26 spilled).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; T=${1:-r04n}
: > gpurun_out/${T}_ab.txt
bash profiles/ab_env.sh "cur shade2" --scene openpbr_showcase --spp 512 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
bash profiles/ab_env.sh "cur shade2" --scene PointInstancedMedCity --width 3840 --height 2160 --spp 128 --steps 2 >> gpurun_out/${T}_ab.txt 2>&1
cat gpurun_out/${T}_ab.txt
echo thirteenth done
