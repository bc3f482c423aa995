#!/bin/bash
# rocprofv3 --kernel-trace --stats of the stress and MedCity lines (the kernels behind bench.py's other_configs)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; T=r04
stats() { local n=$1; shift
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats_$n -- python3 $R/bench.py --no-cpu-baseline --no-other-configs "$@" > $R/gpurun_out/${T}_stats_$n.log 2>&1)
  cp $R/gpurun_out/${T}_stats_$n/*/*kernel_stats.csv $R/gpurun_out/${T}_rocprofv3_kernel_stats_$n.csv; }
stats stress --scene stress --spp-per-step 256 --steps 2 --warmup 1
stats medcity --scene PointInstancedMedCity --width 3840 --height 2160 --spp-per-step 128 --steps 2 --warmup 1
stats veach --scene veach_mis --steps 2 --warmup 1
head -6 gpurun_out/${T}_rocprofv3_kernel_stats_stress.csv | cut -c1-160
echo stats done
