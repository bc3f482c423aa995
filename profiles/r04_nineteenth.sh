#!/bin/bash
# Round 4, nineteenth call: rocprofv3 --kernel-trace --stats of the openpbr_showcase line (the last of other_configs without one);
# the example hosts' own output lines (C host, RCCL host at world 1) on a scene each.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; T=r04
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats_showcase -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --scene openpbr_showcase --steps 2 --warmup 1 > $R/gpurun_out/${T}_stats_showcase.log 2>&1) || exit 1
cp gpurun_out/${T}_stats_showcase/*/*kernel_stats.csv gpurun_out/${T}_rocprofv3_kernel_stats_showcase.csv
python - > gpurun_out/${T}_example_hosts.txt 2>&1 <<'PY'
import importlib, os, sys, tempfile
sys.path.insert(0, "tests")
import host_c_scene as hc
crt = importlib.import_module("crust-render_amd")
tmp = tempfile.mkdtemp()
exe, exe_rccl = hc.build_host(tmp), hc.build_rccl_host(tmp)
for name, w, h, depth, spp in (("veach_mis", 1920, 1080, 8, 64), ("PointInstancedMedCity", 1920, 1080, 8, 16)):
    path = os.path.join("scenes", name + (".usda" if name != "PointInstancedMedCity" else ".usd"))
    desc = crt.usda.load(path, w, h) if path.endswith(".usda") else crt.load_usda(path, w, h, depth)[1]
    desc.settings["max_depth"] = depth
    _s, mats, _p = crt.usda.build_world(desc, crt, crt.default_material)
    blob = hc.scene_blob(crt, desc, mats, spp, spp)
    for e, env in ((exe, {}), (exe_rccl, {"RANK": "0", "WORLD_SIZE": "1", "NCCL_SOCKET_IFNAME": "lo"})):
        res, film = hc.run_host(e, blob, tmp, env, timeout=280)
        print(name, "rc", res.returncode, res.stdout.strip(), res.stderr.strip()[-300:])
PY
cat gpurun_out/${T}_example_hosts.txt
echo nineteenth done
