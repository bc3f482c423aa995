#!/bin/bash
# Round 4, twentieth call: does RCCL accept two ranks on ONE GPU? (If it does, the native host's two-rank path — id file, two shards,
# all-gather, assemble — can run on the one-GPU box; if it refuses ("duplicate GPU"), that is recorded and world 1 is what can run.)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python - > gpurun_out/r04_rccl_two_ranks.txt 2>&1 <<'PY'
import importlib, os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, "tests")
import host_c_scene as hc
crt = importlib.import_module("crust-render_amd")
tmp = tempfile.mkdtemp()
exe1, exe = hc.build_host(tmp), hc.build_rccl_host(tmp)
desc = crt.usda.load("scenes/veach_mis.usda", 200, 120)
desc.settings["max_depth"] = 8
_s, mats, _p = crt.usda.build_world(desc, crt, crt.default_material)
blob = hc.scene_blob(crt, desc, mats, 8, 8)
res, film1 = hc.run_host(exe1, blob, tmp)
one = np.fromfile(film1, dtype=np.uint8).copy()
scene = os.path.join(tmp, "scene.bin")
procs = []
for rank in (0, 1):
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", CRT_NCCL_ID_FILE=os.path.join(tmp, "nccl_id"),
               NCCL_SOCKET_IFNAME="lo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs.append(subprocess.Popen(["timeout", "-k", "5", "100", exe, scene, os.path.join(tmp, "film2.bin")], env=env,
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
for rank, p in enumerate(procs):
    out, err = p.communicate()
    print("rank", rank, "rc", p.returncode, out.strip()[-400:], "|", err.strip()[-600:])
if all(p.returncode == 0 for p in procs):
    two = np.fromfile(os.path.join(tmp, "film2.bin"), dtype=np.uint8)
    print("two ranks' frame and counters equal the single-process host's:", bool(np.array_equal(one, two)))
PY
cat gpurun_out/r04_rccl_two_ranks.txt | tail -12
echo twentieth done
