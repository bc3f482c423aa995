#!/usr/bin/env python3
"""bench.py — Mray/s of the wavefront path tracer on BASELINE.json's configs[1].

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (config.workload): samples/cornellbox.usda at 1920x1080, scene-default depth 32, triangle filter r=1,
adaptive stopping off (variance 0), frame 0. A "step" is ONE wavefront batch: `--spp-per-step` samples (default
256) of every pixel a rank owns — generate, then up to depth+1 rounds of extend / shade / shadow, then the film
fold — with the scene, the path-state planes (87 GB at this size: batches are sized for 288 GB of HBM — longer queue
segments drain the traversal's ray pools less often: 128 / 192 / 256 spp per batch measure 8179 / 8258 / 8430 Mray/s,
profiles/README.md) and the film resident in HBM before the timed region starts. 1024 spp is 4 such steps, the default K.

Metric (stats.rs:150-152): Mray/s = (closest_hit + shadow_rays) / render seconds / 1e6, summed
over all ranks; the timed region is K steps bracketed by barrier + synchronize, MAX over ranks.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL). The frame's 16x16 tiles are dealt round-robin
to the ranks; each rank traces `--spp-per-step * N` samples of its own tiles per step, so per-GPU work per step is
fixed as N grows ("weak": N GPUs advance the frame N times as many spp per step). Tile ownership is disjoint, so
the only collective is ONE RCCL all_gather of the per-rank tile buffers after the last step (inside the timed
region), plus the 64-bit ray counters.

roofline: for the dominant kernel of the pipeline the renderer chose for the scene — k_extend, the BVH4 closest-hit
traversal, for flat triangle scenes (one launch per stage and bounce, four workgroups per CU); k_path, which runs a whole
batch (generate, every bounce's traversal, shading and shadow stage) in one launch, for instance-heavy scenes:
algorithmic bytes (SURVEY §8d formulas; traversal counters from the stats build of the same stages on the same batch)
over the HIP-event duration of its launches in the timed region, against the 8 TB/s HBM peak. cpu_baseline: the
oracle (CPU restatement, "port") rendering a bounded sample of the same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2  # wave64 VALU instructions/s: 256 CUs x 4 SIMD-32, one instruction per 2 cycles at 2.4 GHz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp-per-step", type=int, default=256)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default="cornellbox")
    ap.add_argument("--depth", type=int, default=None, help="override max depth (default: the scene's, 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=24, help="spp of the bounded CPU-baseline sample (per run; 3 runs of 24 spp are ~10 s on 16 cores)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = this box's share: min(affinity, 16 per GPU))")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1; gloo (CPU tensors, ranks may share a GPU) rehearses the N > 1 path on a 1-GPU box")
    ap.add_argument("--cpu-reps", type=int, default=3, help="the CPU baseline is the best of this many runs (BASELINE.md §3: min of 3)")
    ap.add_argument("--dump-frame", default=None, help="rank 0 writes the gathered frame ([height*width, 3] float32, .npy) here")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU) and relay rank 0's
        # line. Nothing in this process has touched the GPU yet (no torch import, no HIP call), and the launcher is a
        # CHILD process whose exit code we return — never an exec.
        raise SystemExit(_self_launch(args.gpus))

    import numpy as np
    import torch
    from __graft_entry__ import load_package

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world
    n_dev = max(torch.cuda.device_count(), 1)
    if args.backend == "nccl" and local_rank >= n_dev:
        raise SystemExit("rank %d has no GPU of its own (%d visible): RCCL needs one GPU per rank" % (local_rank, n_dev))
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    coll = "cuda" if args.backend == "nccl" else "cpu"  # where collective operands live
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist_mod.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist_mod.init_process_group(backend="gloo")
        dist = dist_mod

    crt = load_package()
    crt.lib()
    # a sample scene file, or a labelled synthetic scene ("synthetic:city[:side]", crust-render_amd/synthetic.py)
    path = args.scene if args.scene.startswith("synthetic:") else os.path.join(ROOT, "scenes", args.scene + ".usda")
    if not args.scene.startswith("synthetic:") and not os.path.exists(path):
        path = os.path.join(ROOT, "scenes", args.scene + ".usd")  # binary crate (PointInstancedMedCity)
    r, desc = crt.load_usda(path, args.width, args.height, args.depth, rank=rank, world=world)
    spp_step = args.spp_per_step * world
    stream = torch.cuda.current_stream()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    sample = 0
    for _ in range(args.warmup):
        r.render_samples(sample, spp_step, stream)
        sample += spp_step
    barrier()
    r.clear(stream)  # film sums and ray counters restart; buffers stay resident
    r.profile(True)
    n_pix_total = args.width * args.height
    d_rgb = torch.empty(r.n_pix * 3, dtype=torch.float32, device="cuda")
    plan = crt.shard.GatherPlan(args.width, args.height, world, coll) if dist is not None else None
    if dist is not None:  # untimed: the first collective of a kind pays the communicator's channel setup
        r.film_to(d_rgb, stream)
        plan.gather(d_rgb.reshape(-1, 3).to(coll), dist, dst=0)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        r.render_samples(k * spp_step, spp_step, stream)
    r.film_to(d_rgb, stream)
    frame = None
    if dist is not None:  # the one data-path collective: gather the tile buffers (padded to the largest shard) to rank 0
        frame = plan.gather(d_rgb.reshape(-1, 3).to(coll), dist, dst=0)
    barrier()
    elapsed = time.perf_counter() - t0
    if args.dump_frame and rank == 0:
        full = frame if frame is not None else crt.shard.GatherPlan(args.width, args.height, 1, "cuda").gather(d_rgb.reshape(-1, 3))
        np.save(args.dump_frame, full.cpu().numpy())
    st = r.stats()
    prof = r.profile_read()
    r.profile(False)

    rays = torch.tensor([st.closest_hit, st.shadow_rays, st.camera_rays, st.vertices], dtype=torch.int64, device=coll)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll)
    if dist is not None:
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    rays = rays.cpu().tolist()
    elapsed = float(tmax.item())
    total_rays = rays[0] + rays[1]
    value = total_rays / elapsed / 1e6

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel, N=1 figures of rank 0 ----
    # The renderer picks its pipeline per scene (crt.h, crt_renderer_pipeline). FUSED (instance-heavy, sphere-only):
    # ONE kernel per step, k_path — generate + every bounce's traversal, shading and shadow stage of a workgroup-private
    # queue segment. PER-STAGE (flat triangle scenes: cornellbox, the default workload): one launch per stage and
    # bounce; the dominant kernel is k_extend, the BVH4 closest-hit traversal, at four workgroups per CU.
    # Algorithmic bytes per launch: the per-ray traversal formula summed over the rays the
    # kernel traces in one batch (counters from the stats build of the same stages on the same batch) [+ fused: the
    # shadow rays' and 352 B per shaded vertex: path state 96 B each way + 160 B material record], over its launches.
    pipe = r.pipeline()
    fused = pipe["fused"]
    ext, sh = r.render_samples_stats(0, spp_step, stream)  # stats build, same batch shape; not timed
    trav_bytes = ext.algorithmic_bytes() + sh.algorithmic_bytes()
    shade_bytes = 352 * (st.vertices // max(args.steps, 1))
    launch_bytes = (trav_bytes + shade_bytes) if fused else ext.algorithmic_bytes()
    k_ms, k_n = prof["extend"]["ms"], prof["extend"]["launches"]   # class 0: k_path when fused, k_extend otherwise
    per_launch = launch_bytes * args.steps / max(k_n, 1)
    achieved = per_launch / (k_ms / max(k_n, 1) * 1e-3) / 1e9 if k_ms > 0 else 0.0
    # the other pipeline on the same workload, 2 untimed steps: per-stage split of a fused step / the fused kernel's time
    stage = other = None
    saved = {k: os.environ.get(k) for k in ("CRT_FUSED", "CRT_WIDE")}
    try:
        os.environ["CRT_FUSED"] = "0" if fused else "1"
        r2, _ = crt.load_usda(path, args.width, args.height, args.depth, rank=rank, world=world)
        r2.render_samples(0, spp_step, stream)
        torch.cuda.synchronize()
        r2.profile(True)
        for k in range(2):
            r2.render_samples(k * spp_step, spp_step, stream)
        torch.cuda.synchronize()
        p2 = r2.profile_read()
        if fused:
            stage = {k: round(v["ms"] / 2, 3) for k, v in p2.items()}
            stage["extend_algorithmic_gb_s"] = round(ext.algorithmic_bytes() / (p2["extend"]["ms"] / 2 * 1e-3) / 1e9, 1)
        else:
            other = {"fused_k_path_ms_per_step": round(p2["extend"]["ms"] / 2, 3), "other": round(p2["other"]["ms"] / 2, 3)}
        del r2
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    # Measured counters of the same kernel, from the committed PMC passes IF they are of this build and workload.
    workload_key = "%s %dx%d %dspp" % (args.scene, args.width, args.height, spp_step)
    pmc, pmc_note = _pmc_for(workload_key, "k_path" if fused else "k_extend") if world == 1 else (None, "N > 1")
    avg_s = (k_ms / max(k_n, 1)) * 1e-3
    traffic = hbm_frac = l2_hit = valu_issue = wave = None
    if pmc and avg_s > 0:
        traffic = int(pmc["ea_dram_read_bytes_per_launch"] + pmc["ea_dram_write_bytes_per_launch"])
        hbm_frac = round(traffic / avg_s / 1e9 / HBM_PEAK_GBS, 4)
        l2_hit = round(pmc["l2_hit_rate"], 4)
        # wave64 VALU instructions per second against the SIMDs' issue ceiling: CUs x 4 SIMD-32 x clock / 2 cycles
        valu_issue = round(pmc["valu_insts_per_launch"] / avg_s / (VALU_ISSUE_PEAK), 4)
        # where a wave's time goes (SQ_ACTIVE_INST_ANY, SQ_WAIT_ANY, SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES)
        rnd = lambda x: None if x is None else round(x, 4)
        wave = {"issuing": rnd(pmc.get("inst_active_frac")), "waiting_on_memory": rnd(pmc.get("wait_any_frac")),
                "waiting_for_instructions": rnd(pmc.get("wait_inst_frac"))}
    waves = 4 if pipe["wide"] else 3
    roofline = {
        "kernel": "k_path (one launch per batch: generate + BVH4 traversal + shading + shadow per queue segment)" if fused
                  else "k_extend (BVH4 closest-hit traversal of one bounce's rays, %d workgroups per CU)" % waves,
        "pipeline": "fused" if fused else ("per-stage, %d workgroups per CU in the traversal kernels" % waves),
        # `achieved` / `frac` are the task's definition: ALGORITHMIC bytes over kernel time against the HBM
        # peak. On this cache-resident scene most of those bytes never leave LDS / L1 / L2 — the physical picture is in
        # the measured fields below: fabric-side traffic is a fraction of the peak, a wave spends its time half issuing
        # instructions and half waiting on memory; the kernel is bound by latency and issue at its occupancy, not by HBM.
        "bound": "hbm",
        "achieved": round(achieved, 2),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5),
        "algorithmic_gb_s": round(achieved, 2),
        "traffic": traffic,                    # measured L2<->fabric bytes per launch (TCC_EA0_*_DRAM_32B x 32 B), or null
        "hbm_measured_frac": hbm_frac,         # traffic / launch time / 8 TB/s
        "l2_hit": l2_hit,
        "valu_issue_frac": valu_issue,
        "wave_time": wave,
        "physical_bound": ("dependent-load latency and per-wave instruction issue at %d waves/SIMD (see wave_time, "
                           "hbm_measured_frac, valu_issue_frac)" % waves) if pmc else None,
        "pmc_source": pmc_note,
        "kernel_source_hash": kernel_source_hash(),
        "launches": k_n,
        "launches_per_step": round(k_n / max(args.steps, 1), 2),
        "avg_launch_ms": round(k_ms / max(k_n, 1), 5),
        "bytes_per_launch": int(per_launch),
        "traversal_bytes_per_ray": round(ext.algorithmic_bytes() / max(int(ext.rays), 1), 1),
        "shading_bytes_per_vertex": 352,
        "kernel_ms": {k: round(v["ms"], 3) for k, v in prof.items()},
        "unfused_stage_ms_per_step": stage,
        "other_pipeline": other,
    }

    cpu = None
    if not args.no_cpu_baseline and world == 1:  # reported at N=1 only
        cpu = _cpu_baseline(crt, desc, args)

    out = {
        "metric": "Mray/s (closest-hit + shadow queries) of the path-tracing integrator",
        "value": round(value, 2),
        "unit": "Mray/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": ("synthetic scene %s (generated in code, NOT a reference file), seeded sampler, frame 0" % args.scene)
                if args.scene.startswith("synthetic:") else
                ("the reference's own sample scene file (scenes/%s), seeded sampler, frame 0" % os.path.basename(path)),
        "config": {
            "workload": "%s %dx%d, %d spp per step per GPU-share (x%d GPUs), depth %d, triangle r=1, "
                        "variance 0; %d steps = %d spp" % (args.scene if args.scene.startswith("synthetic:") else "samples/" + os.path.basename(path), args.width, args.height, args.spp_per_step, world,
                                                           r.settings.max_depth, args.steps, args.steps * spp_step),
            "spp_per_step": spp_step,
            "paths_per_step_per_gpu": r.n_pix * spp_step,
            "rays_total": total_rays,
            "closest_hit": rays[0],
            "shadow_rays": rays[1],
            "camera_rays": rays[2],
            "mean_path_length": round(rays[3] / max(rays[2], 1), 3),
            "seconds": round(elapsed, 4),
            "spp_per_second": round(args.steps * spp_step / elapsed, 2),
            # the second half of BASELINE's metric: wall clock to the configuration's target spp (configs[1]: 1024) at this rate
            "target_spp": 1024,
            "seconds_to_target_spp": round(elapsed * 1024.0 / (args.steps * spp_step), 4),
            "sharding": ("16x16 pixel tiles round-robin over ranks; one %s gather of tile buffers to rank 0" % ("RCCL" if args.backend == "nccl" else "gloo (rehearsal)")) if world > 1 else "none",
        },
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def _self_launch(n):
    """Starts `python -m torch.distributed.run --nproc-per-node n bench.py <same flags>` as a child process, relays
    its stderr, prints rank 0's single JSON line, returns the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this image
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    lines = [l for l in res.stdout.splitlines() if l.lstrip().startswith("{")]
    for l in res.stdout.splitlines():
        if l not in lines and l.strip():
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1])
    return res.returncode if res.returncode else (0 if lines else 1)


def kernel_source_hash():
    """sha256 (first 16 hex digits) over the native sources the kernels are built from (crust-render_amd/csrc: *.cpp,
    *.h, kernels/*, Makefile) — the key that ties a committed PMC profile to the build it was taken on."""
    import hashlib
    base = os.path.join(ROOT, "crust-render_amd", "csrc")
    files = []
    for d, _dirs, names in os.walk(base):
        if os.path.basename(d).startswith("_obj"):
            continue
        for n in names:
            if n.endswith((".cpp", ".h", ".hip", "Makefile")):
                files.append(os.path.join(d, n))
    h = hashlib.sha256()
    for f in sorted(files):
        h.update(os.path.relpath(f, base).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


PMC_PROFILE = os.path.join(ROOT, "profiles", "r02_pmc_bench.json")


def _pmc_for(workload_key, kernel):
    """The committed rocprofv3 --pmc passes for this workload and kernel (profiles/r02_pmc_bench.json, written by
    profiles/summarize_pmc_bench.py) — but only if they were taken on THIS build: the file carries the kernel source
    hash of the build it profiled, and a profile of other sources says nothing about the kernel being timed now.
    Returns (entry or None, note)."""
    if not os.path.exists(PMC_PROFILE):
        return None, "no PMC profile committed"
    try:
        with open(PMC_PROFILE) as f:
            prof = json.load(f)
    except Exception as e:  # noqa: BLE001
        return None, "unreadable PMC profile: %s" % e
    here = kernel_source_hash()
    if prof.get("kernel_source_hash") != here:
        return None, "PMC profile is of build %s, this build is %s: traffic not reported" % (prof.get("kernel_source_hash"), here)
    e = prof.get("workloads", {}).get(workload_key)
    if e is None:
        return None, "no PMC passes for workload %r" % workload_key
    if not str(e.get("kernel", "")).startswith(kernel):
        return None, "PMC passes for workload %r are of %s, the pipeline now runs %s" % (workload_key, e.get("kernel"), kernel)
    return e, "profiles/r02_pmc_bench.json (git %s)" % prof.get("git_commit", "?")


def _cpu_baseline(crt, desc, args):
    """The oracle (kind 'port': CPU restatement of the reference's algorithm, all host cores, 16x16 tiles as
    tracer.rs:424-459) on a bounded sample of the same workload: the full frame at --cpu-spp samples."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ora_world
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    cores = args.cpu_threads if args.cpu_threads > 0 else min(avail, 16)  # a 1-GPU box's CPU share is 16 cores
    o = ora_world.OracleRenderer(desc, crt.usda, max_depth=args.depth, forward=0)
    times = []
    for _ in range(max(args.cpu_reps, 1)):  # min of N render-phase times (scripts/bench_scenes.sh:33 convention)
        t0 = time.perf_counter()
        _, st = o.render(args.cpu_spp, threads=cores)
        times.append(time.perf_counter() - t0)
    dt = min(times)
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except Exception:
        pass
    return {
        "value": round(st.total_rays() / dt / 1e6, 3),
        "unit": "Mray/s",
        "cores": cores,
        "kind": "port",
        "sample": "%dx%d full frame at %d spp (%d rays), best of %d runs (%s s), reference-order estimator, host: %s" % (
            args.width, args.height, args.cpu_spp, st.total_rays(), len(times), " / ".join("%.1f" % t for t in times), model),
    }


if __name__ == "__main__":
    main()
