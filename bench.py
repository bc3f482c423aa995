#!/usr/bin/env python3
"""bench.py — Mray/s of the wavefront path tracer on BASELINE.json's configs[1].

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (config.workload): samples/cornellbox.usda at 1920x1080, scene-default depth 32, triangle filter r=1,
adaptive stopping off (variance 0), frame 0. A "step" is ONE wavefront batch: `--spp-per-step` samples (default
512) of every pixel a rank owns — generate, then up to depth+1 rounds of extend / shade / shadow, then the film
fold — with the scene, the path-state planes (174 GB at this size: batches are sized for 288 GB of HBM — a launch's
ramp-up and drain are paid once per batch and bounce: 256 / 384 / 512 / 640 / 768 spp per batch measure 9307 / 9529 /
9832 / 9919 / 9970 Mray/s, profiles/README.md; a part with less free HBM halves the batch until it fits) and the film
resident in HBM before the timed region starts. 1024 spp is 2 such steps; the default K of 4 is 2048 spp.

Metric (stats.rs:150-152): Mray/s = (closest_hit + shadow_rays) / render seconds / 1e6, summed
over all ranks; the timed region is K steps bracketed by barrier + synchronize, MAX over ranks.

A batch runs as up to four LANES (crt.h, crt_renderer_lanes): sub-batches of consecutive samples on their own HIP
streams whose launches overlap; the film is folded in sample order. config/roofline.lanes reports the count.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL). The frame's 16x16 tiles are dealt round-robin
to the ranks; each rank traces `--spp-per-step * N` samples of its own tiles per step, so per-GPU work per step is
fixed as N grows ("weak": N GPUs advance the frame N times as many spp per step). Tile ownership is disjoint, so
the only collective is ONE RCCL all_gather of the per-rank tile buffers after the last step (inside the timed
region), plus the 64-bit ray counters.

roofline: for the dominant kernel of the pipeline the renderer chose for the scene — k_extend, the BVH4 closest-hit
traversal, for flat triangle scenes (one launch per stage and bounce, four workgroups per CU); k_path, which runs a whole
batch (generate, every bounce's traversal, shading and shadow stage) in one launch, elsewhere: `achieved` = algorithmic
bytes (SURVEY §8d formulas; traversal counters from the stats build of the same stages on the same batch) over the
HIP-event duration of its launches in the timed region; `bound` / `peak` / `frac` name the memory tier that serves those
bytes according to the committed PMC passes of this build (HBM when the measured fabric traffic reaches 30 % of its
8 TB/s, else the L2 aggregate with bound "latency/issue"); the other stages' figures are under roofline.kernels. cpu_baseline: the
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
    ap.add_argument("--spp-per-step", type=int, default=512)
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
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the other BASELINE configurations that follow the headline at N=1 (other_configs)")
    ap.add_argument("--other-configs-scale", type=int, default=1,
                    help="divide the other configurations' width and height by this and their spp by its square (tests)")
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
    path = crt.scene_path(args.scene)  # .usda text, .usd binary crate (PointInstancedMedCity), .usda.xz (stress)
    # N > 1: rank 0 imports the file, the other ranks receive the description in one broadcast (shard.import_once)
    setup = {}
    hbm_free0 = torch.cuda.mem_get_info()[0]
    r, desc = crt.load_usda(path, args.width, args.height, args.depth, rank=rank, world=world, dist=dist, timings=setup)
    spp_step = args.spp_per_step * world
    stream = torch.cuda.current_stream()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # A part with less free HBM than the batch needs (the default shape holds 174 GB of path state) halves the batch until
    # it fits; the shape actually run is what the line reports (config.spp_per_step). N > 1: every rank runs the smallest
    # shape any rank settled on.
    requested = args.spp_per_step
    args.spp_per_step = fit_batch(crt, r, args.spp_per_step, world, stream, dist, coll)
    spp_step = args.spp_per_step * world
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
    # what the job cost to set up and to hold, per rank (max over ranks): the import (rank 0 parses, the others wait for the
    # broadcast), every rank's own commit, and the HBM this rank's scene image + path state + film occupy
    setup["hbm_used_gb"] = (hbm_free0 - torch.cuda.mem_get_info()[0]) / 1e9
    tset = torch.tensor([setup.get("import_s", 0.0), setup.get("commit_s", 0.0), setup["hbm_used_gb"]], dtype=torch.float64, device=coll)
    if dist is not None:
        dist.all_reduce(tset, op=dist.ReduceOp.MAX)
    setup = dict(zip(("import_s", "commit_s", "hbm_used_gb_per_rank"), (round(float(x), 2) for x in tset.tolist())))

    # the job's RayStats: all eight counters (stats.rs:128-147), one all_reduce
    all_stats = crt.shard.reduce_ray_stats(st, dist, coll)
    stat_names = [f for f, _t in st._fields_]
    rays = [all_stats[stat_names.index(f)] for f in ("closest_hit", "shadow_rays", "camera_rays", "vertices")]
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    total_rays = rays[0] + rays[1]
    value = total_rays / elapsed / 1e6

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- roofline, N=1 figures of rank 0 ----
    # The renderer picks its pipeline per scene and batch (crt.h, crt_renderer_pipeline). FUSED: ONE kernel per step,
    # k_path — generate + every bounce's traversal, shading and shadow stage of a workgroup-private queue segment.
    # PER-STAGE: one launch per stage and bounce (k_extend / k_shade / k_shadow). Per kernel: ALGORITHMIC bytes per launch
    # (the per-ray traversal formula summed over the rays the kernel traces in one batch —
    # counters from the stats build of the same stages on the same batch shape — and 352 B per shaded vertex: path state
    # 96 B each way + 160 B material record) over the HIP-event duration of its launches in the timed region.
    pipe = r.pipeline()
    fused = pipe["fused"]
    lanes = r.lanes()
    ext, sh = r.render_samples_stats(0, spp_step, stream)  # stats build, same batch shape; not timed
    n_steps = max(args.steps, 1)
    alg = {"extend": ext.algorithmic_bytes(), "shadow": sh.algorithmic_bytes(), "shade": 352 * (st.vertices // n_steps)}
    if fused:
        alg = {"extend": alg["extend"] + alg["shadow"] + alg["shade"]}  # class 0 of the profile is k_path
    names = {"extend": "k_path" if fused else "k_extend", "shade": "k_shade", "shadow": "k_shadow"}
    workload_key = "%s %dx%d %dspp" % (args.scene, args.width, args.height, spp_step)
    waves = 4 if pipe["wide"] else 3
    kernels = {}
    for cls, per_step in alg.items():
        k_ms, k_n = prof[cls]["ms"], prof[cls]["launches"]
        if k_n == 0 or k_ms <= 0:
            continue
        # (the committed PMC passes are taken with CRT_LANES=1: their per-launch figures are a whole batch's launch)
        kernels[names[cls]] = _roofline_entry(names[cls], per_step * n_steps / k_n, k_ms, k_n, n_steps, workload_key, world,
                                              traffic_scale=1.0 / lanes)
    dominant = max(kernels, key=lambda k: kernels[k]["total_ms"]) if kernels else None
    # the other pipeline on the same workload, 2 untimed steps: per-stage split of a fused step / the fused kernel's time.
    # The timed renderer's buffers (174 GB at the default shape) are released first.
    settings_depth = r.settings.max_depth
    n_pix = r.n_pix
    del r
    stage = other = None
    # With lanes the launches of the timed region overlap: a launch's own duration then includes the time it shares the
    # chip with the other lanes' launches, and says little about the kernel. The same batch once more as ONE lane
    # (CRT_LANES=1, 2 untimed steps): every kernel alone on the chip, the durations the fractions below are priced on.
    serial = None
    if lanes > 1:
        saved_l = os.environ.get("CRT_LANES")
        try:
            os.environ["CRT_LANES"] = "1"
            r1 = None
            r1, _ = crt.load_usda(path, args.width, args.height, args.depth, rank=rank, world=world)
            r1.render_samples(0, spp_step, stream)
            torch.cuda.synchronize()
            r1.profile(True)
            for k in range(2):
                r1.render_samples(k * spp_step, spp_step, stream)
            torch.cuda.synchronize()
            p1 = r1.profile_read()
            serial = {}
            for cls, per_step in alg.items():
                if p1[cls]["launches"] and p1[cls]["ms"] > 0:
                    serial[names[cls]] = _roofline_entry(names[cls], per_step * 2 / p1[cls]["launches"], p1[cls]["ms"], p1[cls]["launches"], 2,
                                                         workload_key, world)
            serial_ms = {k: round(v["ms"] / 2, 3) for k, v in p1.items()}
        except crt.CrtError as e:
            serial = None
        finally:
            r1 = None  # its whole-batch buffer (174 GB at the default shape) must be gone before the next renderer is built
            if saved_l is None:
                os.environ.pop("CRT_LANES", None)
            else:
                os.environ["CRT_LANES"] = saved_l
    saved = {k: os.environ.get(k) for k in ("CRT_FUSED", "CRT_WIDE", "CRT_LANES")}
    try:
        os.environ["CRT_FUSED"] = "0" if fused else "1"
        os.environ["CRT_LANES"] = "1"  # one lane: the other pipeline's launches alone on the chip, comparable with `serial`
        r2 = None
        r2, _ = crt.load_usda(path, args.width, args.height, args.depth, rank=rank, world=world)
        r2.render_samples(0, spp_step, stream)
        torch.cuda.synchronize()
        if r2.pipeline()["fused"] != fused:  # scenes with lights at infinity have no fused form
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
    except crt.CrtError as e:  # the probe is informational: a box with less free HBM still prints the line
        other = {"error": str(e)}
    finally:
        r2 = None
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    roofline = dict(kernels[dominant]) if dominant else {}
    if serial and dominant in serial:
        # headline figures on the un-overlapped launch; what the timed region's HIP events measured stays beside them
        timed = {k: roofline.get(k) for k in ("achieved", "frac", "avg_launch_ms", "launches", "launches_per_step", "total_ms",
                                              "bytes_per_launch", "hbm_measured_frac", "valu_issue_frac", "bound", "peak")}
        roofline = dict(serial[dominant])
        roofline["timing"] = ("2 untimed steps as ONE lane (CRT_LANES=1) after the timed region: every launch alone on the chip. "
                              "In the timed region the batch runs as %d lanes whose launches overlap (timed_region: a launch's own "
                              "start-to-end time on its stream, which includes the time it shares the chip)" % lanes)
        roofline["timed_region"] = timed
        roofline["serial_kernel_ms_per_step"] = serial_ms
    roofline.update({
        "kernel": {"k_path": "k_path (one launch per batch: generate + BVH4 traversal + shading + shadow per queue segment)",
                   "k_extend": "k_extend (BVH4 closest-hit traversal of one bounce's rays, %d workgroups per CU)" % waves,
                   "k_shade": "k_shade (one path vertex per live path: emission, NEE sample, OpenPBR sample, roulette, compaction)",
                   "k_shadow": "k_shadow (BVH4 any-hit traversal of one bounce's shadow rays, %d workgroups per CU)" % waves}.get(dominant),
        "pipeline": "fused" if fused else ("per-stage, %d workgroups per CU in the traversal kernels" % waves),
        # a batch runs as `lanes` sub-batches on their own streams: their launches OVERLAP, so a launch's own duration
        # (avg_launch_ms, HIP events on its stream) includes the time it shares the chip, and the per-step sum of the
        # kernels' times (kernel_ms) exceeds ms_per_step by that overlap
        "lanes": lanes,
        "kernel_source_hash": kernel_source_hash(),
        "traversal_bytes_per_ray": round(ext.algorithmic_bytes() / max(int(ext.rays), 1), 1),
        "shadow_bytes_per_ray": round(sh.algorithmic_bytes() / max(int(sh.rays), 1), 1) if int(sh.rays) else None,
        "shading_bytes_per_vertex": 352,
        "kernel_ms": {k: round(v["ms"], 3) for k, v in prof.items()},
        "kernels": {k: v for k, v in (serial if serial else kernels).items() if k != dominant},  # the other stages of the per-stage pipeline
        "unfused_stage_ms_per_step": stage,
        "other_pipeline": other,
    })

    cpu = cpu_all = None
    if not args.no_cpu_baseline and world == 1:  # reported at N=1 only
        cpu = _cpu_baseline(crt, desc, args)
        # ... and on EVERY hardware thread this process may run on, as the reference's Rayon pool would take them
        # (tracer.rs:424-459); 4x the samples so the sample is still seconds of work
        cpu_all = _cpu_baseline(crt, desc, args, all_cores=True)
    # The other BASELINE configurations on the same clock (N = 1, headline scene only): after the headline, 2 timed steps
    # each of the reference's stress scene, configs[4]'s own file at its own 3840x2160, configs[3] and configs[2] at 1080p.
    others = None
    if world == 1 and not args.no_other_configs and args.scene == "cornellbox":
        others = []
        k = max(args.other_configs_scale, 1)
        for scene, w, h, spp in [(sc, max(w // k, 16), max(h // k, 16), max(spp // (k * k), 2)) for sc, w, h, spp in OTHER_CONFIGS]:
            try:
                others.append(other_config(crt, torch, scene, w, h, spp, stream))
            except Exception as e:  # noqa: BLE001  (informational: the headline still prints)
                others.append({"workload": "%s %dx%d %dspp" % (scene, w, h, spp), "error": "%s: %s" % (type(e).__name__, e)})

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
                                                           settings_depth, args.steps, args.steps * spp_step),
            "spp_per_step": spp_step,
            "spp_per_step_requested": requested * world,
            "paths_per_step_per_gpu": n_pix * spp_step,
            "rays_total": total_rays,
            "closest_hit": rays[0],
            "shadow_rays": rays[1],
            "camera_rays": rays[2],
            "mean_path_length": round(rays[3] / max(rays[2], 1), 3),
            "ray_stats": dict(zip(stat_names, all_stats)),
            "seconds": round(elapsed, 4),
            "spp_per_second": round(args.steps * spp_step / elapsed, 2),
            # the second half of BASELINE's metric: wall clock to the configuration's target spp (configs[1]: 1024) at this rate
            "target_spp": 1024,
            "seconds_to_target_spp": round(elapsed * 1024.0 / (args.steps * spp_step), 4),
            "setup": setup,
            "sharding": ("16x16 pixel tiles round-robin over ranks; one %s gather of tile buffers to rank 0" % ("RCCL" if args.backend == "nccl" else "gloo (rehearsal)")) if world > 1 else "none",
        },
        "roofline": roofline,
        "cpu_baseline": cpu,
        "cpu_baseline_all_cores": cpu_all,
        "other_configs": others,
    }
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


# (scene, width, height, spp per step): BASELINE.json configs[2..4] at their own resolutions + the reference's stress
# scene (scripts/gen_stress_scene.py, docs/simd.md:221) — the path-traced workloads that call `occluded`.
OTHER_CONFIGS = [("stress", 1920, 1080, 256), ("PointInstancedMedCity", 3840, 2160, 128), ("veach_mis", 1920, 1080, 512),
                 ("openpbr_showcase", 1920, 1080, 512)]


def fit_batch(crt, r, spp_per_step, world, stream, dist=None, coll="cuda"):
    """One untimed batch of spp_per_step * world samples; while the renderer cannot allocate it (CrtError), half of it.
    Returns the per-GPU-share spp that fits — with N > 1 the smallest any rank settled on, so every rank times the same
    shape. The probe's samples are cleared from the film and the counters."""
    import torch
    sync = torch.cuda.synchronize if torch.cuda.is_available() else (lambda: None)  # (the CPU test drives it with stand-in renderers)
    while True:
        try:
            r.render_samples(0, spp_per_step * world, stream)
            sync()
            break
        except crt.CrtError:
            if spp_per_step <= 1:
                raise
            spp_per_step //= 2
    if dist is not None:
        t = torch.tensor([spp_per_step], dtype=torch.int64, device=coll)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        spp_per_step = int(t.item())
    r.clear(stream)
    sync()
    return spp_per_step


def other_config(crt, torch, scene, width, height, spp, stream, steps=2):
    """One of the other configurations: scene import and commit untimed, one untimed batch (which also sizes the batch to
    the free HBM), then `steps` timed batches bracketed by synchronize; the dominant kernel's roofline tier from the
    committed PMC passes of this build, priced on the timed region's HIP events and the stats build's counters."""
    path = crt.scene_path(scene)
    t_imp = time.perf_counter()
    r, _desc = crt.load_usda(path, width, height, None)
    import_s = time.perf_counter() - t_imp
    spp = fit_batch(crt, r, spp, 1, stream)
    r.profile(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        r.render_samples(k * spp, spp, stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    st, prof, pipe, lanes = r.stats(), r.profile_read(), r.pipeline(), r.lanes()
    r.profile(False)
    timed_ms = {k: round(v["ms"] / steps, 3) for k, v in prof.items()}
    # With lanes the launches of the timed region overlap and a launch's own duration includes the time it shares the chip:
    # ONE more untimed batch as one lane — every launch alone — is what the roofline fractions are priced on (as the
    # headline's are).
    if lanes > 1 and hasattr(r, "set_lanes"):
        r.set_lanes(1)
        r.profile(True)
        r.render_samples(0, spp, stream)
        torch.cuda.synchronize()
        prof = r.profile_read()
        r.profile(False)
        r.set_lanes(4)
        priced_steps, priced_scale, timing = 1, 1.0, "one untimed batch as ONE lane after the timed region (every launch alone on the chip)"
    else:
        priced_steps, priced_scale, timing = steps, 1.0 / lanes, "HIP events of the timed region (%d lane%s)" % (lanes, "" if lanes == 1 else "s, overlapping")
    ext, sh = r.render_samples_stats(0, spp, stream)
    depth = r.settings.max_depth
    r = None
    alg = {"extend": ext.algorithmic_bytes(), "shadow": sh.algorithmic_bytes(), "shade": 352 * (st.vertices // steps)}
    if pipe["fused"]:
        alg = {"extend": alg["extend"] + alg["shadow"] + alg["shade"]}
    names = {"extend": "k_path" if pipe["fused"] else "k_extend", "shade": "k_shade", "shadow": "k_shadow"}
    key = "%s %dx%d %dspp" % (scene, width, height, spp)
    kernels = {}
    for cls, per_step in alg.items():
        if prof[cls]["launches"] and prof[cls]["ms"] > 0:
            kernels[names[cls]] = _roofline_entry(names[cls], per_step * priced_steps / prof[cls]["launches"], prof[cls]["ms"],
                                                  prof[cls]["launches"], priced_steps, key, 1, traffic_scale=priced_scale)
    dom = max(kernels, key=lambda k: kernels[k]["total_ms"]) if kernels else None
    e = kernels.get(dom, {})
    total = st.total_rays()
    return {
        "workload": "%s %dx%d, %d spp per step, depth %d" % (scene, width, height, spp, depth),
        "value": round(total / elapsed / 1e6, 2), "unit": "Mray/s", "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps,
        "closest_hit": int(st.closest_hit), "shadow_rays": int(st.shadow_rays),
        "pipeline": "fused" if pipe["fused"] else ("per-stage, %d workgroups per CU in the traversal kernels" % (4 if pipe["wide"] else 3)),
        "lanes": lanes, "import_and_commit_s": round(import_s, 2),
        # what the number is NOT pinned by: the crate reader that decodes config 5's file is checked
        # against itself only — openusd 0.6 and any reference fixture for this file are absent from the reference tree
        **({"note": "binary USDC crate decoded by crust-render_amd/usdc.py: decoding unpinned against openusd (absent); GPU == oracle on the decoded scene"}
           if scene == "PointInstancedMedCity" else {}),
        "roofline": {"kernel": dom, "bound": e.get("bound"), "tier": e.get("tier"), "frac": e.get("frac"),
                     "achieved": e.get("achieved"), "peak": e.get("peak"), "unit": "GB/s",
                     "hbm_measured_frac": e.get("hbm_measured_frac"), "l2_hit": e.get("l2_hit"), "traffic": e.get("traffic"),
                     "avg_launch_ms": e.get("avg_launch_ms"), "launches": e.get("launches"), "pmc_source": e.get("pmc_source"),
                     "timing": timing, "serial_kernel_ms_per_step": {k: round(v["ms"] / priced_steps, 3) for k, v in prof.items()},
                     "timed_region_kernel_ms_per_step": timed_ms},
    }


L2_PEAK_GBS = 34500.0  # MI355X_MICROARCH.md: L2 aggregate ~34.5 TB/s (eight XCDs x 4 MiB)


def _roofline_entry(kernel, bytes_per_launch, k_ms, k_n, n_steps, workload_key, world, traffic_scale=1):
    """One kernel's roofline figures. `achieved` is the contract's definition — ALGORITHMIC bytes per launch over the
    average launch duration (HIP events). `bound` / `peak` / `frac` name the tier that actually serves those bytes:
    with the committed PMC passes of this build at hand (profiles/r03_pmc_bench.json, keyed by kernel source hash), a
    kernel whose measured fabric traffic is below 30 % of the HBM peak is NOT HBM-bound — its bytes come out of L2 / L1 /
    LDS and it is priced against the L2 aggregate (bound "latency/issue": see wave_time); otherwise against HBM. Without
    PMC passes for this build the tier is decided from the algorithmic rate alone (more bytes per second than HBM can
    deliver cannot be an HBM figure). The algorithmic rate against the HBM peak stays in hbm_frac_algorithmic."""
    avg_s = k_ms / k_n * 1e-3
    achieved = bytes_per_launch / avg_s / 1e9
    pmc, note = _pmc_for(workload_key, kernel) if world == 1 else (None, "N > 1")
    traffic = hbm_frac = l2_hit = valu_issue = wave = None
    if pmc:
        # the PMC passes profile whole-batch launches (CRT_LANES=1); a lane's launch covers `traffic_scale` of one
        traffic = int((pmc["ea_dram_read_bytes_per_launch"] + pmc["ea_dram_write_bytes_per_launch"]) * traffic_scale)
        hbm_frac = round(traffic / avg_s / 1e9 / HBM_PEAK_GBS, 4)
        l2_hit = round(pmc["l2_hit_rate"], 4)
        valu_issue = round(pmc["valu_insts_per_launch"] * traffic_scale / avg_s / VALU_ISSUE_PEAK, 4)
        rnd = lambda x: None if x is None else round(x, 4)
        wave = {"issuing": rnd(pmc.get("inst_active_frac")), "waiting_on_memory": rnd(pmc.get("wait_any_frac")),
                "waiting_for_instructions": rnd(pmc.get("wait_inst_frac"))}
    hbm_bound = (hbm_frac >= 0.3) if hbm_frac is not None else (achieved <= HBM_PEAK_GBS * 0.9)
    peak = HBM_PEAK_GBS if hbm_bound else L2_PEAK_GBS
    algorithmic = achieved
    basis = "algorithmic bytes (SURVEY 8d) / launch duration"
    if hbm_bound and traffic is not None and algorithmic > traffic / avg_s / 1e9:
        # An HBM-bound kernel whose caches serve part of its algorithmic bytes: against the HBM peak only the bytes that
        # crossed the fabric count (the PMC passes' figure) — an algorithmic rate above the peak is not a fraction of it
        achieved = traffic / avg_s / 1e9
        basis = "measured fabric bytes (PMC) / launch duration: the kernel is HBM-bound and L2 serves part of its algorithmic bytes"
    return {
        "bound": "hbm" if hbm_bound else "latency/issue",
        "tier": "hbm" if hbm_bound else "l2",
        "achieved": round(achieved, 2),
        "achieved_basis": basis,
        "peak": peak,
        "unit": "GB/s",
        "frac": round(achieved / peak, 5),
        "algorithmic_gb_s": round(algorithmic, 2),
        "hbm_frac_algorithmic": round(algorithmic / HBM_PEAK_GBS, 5),
        "l2_frac": round(algorithmic / L2_PEAK_GBS, 5),
        "traffic": traffic,                    # measured L2<->fabric bytes per launch (TCC_EA0_*_DRAM_32B x 32 B), or null
        "hbm_measured_frac": hbm_frac,         # traffic / launch time / 8 TB/s
        "l2_hit": l2_hit,
        "valu_issue_frac": valu_issue,
        "wave_time": wave,
        "pmc_source": note,
        "launches": k_n,
        "launches_per_step": round(k_n / n_steps, 2),
        "avg_launch_ms": round(k_ms / k_n, 5),
        "total_ms": round(k_ms, 3),
        "bytes_per_launch": int(bytes_per_launch),
    }


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


def _pmc_profile_path():
    """The committed per-kernel PMC summary to read: the newest profiles/r*_pmc_bench.json (named per round)."""
    d = os.path.join(ROOT, "profiles")
    names = sorted(n for n in os.listdir(d) if n.startswith("r") and n.endswith("_pmc_bench.json")) if os.path.isdir(d) else []
    return os.path.join(d, names[-1]) if names else os.path.join(d, "r04_pmc_bench.json")


PMC_PROFILE = _pmc_profile_path()


def _pmc_for(workload_key, kernel):
    """The committed rocprofv3 --pmc passes for this workload and kernel (profiles/r03_pmc_bench.json, written by
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
    w = prof.get("workloads", {}).get(workload_key)
    if w is None:
        return None, "no PMC passes for workload %r" % workload_key
    e = w.get("kernels", {}).get(kernel)
    if e is None:
        return None, "PMC passes for workload %r hold %s, not %s" % (workload_key, sorted(w.get("kernels", {})), kernel)
    return e, "%s (git %s)" % (os.path.relpath(PMC_PROFILE, ROOT), prof.get("git_commit", "?"))


def _cpu_quota(default):
    """Cores' worth of CPU time this container is granted: cgroup v2 cpu.max / v1 cfs quota; a huge number if none."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
            if q != "max":
                return max(int(round(int(q) / int(p))), 1)
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
            q, p = int(f.read()), int(g.read())
            if q > 0:
                return max(int(round(q / p)), 1)
    except Exception:
        pass
    return 1 << 30


def _cpu_baseline(crt, desc, args, all_cores=False):
    """The oracle (kind 'port': CPU restatement of the reference's algorithm, 16x16 tiles on worker threads as
    tracer.rs:424-459) on a bounded sample of the same workload: the full frame at --cpu-spp samples, on this box's
    16-core share — or, all_cores, on every hardware thread the process may run on (what the reference's Rayon pool
    would take), capped by the container's cgroup CPU quota where one is visible."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ora_world
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    granted = _cpu_quota(avail)  # what the container may actually burn: the cgroup CPU quota, if one is set
    cores = args.cpu_threads if args.cpu_threads > 0 else min(avail, 16)  # a 1-GPU box's CPU share is 16 cores
    cpu_spp = args.cpu_spp
    if all_cores:
        # every core the box grants this process: its affinity mask (256 hardware threads on the pool's hosts) capped by
        # the cgroup quota — 256 threads on a 16-core quota only thrash (measured: 20.8 against 31.2 Mray/s on 16)
        cores = max(min(avail, granted), 1)
    o = ora_world.OracleRenderer(desc, crt.usda, max_depth=args.depth, forward=0)
    times = []
    # min of N render-phase times (scripts/bench_scenes.sh:33 convention); the all-cores leg is one run: where the pool
    # grants a 1-GPU box 16 cores' worth of CPU without a visible quota, 256 threads are 3 x slower per ray than 16
    for _ in range(1 if all_cores else max(args.cpu_reps, 1)):
        t0 = time.perf_counter()
        _, st = o.render(cpu_spp, threads=cores)
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
        "host_affinity": avail,  # hardware threads this process may run on; `cores` of them were used
        "cgroup_cpu_quota": granted if granted < 1 << 20 else None,  # cores' worth of CPU time the container is granted (null: no quota)
        "kind": "port",
        "sample": "%dx%d full frame at %d spp (%d rays), best of %d runs (%s s), reference-order estimator, host: %s" % (
            args.width, args.height, cpu_spp, st.total_rays(), len(times), " / ".join("%.1f" % t for t in times), model),
        # the host is shared with the other GPUs' tenants: what 256 threads get of it depends on their load at the moment
        "host_loadavg_1m": round(os.getloadavg()[0], 1) if hasattr(os, "getloadavg") else None,
        **({"note": "threads = every hardware thread this process may run on (what the reference's Rayon pool would take), capped by a "
                    "visible cgroup quota; the pool grants a 1-GPU box 16 cores' worth of a host it shares, so this figure can be "
                    "LOWER than the 16-thread one (round 4: 20.8 against 31.2 Mray/s) — both are reported, neither is the target"}
           if all_cores else {}),
    }


if __name__ == "__main__":
    main()
