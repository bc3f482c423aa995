"""Scene files for examples/host_c/crt_host.c (the plain-C host over include/crt.h) and its build: the content
crust-render_amd/usda.py's build_world feeds through the Python mirror, as one little-endian blob. Test infrastructure."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = 0x53545243  # "CRTS"


def _u32(*v):
    return np.asarray(v, dtype=np.uint32).tobytes()


def _f32(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    assert n is None or a.size == n, (a.size, n)
    return a.tobytes()


def _mesh(p):
    v = np.ascontiguousarray(p["verts"], dtype=np.float32).reshape(-1, 3)
    i = np.ascontiguousarray(p["idx"], dtype=np.uint32).reshape(-1, 3)
    return _u32(len(v), len(i)) + v.tobytes() + i.tobytes()


def scene_blob(crt, desc, materials, spp, batch):
    """desc: usda.SceneDesc; materials: the CrtMaterial list build_world returned for it."""
    out = [_u32(MAGIC, len(desc.protos), len(desc.geoms))]
    for p in desc.protos:
        if "radius" in p:
            out += [_u32(1), _f32(p.get("center", (0.0, 0.0, 0.0)), 3), _f32([p["radius"]])]
        elif "instances" in p:
            out.append(_u32(2, len(p["instances"])))
            for it in p["instances"]:
                out += [_u32(it["proto"], it["mask"]), _f32(it["l2w"], 12)]
        else:
            out += [_u32(0), _mesh(p)]
    assert len(materials) == len(desc.geoms)
    for g, m in zip(desc.geoms, materials):
        if g["kind"] == "mesh":
            out += [_u32(0, g["mask"]), _mesh(g)]
        elif g["kind"] == "sphere":
            out += [_u32(1, g["mask"]), _f32(g["center"], 3), _f32([g["radius"]])]
        elif g["kind"] == "instance":
            end = g.get("l2w_end")
            out += [_u32(2, g["mask"], g["proto"], 1 if end is not None else 0), _f32(g["l2w"], 12)]
            if end is not None:
                out.append(_f32(end, 12))
        else:
            out.append(_u32(3, g["mask"]))
        out.append(bytes(m))
    lights = crt.make_lights(desc.lights)
    out.append(_u32(len(desc.lights)))
    out.append(bytes(lights)[:len(desc.lights) * 84])
    c = desc.camera
    out += [_f32(c["lookfrom"], 3), _f32(c["lookat"], 3), _f32(c["vup"], 3),
            _f32([c["vfov_deg"], c["aspect"], c["aperture"], c["focus_dist"]])]
    s = desc.settings
    settings = crt.RenderSettings(s["width"], s["height"], s["max_depth"], s["frame"], s["strategy"], s["filter"],
                                  s["filter_radius"], 0.0)
    out.append(bytes(settings.c()))
    out.append(_u32(spp, batch))
    return b"".join(out)


def build_host(out_dir):
    """gcc, C99, nothing but include/crt.h and libcrt_amd.so -> path of the executable."""
    exe = os.path.join(str(out_dir), "crt_host")
    lib_dir = os.path.join(ROOT, "crust-render_amd")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "host_c", "crt_host.c"), "-L", lib_dir, "-lcrt_amd", "-Wl,-rpath," + lib_dir,
           "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    return exe


def build_rccl_host(out_dir):
    """hipcc (the host calls HIP and RCCL itself) -> path of examples/host_rccl's executable."""
    exe = os.path.join(str(out_dir), "crt_rccl_host")
    lib_dir = os.path.join(ROOT, "crust-render_amd")
    cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "host_rccl", "crt_rccl_host.cpp"), "-L", lib_dir, "-lcrt_amd", "-L/opt/rocm/lib",
           "-lrccl", "-Wl,-rpath," + lib_dir, "-o", exe]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    return exe


def run_host(exe, blob, tmp_dir, env=None, timeout=300):
    """-> (CompletedProcess, film path)."""
    scene, film = os.path.join(str(tmp_dir), "scene.bin"), os.path.join(str(tmp_dir), "film.bin")
    with open(scene, "wb") as f:
        f.write(blob)
    if os.path.exists(film):
        os.remove(film)
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([exe, scene, film], capture_output=True, text=True, timeout=timeout, env=e), film
