"""The C-ABI library loads and exports every symbol include/crt.h declares (no compute calls: runs without a GPU)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "crt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(crt_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(crt):
    names = _declared()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(crt.lib(), n)]
    assert not missing, missing
    assert sorted(crt.ABI_SYMBOLS) == names, set(names) ^ set(crt.ABI_SYMBOLS)


def test_struct_layouts_match_the_header(crt, tmp_path):
    """sizeof of every struct, as gcc lays out include/crt.h, equals the ctypes mirror's."""
    import ctypes as C
    import subprocess
    names = ["CrtRay", "CrtRayHit", "CrtMaterial", "CrtLight", "CrtCamera", "CrtRenderSettings", "CrtTravStats", "CrtRayStats"]
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "crt.h"\nint main(void){' +
                   "".join('printf("%s %%zu\\n", sizeof(%s));' % (n, n) for n in names) + "return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for n in names:
        assert int(out[n]) == C.sizeof(getattr(crt, n)), n
    assert C.sizeof(crt.CrtRay) == 48 and C.sizeof(crt.CrtRayHit) == 40  # ray.rs:18-23, scene.rs:134-142


def test_host_only_entry_points_work_without_a_device(crt):
    b = crt.SceneBuilder()
    b.attach_sphere((0, 0, 0), 1.0)
    s = b.commit()  # host-side build: no HIP call
    assert s.primitive_count() == 1 and s.geometry_count() == 1
    assert crt.lib().crt_version().startswith(b"crt_amd")
    m = crt.default_material()
    assert abs(m.specular_ior - 1.5) < 1e-7 and m.kind == crt.MAT_OPENPBR


def test_no_product_module_touches_the_oracle():
    """The product path must never import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "crust-render_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for needle in ("liboracle", "import ora", '#include "ora', "oracle/_build", "-loracle"):
                    assert needle not in text, (os.path.join(dirpath, f), needle)


def test_the_library_reads_the_environment_in_one_place():
    """Every A/B knob goes through crt::read_knobs (csrc/scene.cpp; crt_internal.h, Knobs): validated and clamped there.
    Nothing else under csrc/ calls getenv (21 call sites in the scene / renderer set-up paths, one of
    which took any integer)."""
    import re
    csrc = os.path.join(ROOT, "crust-render_amd", "csrc")
    sites = []
    for dirpath, _, files in os.walk(csrc):
        if os.path.basename(dirpath).startswith("_obj"):
            continue
        for f in files:
            if f.endswith((".cpp", ".h", ".hip")):
                for n, line in enumerate(open(os.path.join(dirpath, f), errors="ignore"), 1):
                    if re.search(r"\bgetenv\s*\(", line) and not line.lstrip().startswith("//"):
                        sites.append((f, n))
    assert len(sites) == 1 and sites[0][0] == "scene.cpp", sites


def test_shading_seam_records_have_the_sizes_the_header_states(crt):
    import ctypes as C
    sh = crt.shading
    assert (sh.SHADE_QUERY.itemsize, sh.SCATTER_SAMPLE.itemsize, sh.BSDF_EVAL.itemsize, sh.LIGHT_QUERY.itemsize,
            sh.LIGHT_SAMPLE.itemsize) == (80, 48, 32, 48, 48)
    assert C.sizeof(crt.CrtMaterial) == 224 and C.sizeof(crt.CrtLight) == 84
    # host-only entry points of the round: the padded tile count needs no device
    assert crt.shard.padded_count(1920, 1080, 8) == crt.lib().crt_shard_padded_count(1920, 1080, 8) > 0
    assert crt.shard.padded_count(1920, 1080, 8) % 256 == 0 and crt.shard.padded_count(1920, 1080, 8) >= crt.shard.shard_pixels(1920, 1080, 0, 8).size


def test_shading_seam_argument_checks_need_no_device(crt):
    """The seam's entry points validate their arguments before they touch a device: nothing to do is CRT_OK, a missing
    query / result pointer is CRT_ERR_BAD_ARG (and, on a box without a gfx950 device, anything else CRT_ERR_NO_DEVICE —
    there is no CPU fallback behind the ABI)."""
    L = crt.lib()
    for name in ("crt_material_scatter_n", "crt_material_eval_n", "crt_material_emitted_n", "crt_light_sample_n",
                 "crt_light_pdf_n", "crt_light_escaped_n"):
        f = getattr(L, name)
        assert f(None, 0, None, 0, None, None) == 0                      # n == 0
        assert f(None, 0, None, 5, None, None) == -1                     # queries / results missing
        assert f(None, 3, 1, 5, 1, None) == -1                           # a table of 3 records without a pointer
    assert L.crt_gather_plan_assemble(None, None, None, None) == -1 and L.crt_gather_plan_padded_count(None) == 0
    assert L.crt_renderer_set_lanes(None, 2) == -1


def test_nothing_unwinds_through_the_abi(crt):
    """A host in C or Rust cannot catch a C++ exception: entry points that allocate host memory report
    CRT_ERR_NO_MEMORY (reason in crt_last_error) instead — crt_reserve with a capacity no allocator can give
    (Vec::reserve panics on it, scene.rs:174), and the builder is usable afterwards."""
    import ctypes as C
    L = crt.lib()
    b = L.crt_builder_new()
    try:
        L.crt_reserve.argtypes = [C.c_void_p, C.c_size_t]
        assert L.crt_reserve(b, C.c_size_t(2 ** 64 - 1)) == -6
        assert b"crt_reserve" in L.crt_last_error()
        assert L.crt_reserve(b, C.c_size_t(2 ** 62)) == -6  # under max_size, far beyond memory: bad_alloc
        assert L.crt_reserve(b, 4) == 0
        gid = C.c_uint32(99)
        c = (C.c_float * 3)(0.0, 0.0, 0.0)
        assert L.crt_attach_sphere(b, c, C.c_float(1.0), C.c_uint32(0xFFFFFFFF), C.byref(gid)) == 0 and gid.value == 0
        assert L.crt_count(b) == 1
    finally:
        L.crt_builder_free(b)
    assert crt._ERRORS[-6] == "CRT_ERR_NO_MEMORY"
