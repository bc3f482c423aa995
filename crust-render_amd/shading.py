"""The shading seam as callable functions — host mirror of `trait Material` (crates/crust-core/src/material/
material.rs:26-116) and `trait Light` (light.rs:120-151) over the C ABI's batched device entry points
(include/crt.h: crt_material_scatter_n / _eval_n / _emitted_n, crt_light_sample_n / _pdf_n / _escaped_n).

A host integrator that keeps its own trace_path (tracer.rs:1086-1558) calls Scene.intersect_n / occluded_n for the
kernel seam and these for the per-hit shading; the arithmetic is the device code crt_render_samples runs. Tables and
query / result records live in HBM as torch uint8 tensors; numpy structured dtypes give them field names on the host.
There is no CPU fallback: everything here launches kernels of libcrt_amd.so.
"""
import ctypes as C

import numpy as np

SHADE_QUERY = np.dtype([("ray_dir", np.float32, 3), ("material", np.uint32), ("p", np.float32, 3), ("t", np.float32),
                        ("normal", np.float32, 3), ("front_face", np.uint32), ("wi", np.float32, 3),
                        ("cos_theta_o", np.float32), ("sampler_pattern", np.uint32), ("sampler_index", np.uint32),
                        ("_pad", np.uint32, 2)])
SCATTER_SAMPLE = np.dtype([("origin", np.float32, 3), ("some", np.uint32), ("dir", np.float32, 3), ("pdf", np.float32),
                           ("value", np.float32, 3), ("flags", np.uint32)])
BSDF_EVAL = np.dtype([("value", np.float32, 3), ("pdf", np.float32), ("some", np.uint32), ("_pad", np.uint32, 3)])
LIGHT_QUERY = np.dtype([("from", np.float32, 3), ("light", np.uint32), ("u", np.float32), ("v", np.float32),
                        ("_pad", np.uint32, 2), ("point", np.float32, 3), ("_pad2", np.uint32)])
LIGHT_SAMPLE = np.dtype([("direction", np.float32, 3), ("distance", np.float32), ("radiance", np.float32, 3),
                         ("pdf", np.float32), ("some", np.uint32), ("_pad", np.uint32, 3)])
assert (SHADE_QUERY.itemsize, SCATTER_SAMPLE.itemsize, BSDF_EVAL.itemsize, LIGHT_QUERY.itemsize,
        LIGHT_SAMPLE.itemsize) == (80, 48, 32, 48, 48)
FLAG_DELTA, FLAG_MEDIUM = 1, 2


def _crt():
    import sys
    return sys.modules[__name__.rsplit(".", 1)[0]]


def to_device(records, device="cuda:0"):
    """numpy record array (or ctypes array) -> torch uint8 tensor in HBM."""
    import torch
    buf = np.frombuffer(bytes(records), dtype=np.uint8) if not isinstance(records, np.ndarray) else \
        np.ascontiguousarray(records).view(np.uint8).reshape(-1)
    return torch.from_numpy(buf.copy()).to(device)


def _launch(fn, d_table, n_table, d_queries, itemsize, out_bytes, stream):
    import torch
    crt = _crt()
    n = d_queries.numel() * d_queries.element_size() // itemsize
    d_out = torch.empty(max(n * out_bytes, 1), dtype=torch.uint8, device=d_queries.device)
    rc = getattr(crt.lib(), fn)(C.c_void_p(d_table.data_ptr() if n_table else 0), n_table, C.c_void_p(d_queries.data_ptr()), n,
                                C.c_void_p(d_out.data_ptr()), crt._stream_ptr(stream))
    crt._check(rc, fn)
    return d_out[:n * out_bytes]


class DeviceMaterials:
    """The closed Material table (two implementations: OpenPBR, Emissive) resident in HBM; its methods are the trait's,
    batched: queries in, results out, both device tensors (to_device / .cpu().numpy().view(dtype) at the edges)."""

    def __init__(self, materials, device="cuda:0"):
        crt = _crt()
        self.n = len(materials)
        arr = (crt.CrtMaterial * max(self.n, 1))(*materials)
        self.d = to_device(arr, device)

    def scatter_importance(self, d_queries, stream=None):  # material.rs:40-45 -> SCATTER_SAMPLE records
        return _launch("crt_material_scatter_n", self.d, self.n, d_queries, 80, 48, stream)

    def eval(self, d_queries, stream=None):  # material.rs:56-74 -> BSDF_EVAL records
        return _launch("crt_material_eval_n", self.d, self.n, d_queries, 80, 32, stream)

    def emitted_directional(self, d_queries, stream=None):  # material.rs:112-115 -> 3 floats per query
        return _launch("crt_material_emitted_n", self.d, self.n, d_queries, 80, 12, stream)


class DeviceLights:
    """The light list (light.rs:392-395) resident in HBM; the Light trait's methods, batched."""

    def __init__(self, lights, device="cuda:0"):
        crt = _crt()
        self.n = len(lights)
        arr = crt.make_lights(lights) if (self.n and isinstance(lights[0], dict)) else lights
        self.d = to_device(arr, device)

    def sample_li(self, d_queries, stream=None):  # light.rs:126 -> LIGHT_SAMPLE records
        return _launch("crt_light_sample_n", self.d, self.n, d_queries, 48, 48, stream)

    def pdf_at_point(self, d_queries, stream=None):  # light.rs:132 -> one float per query
        return _launch("crt_light_pdf_n", self.d, self.n, d_queries, 48, 4, stream)

    def escaped(self, d_queries, stream=None):  # light.rs:141 -> LIGHT_SAMPLE records (radiance, pdf, some)
        return _launch("crt_light_escaped_n", self.d, self.n, d_queries, 48, 48, stream)
