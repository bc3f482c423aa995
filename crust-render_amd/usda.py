"""Minimal USDA (text USD) reader for the crust-render sample scenes.

Covers exactly what samples/{cornellbox,veach_mis,openpbr_showcase}.usda author: Xform / Mesh / Sphere /
Camera / SphereLight / RectLight / DistantLight / DomeLight / Scope / Material / Shader / RenderSettings prims, xformOp stacks,
`rel material:binding`, `crust:openpbr` shader inputs and the `crust:` render settings. It restates the
importer's decisions, not a USD composition engine (no references, payloads, variants or instancing):

    traversal order, chunked vs single-stage      scene/usd_import.rs:91-115, :139-269, :329-360
    xformOp composition                           :574-649
    mesh dedup -> bake (1 placement) / instance   :800-1100
    fan triangulation                             :1164-1214
    spheres, sphere/rect lights, light ray mask   :1234-1271, :2259-2349, :747-753
    camera                                        :2174-2220
    materials                                     :2566-2639, :2902-2980
    render settings + defaults                    :46-52, :3015-3100

The result is a neutral `SceneDesc` (numpy arrays + dicts); `build_world` feeds it to any SceneBuilder-shaped
API — the product's C ABI mirror or, in tests, the oracle — so both sides are built from identical inputs.
All arithmetic that reaches geometry is float32 in glam's operation order.
"""
import math
import re

import numpy as np

f32 = np.float32

MASK_CAMERA, MASK_SHADOW, MASK_INDIRECT, MASK_ALL = 1, 2, 4, 0xFFFFFFFF

DEFAULTS = dict(spp=128, max_depth=32, width=640, height=360, min_spp=32, variance=0.05, frame=0)  # :46-52


# ---------------------------------------------------------------------------------------------- parsing
class Prim:
    def __init__(self, spec, type_name, name):
        self.spec, self.type, self.name = spec, type_name, name
        self.attrs, self.rels, self.children, self.meta = {}, {}, [], {}

    def attr(self, name, default=None):
        return self.attrs.get(name, default)


_TOKEN = re.compile(r'''\s*(?:(\#[^\n]*)|("""(?:.|\n)*?""")|("(?:[^"\\]|\\.)*")|(<[^>]*>)|([\[\](){}=,;])|(@[^@]*@)|([^\s\[\](){}=,;"<>]+))''')


def _tokens(text):
    pos, out = 0, []
    n = len(text)
    while pos < n:
        m = _TOKEN.match(text, pos)
        if not m:
            if text[pos:].strip() == "":
                break
            raise ValueError(f"usda: cannot tokenise at {pos}: {text[pos:pos + 40]!r}")
        pos = m.end()
        if m.group(1) is not None:
            continue
        for k, kind in ((2, "str"), (3, "str"), (4, "path"), (5, "punct"), (6, "asset"), (7, "word")):
            if m.group(k) is not None:
                out.append((kind, m.group(k)))
                break
    return out


class _Parser:
    def __init__(self, text):
        self.t = _tokens(text)
        self.i = 0

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else (None, None)

    def next(self):
        tok = self.peek()
        self.i += 1
        return tok

    def expect(self, val):
        k, v = self.next()
        if v != val:
            raise ValueError(f"usda: expected {val!r}, got {v!r}")

    def skip_parens(self):
        """Skips a balanced ( ... ) metadata block."""
        self.expect("(")
        depth = 1
        while depth:
            _, v = self.next()
            if v is None:
                raise ValueError("usda: unbalanced metadata")
            depth += v == "("
            depth -= v == ")"

    def value(self):
        k, v = self.next()
        if v == "[":
            items = []
            while self.peek()[1] != "]":
                items.append(self.value())
                if self.peek()[1] == ",":
                    self.next()
            self.expect("]")
            return items
        if v == "(":
            items = []
            while self.peek()[1] != ")":
                items.append(self.value())
                if self.peek()[1] == ",":
                    self.next()
            self.expect(")")
            return tuple(items)
        if v == "{":  # dictionary-valued metadata (customData, assetInfo): typed entries `type name = value`
            d = {}
            while self.peek()[1] != "}":
                words = []
                while self.peek()[1] not in ("=", "}", None):
                    words.append(self.next()[1])
                if self.peek()[1] == "=":
                    self.next()
                    d[words[-1] if words else ""] = self.value()
                while self.peek()[1] in (",", ";"):
                    self.next()
            self.expect("}")
            return d
        if k == "str":
            return v[3:-3] if v.startswith('"""') else v.strip('"')
        if k in ("path", "asset"):
            return v[1:-1]
        if v in ("true", "True"):
            return True
        if v in ("false", "False"):
            return False
        try:
            return int(v)
        except ValueError:
            return float(v)

    def metadata(self):
        """A ( ... ) metadata block as a dict: `key = value` entries separated by newlines or `;`, list-op
        qualifiers (prepend / append / add / delete / reorder) dropped, bare doc strings skipped."""
        meta = {}
        if self.peek()[1] == "(":
            self.next()
            while self.peek()[1] != ")":
                k, v = self.next()
                if v is None:
                    raise ValueError("usda: unbalanced metadata")
                if k == "str" or v == ";":  # a bare doc string / separator
                    continue
                words = [v]
                while self.peek()[0] == "word" and self.peek()[1] is not None:
                    words.append(self.next()[1])
                if self.peek()[1] == "=":
                    self.next()
                    meta[words[-1]] = self.value()
            self.expect(")")
        return meta

    layer_meta = metadata

    def prims(self, closing=None):
        out = []
        while True:
            k, v = self.peek()
            if v is None or v == closing:
                return out
            if v in ("def", "over", "class"):
                out.append(self.prim())
            else:
                raise ValueError(f"usda: unexpected {v!r} at top of prim list")

    def prim(self):
        _, spec = self.next()
        k, v = self.next()
        if k == "str":
            type_name, name = "", v.strip('"')
        else:
            type_name, name = v, self.next()[1].strip('"')
        p = Prim(spec, type_name, name)
        p.meta = self.metadata()  # instanceable / references / active / apiSchemas / kind
        self.expect("{")
        while self.peek()[1] != "}":
            k, v = self.peek()
            if v in ("def", "over", "class"):
                p.children.append(self.prim())
                continue
            self.property(p)
        self.expect("}")
        return p

    def property(self, p):
        words = []
        while True:
            k, v = self.peek()
            if v in ("=", "}") or v is None or v in ("def", "over", "class"):
                break
            if v == "(":  # metadata after a declaration without value
                self.skip_parens()
                continue
            words.append(self.next()[1])
        if not words:
            raise ValueError("usda: empty property")
        name = words[-1]
        quals = words[:-1]
        val = None
        if self.peek()[1] == "=":
            self.next()
            val = self.value()
            if self.peek()[1] == "(":
                self.skip_parens()
        if "rel" in quals:
            p.rels[name] = val
        elif name.endswith(".connect"):
            p.attrs[name] = val
        else:
            p.attrs[name] = val


def parse(text):
    ps = _Parser(text)
    k, v = ps.peek()
    meta = ps.layer_meta()
    return meta, ps.prims()


# ---------------------------------------------------------------------------------------------- glam Mat4 (f32)
def _m4(rows_cols=None):
    return np.eye(4, dtype=np.float32) if rows_cols is None else np.asarray(rows_cols, dtype=np.float32)


def _mul(a, b):
    """glam Mat4 * Mat4 (column vectors): result column j = a.x*b[0,j] + a.y*b[1,j] + a.z*b[2,j] + a.w*b[3,j]."""
    out = np.zeros((4, 4), dtype=np.float32)
    for j in range(4):
        col = a[:, 0] * b[0, j]
        col = (col + a[:, 1] * b[1, j]).astype(np.float32)
        col = (col + a[:, 2] * b[2, j]).astype(np.float32)
        col = (col + a[:, 3] * b[3, j]).astype(np.float32)
        out[:, j] = col
    return out


def _det4(m):
    """Determinant of an affine Mat4 (last row 0 0 0 1) = det of its 3x3, evaluated in double."""
    return np.linalg.det(np.asarray(m, dtype=np.float64)[0:3, 0:3])


def _translation(v):
    m = _m4()
    m[0:3, 3] = np.asarray(v, dtype=np.float32)
    return m


def _scale(v):
    m = _m4()
    m[0, 0], m[1, 1], m[2, 2] = (f32(x) for x in v)
    return m


def _rot(axis, deg):
    a = f32(f32(deg) * f32(0.017453292519943295))  # f32::to_radians (x * (PI/180)), then glam sin_cos in f32
    s, c = f32(math.sin(a)), f32(math.cos(a))
    m = _m4()
    if axis == "x":
        m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, -s, s, c
    elif axis == "y":
        m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
    else:
        m[0, 0], m[0, 1], m[1, 0], m[1, 1] = c, -s, s, c
    return m


def _op_matrix(prim, name):  # usd_import.rs:604-649
    kind = name[len("xformOp:"):].split(":")[0]
    val = prim.attr(name)
    if val is None:
        return None
    if kind == "translate":
        return _translation(val)
    if kind == "scale":
        return _scale(val)
    if kind == "transform":
        # USD matrices are row-vector convention: rows of the authored matrix are glam's columns.
        return np.asarray(val, dtype=np.float64).astype(np.float32).T.copy()
    if kind in ("rotateX", "rotateY", "rotateZ"):
        return _rot(kind[-1].lower(), val)
    if kind.startswith("rotate") and len(kind) == 9:
        rx, ry, rz = _rot("x", val[0]), _rot("y", val[1]), _rot("z", val[2])
        order = {"rotateXYZ": (rz, ry, rx), "rotateXZY": (ry, rz, rx), "rotateYXZ": (rz, rx, ry),
                 "rotateYZX": (rx, rz, ry), "rotateZXY": (ry, rx, rz), "rotateZYX": (rx, ry, rz)}[kind]
        return _mul(_mul(order[0], order[1]), order[2])
    return None


def local_matrix(prim):  # compose_xform_ops, usd_import.rs:574-602
    order = prim.attr("xformOpOrder")
    if order is None:
        return _m4()
    local = _m4()
    for tok in order:
        if tok == "!resetXformStack!":
            continue
        inverted = tok.startswith("!invert!")
        name = tok[len("!invert!"):] if inverted else tok
        m = _op_matrix(prim, name)
        if m is None:
            return _m4()
        if inverted:
            m = np.linalg.inv(m.astype(np.float64)).astype(np.float32)
        local = _mul(local, m)
    return local


def _from_scale_rotation_translation(scale, q_xyzw, t):
    """glam Mat4::from_scale_rotation_translation with Quat::normalize first (usd_import.rs:1889, :1778-1784), f32."""
    x, y, z, w = (f32(v) for v in q_xyzw)
    inv = f32(1.0) / f32(np.sqrt(f32(f32(f32(x * x + y * y) + z * z) + w * w)))
    x, y, z, w = f32(x * inv), f32(y * inv), f32(z * inv), f32(w * inv)
    x2, y2, z2 = f32(x + x), f32(y + y), f32(z + z)
    xx, xy, xz = f32(x * x2), f32(x * y2), f32(x * z2)
    yy, yz, zz = f32(y * y2), f32(y * z2), f32(z * z2)
    wx, wy, wz = f32(w * x2), f32(w * y2), f32(w * z2)
    m = _m4()
    sx, sy, sz = (f32(v) for v in scale)
    m[0:3, 0] = np.array([f32(1.0) - f32(yy + zz), f32(xy + wz), f32(xz - wy)], dtype=np.float32) * sx
    m[0:3, 1] = np.array([f32(xy - wz), f32(1.0) - f32(xx + zz), f32(yz + wx)], dtype=np.float32) * sy
    m[0:3, 2] = np.array([f32(xz + wy), f32(yz - wx), f32(1.0) - f32(xx + yy)], dtype=np.float32) * sz
    m[0:3, 3] = np.asarray(t, dtype=np.float32)
    return m


def affine12(m4):
    """Affine3A::from_mat4 as the 12-float layout of the C ABI (columns x, y, z, translation)."""
    return np.concatenate([m4[0:3, 0], m4[0:3, 1], m4[0:3, 2], m4[0:3, 3]]).astype(np.float32)


def _xf_point(m4, v):  # Affine3A::transform_point3a: ((x*vx + y*vy) + z*vz) + t, per vertex, in f32
    v = np.asarray(v, dtype=np.float32).reshape(-1, 3)
    r = m4[0:3, 0][None, :] * v[:, 0:1]
    r = (r + m4[0:3, 1][None, :] * v[:, 1:2]).astype(np.float32)
    r = (r + m4[0:3, 2][None, :] * v[:, 2:3]).astype(np.float32)
    return (r + m4[0:3, 3][None, :]).astype(np.float32)


def _xf_vec(m4, v):
    v = np.asarray(v, dtype=np.float32)
    r = m4[0:3, 0] * v[0]
    r = (r + m4[0:3, 1] * v[1]).astype(np.float32)
    return (r + m4[0:3, 2] * v[2]).astype(np.float32)


def _normalize(v):
    v = np.asarray(v, dtype=np.float32)
    ln = np.sqrt(f32(f32(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]))
    return (v * (f32(1.0) / ln)).astype(np.float32)  # glam Vec3::normalize = v * length_recip


# ---------------------------------------------------------------------------------------------- scene description
OPENPBR_INPUTS = {  # decode_crust_openpbr, usd_import.rs:2902-2980: inputs:<camelCase> -> CrtMaterial field
    "baseWeight": "base_weight", "baseColor": "base_color", "baseDiffuseRoughness": "base_diffuse_roughness",
    "baseMetalness": "base_metalness", "specularWeight": "specular_weight", "specularColor": "specular_color",
    "specularRoughness": "specular_roughness", "specularIor": "specular_ior",
    "specularRoughnessAnisotropy": "specular_roughness_anisotropy", "transmissionWeight": "transmission_weight",
    "transmissionColor": "transmission_color", "transmissionDepth": "transmission_depth",
    "transmissionScatter": "transmission_scatter", "transmissionScatterAnisotropy": "transmission_scatter_anisotropy",
    "transmissionDispersionScale": "transmission_dispersion_scale",
    "transmissionDispersionAbbeNumber": "transmission_dispersion_abbe_number",
    "subsurfaceWeight": "subsurface_weight", "subsurfaceColor": "subsurface_color",
    "subsurfaceRadius": "subsurface_radius", "subsurfaceRadiusScale": "subsurface_radius_scale",
    "subsurfaceScatterAnisotropy": "subsurface_scatter_anisotropy", "fuzzWeight": "fuzz_weight",
    "fuzzColor": "fuzz_color", "fuzzRoughness": "fuzz_roughness", "coatWeight": "coat_weight",
    "coatColor": "coat_color", "coatRoughness": "coat_roughness", "coatRoughnessAnisotropy": "coat_roughness_anisotropy",
    "coatIor": "coat_ior", "coatDarkening": "coat_darkening", "thinFilmWeight": "thin_film_weight",
    "thinFilmThickness": "thin_film_thickness", "thinFilmIor": "thin_film_ior",
    "emissionLuminance": "emission_luminance", "emissionColor": "emission_color", "geometryOpacity": "geometry_opacity",
    "geometryThinWalled": "thin_walled",
}

DEFAULT_MATERIAL = {"_preset": "diffuse", "base_color": (0.5, 0.5, 0.5), "specular_weight": 0.0}  # :2637-2639


class SceneDesc:
    def __init__(self):
        self.geoms = []      # dicts: kind mesh|sphere|instance, mask, material (dict of overrides), + data
        self.protos = []     # shared local-space meshes: dict(verts, idx)
        self.lights = []     # dicts: kind, geom_id, radiance, center/radius or origin/edge_u/edge_v/normal
        self.camera = None   # dict(lookfrom, lookat, vup, vfov_deg, aspect, aperture, focus_dist)
        self.settings = dict(DEFAULTS, strategy="power", filter="triangle", filter_radius=1.0)


def _triangulate(counts, indices, n_verts):  # usd_import.rs:1164-1214
    tris, off = [], 0
    for fc in counts:
        if fc < 3 or off + fc > len(indices):
            off += fc
            continue
        for k in range(1, fc - 1):
            i0, i1, i2 = indices[off], indices[off + k], indices[off + k + 1]
            if min(i0, i1, i2) < 0 or max(i0, i1, i2) >= n_verts:
                continue
            tris.append((i0, i1, i2))
        off += fc
    return np.asarray(tris, dtype=np.uint32).reshape(-1, 3)


def _material_of(prim, by_path):
    target = prim.rels.get("material:binding")
    if isinstance(target, (list, tuple)):
        target = target[0] if target else None
    if not target:
        return DEFAULT_MATERIAL  # unbound prims (usd_import.rs:2544-2546)
    mat = by_path.get(target)
    if mat is None:
        return DEFAULT_MATERIAL
    shader = next((c for c in mat.children if c.type == "Shader"), None)
    if shader is not None and shader.attr("info:id") == "UsdPreviewSurface":
        # preview_surface_openpbr (usd_import.rs:2658-2700): OpenPBR::default() with the preview surface's AUTHORED inputs
        # mapped onto it. (What openusd's read_preview_surface reports for unauthored inputs is not in this container;
        # the one stage that uses this — scenes/stress.usda.xz — authors diffuseColor and roughness, and the spec's
        # defaults for the rest equal OpenPBR's except clearcoatRoughness 0.01 against coat_roughness 0 at coat weight 0.)
        out = {"_path": target}
        for usd_name, field in (("diffuseColor", "base_color"), ("metallic", "base_metalness"), ("roughness", "specular_roughness"),
                                ("opacity", "geometry_opacity"), ("ior", "specular_ior"), ("clearcoat", "coat_weight"),
                                ("clearcoatRoughness", "coat_roughness")):
            v = shader.attr("inputs:" + usd_name)
            if v is not None:
                out[field] = v
        e = shader.attr("inputs:emissiveColor")
        if e is not None:
            out["emission_color"] = e
            if max(float(x) for x in e) > 0.0:
                out["emission_luminance"] = 1.0
        return out
    if shader is None or shader.attr("info:id") != "crust:openpbr":
        return DEFAULT_MATERIAL  # no surface shader / unknown id -> default grey (usd_import.rs:2596-2631)
    out = {"_path": target}
    for usd_name, field in OPENPBR_INPUTS.items():
        v = shader.attr("inputs:" + usd_name)
        if v is not None:
            out[field] = v
    return out


def _ray_mask(prim):  # prim_ray_mask: authored crust:rayMask or ALL
    m = prim.attr("crust:rayMask")
    return MASK_ALL if m is None else int(m) & 0xFFFFFFFF


def _light_mask(prim):  # usd_import.rs:747-753
    m = prim.attr("crust:rayMask")
    if m is not None:
        return int(m) & 0xFFFFFFFF
    vis = bool(prim.attr("crust:light:cameraVisible", False))
    return MASK_SHADOW | MASK_INDIRECT | (MASK_CAMERA if vis else 0)


def _lux_emission(prim):  # usd_import.rs:2252-2258
    intensity = f32(prim.attr("inputs:intensity", 1.0))
    exposure = f32(prim.attr("inputs:exposure", 0.0))
    color = np.asarray(prim.attr("inputs:color", (1.0, 1.0, 1.0)), dtype=np.float32)
    gain = f32(intensity * f32(2.0 ** float(exposure)))
    return (color * gain).astype(np.float32)


def distant_light(direction, irradiance, angle_deg=0.53):
    """DistantLight::new (light.rs:255-266) in its derived form: unit travel direction, the cone's cos(half angle)
    and solid angle. Returns None for a degenerate direction (usd_import.rs:2362-2365)."""
    d = np.asarray(direction, dtype=np.float32)
    l2 = f32(f32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
    if l2 < f32(1e-12):
        return None
    d = (d / np.sqrt(l2)).astype(np.float32)
    diameter = np.clip(f32(angle_deg), f32(0.05), f32(179.0))
    half = f32(f32(0.5) * f32(diameter * f32(np.pi / 180.0)))
    cos_half = f32(np.cos(half, dtype=np.float32))
    omega = f32(f32(f32(2.0) * f32(np.pi)) * f32(f32(1.0) - cos_half))
    return dict(kind="distant", geom_id=0xFFFFFFFF, radiance=np.asarray(irradiance, dtype=np.float32), direction=d,
                cos_half_angle=cos_half, solid_angle=omega)


def dome_light(tint):
    """DomeLight without an environment map (light.rs:320-390): a uniform sky of radiance `tint`."""
    return dict(kind="dome", geom_id=0xFFFFFFFF, radiance=np.asarray(tint, dtype=np.float32))


def load(path, width=None, height=None):
    """Reads a .usda file into a SceneDesc. width/height override the RenderSettings resolution BEFORE the
    camera is built (the aspect ratio feeds Camera::new; the reference can only do this by editing the USD)."""
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:6] == b"\xfd7zXZ\x00":  # an .xz-compressed stage (scenes/stress.usda.xz: 15 MB of generated text, 0.7 MB packed)
        import lzma
        raw = lzma.decompress(raw)
    if raw[:8] == b"PXR-USDC":  # binary crate: same Prim trees from the crate reader
        from . import usdc
        meta, roots = usdc.parse(raw)
    else:
        meta, roots = parse(raw.decode("utf-8"))
    desc = SceneDesc()

    by_path = {}

    def index(p, prefix):
        path_ = prefix + "/" + p.name
        by_path[path_] = p
        p.path = path_
        for c in p.children:
            index(c, path_)
    for r in roots:
        index(r, "")

    # ---- render settings first: the camera needs the aspect ratio (usd_import.rs:308, :3015-3100) ----
    rs = by_path.get("/Render/settings")
    s = desc.settings
    if rs is not None and rs.type == "RenderSettings":
        res = rs.attr("resolution")
        if res is not None:
            s["width"], s["height"] = int(res[0]), int(res[1])
        for key, attr in (("spp", "crust:samplesPerPixel"), ("max_depth", "crust:maxDepth"),
                          ("min_spp", "crust:minSamplesPerPixel"), ("frame", "crust:frame")):
            if rs.attr(attr) is not None:
                s[key] = int(rs.attr(attr))
        if rs.attr("crust:varianceThreshold") is not None:
            s["variance"] = float(rs.attr("crust:varianceThreshold"))
        st = rs.attr("crust:samplingStrategy")
        s["strategy"] = {None: "power", "power": "power", "mis": "power", "balance": "balance", "light": "light",
                         "bsdf": "bsdf"}.get(st, "power")
        flt = rs.attr("crust:pixelFilter")
        if flt in ("box", "triangle"):
            s["filter"] = flt
            s["filter_radius"] = 0.5 if flt == "box" else 1.0
        elif flt is not None:
            raise NotImplementedError(f"pixel filter {flt!r}: only box and triangle are on the device path")
        if rs.attr("crust:pixelFilterRadius") is not None:
            s["filter_radius"] = max(float(rs.attr("crust:pixelFilterRadius")), 0.01)
    if width is not None:
        s["width"] = int(width)
    if height is not None:
        s["height"] = int(height)

    pending = []   # deferred mesh placements (usd_import.rs:986-1003)
    slots = []     # distinct meshes by content + material
    slot_by_key = {}
    proto_of_slot = {}   # slots that exist as kernel scenes of their own -> index into desc.protos
    sphere_protos = {}

    def intern_mesh(prim, mat):
        """MeshArena::intern (usd_import.rs:855-915): distinct meshes by content + material."""
        pts = prim.attr("points")
        counts, idx = prim.attr("faceVertexCounts"), prim.attr("faceVertexIndices")
        if pts is None or counts is None or idx is None:
            return None
        pts = np.asarray(pts, dtype=np.float32).reshape(-1, 3)
        counts, idx = np.asarray(counts), np.asarray(idx)
        key = (pts.tobytes(), counts.tobytes(), idx.tobytes(), mat.get("_path", id(mat)))
        slot = slot_by_key.get(key)
        if slot is None:
            tris = _triangulate(counts, idx, pts.shape[0])
            if tris.shape[0] == 0:
                return None
            slot = len(slots)
            slots.append(dict(verts=pts, idx=tris, n_place=0, committed=False))
            slot_by_key[key] = slot
        return slot

    def committed_proto(slot):
        """MeshArena::committed_scene: the slot as a kernel scene of its own (never baked afterwards)."""
        slots[slot]["committed"] = True
        if slot not in proto_of_slot:
            proto_of_slot[slot] = len(desc.protos)
            desc.protos.append(dict(verts=slots[slot]["verts"], idx=slots[slot]["idx"]))
        return proto_of_slot[slot]

    MAX_INSTANCE_NESTING = 8  # usd_import.rs:1347

    def is_active(prim):
        return prim.meta.get("active", True) is not False and prim.attr("active", True) is not False

    def ref_target(prim):
        """The internal reference a prim composes (`references = </Path>`), as a prim, or None. Only same-layer
        references are resolved: that is all the sample stages author."""
        ref = prim.meta.get("references")
        if isinstance(ref, (list, tuple)):
            ref = ref[0] if ref else None
        return by_path.get(ref) if isinstance(ref, str) and ref else None

    def is_instance(prim):
        """Prim::is_instance: `instanceable = true` together with a composition arc (usd_import.rs:165-170)."""
        return prim.meta.get("instanceable") is True and ref_target(prim) is not None

    def composed_children(prim):
        """The referenced prim's children come first (weaker opinions), then the prim's own."""
        tgt = ref_target(prim)
        return (list(tgt.children) if tgt is not None else []) + list(prim.children)

    proto_cache = {}

    def prototype_parts(root, depth):
        """prototype_parts (usd_import.rs:1808-1830): a prototype's parts, built once per path."""
        if root is None:
            return []
        if root.path not in proto_cache:
            proto_cache[root.path] = collect_proto_parts(root, depth)
        return proto_cache[root.path]

    def collect_proto_parts(root, depth=0):
        """collect_proto_parts (usd_import.rs:1379-1545): the leaf geometries under a prototype root, each with its
        root-relative placement, material and mask. The root's own transform is excluded; class prims are kept."""
        parts = []
        if depth > MAX_INSTANCE_NESTING:
            return parts
        stack = [(root, _m4())]
        while stack:
            prim, parent_local = stack.pop()
            if not is_active(prim):
                continue
            this_local = _m4() if prim is root else _mul(parent_local, local_matrix(prim))
            if prim is not root and is_instance(prim):
                # a natively-instanced prim inside a prototype is unreadable upstream and skipped with a warning
                # (usd_import.rs:1435-1443, pinned by usd_scene.rs nested_native_instance_degrades_gracefully)
                import warnings
                warnings.warn(f"nested native instance at {prim.path} skipped")
                continue
            if prim.type == "Mesh":
                mat = _material_of(prim, by_path)
                slot = intern_mesh(prim, mat)
                if slot is not None:
                    parts.append(dict(proto=committed_proto(slot), local=this_local, material=mat, mask=_ray_mask(prim)))
            elif prim.type == "Sphere":
                key = ("sphere", float(f32(prim.attr("radius", 1.0))))
                if key not in sphere_protos:
                    sphere_protos[key] = len(desc.protos)
                    desc.protos.append(dict(radius=key[1]))
                parts.append(dict(proto=sphere_protos[key], local=this_local, material=_material_of(prim, by_path),
                                  mask=_ray_mask(prim)))
            elif prim.type == "PointInstancer":
                parts.extend(nested_instancer_parts(prim, this_local, _ray_mask(prim), depth + 1))
                continue  # its prototypes are reached through it, never drawn directly
            for c in composed_children(prim):
                stack.append((c, this_local))
        return parts

    def nested_instancer_parts(prim, local, mask, depth):
        """nested_instancer_parts (usd_import.rs:1549-1606): a PointInstancer inside a prototype becomes one part per
        prototype-part of the nested instancer, each a committed sub-scene holding that part once per placement."""
        layout = read_instancer(prim)
        if layout is None:
            return []
        targets, placements = layout
        out = []
        for k, target in enumerate(targets):
            for part in prototype_parts(by_path.get(target), depth):
                inst = []
                for kk, xf in placements:
                    if kk != k:
                        continue
                    m = _mul(xf, part["local"])
                    if abs(float(_det4(m))) < 1e-12:
                        continue
                    inst.append(dict(proto=part["proto"], l2w=affine12(m), mask=part["mask"]))
                if not inst:
                    continue
                desc.protos.append(dict(instances=inst))
                out.append(dict(proto=len(desc.protos) - 1, local=local, material=part["material"], mask=mask))
        return out

    def read_instancer(prim):
        """read_instancer (usd_import.rs:1702-1792): translate * orient * scale per instance; orientationsf wins over
        orientations (half); invisibleIds prunes by ids (the array index where ids is absent). Returns
        (targets, [(index into targets, instancer-relative Mat4)])."""
        targets = prim.rels.get("prototypes") or []
        if isinstance(targets, str):
            targets = [targets]
        proto_indices = prim.attr("protoIndices")
        if not len(targets) or proto_indices is None:
            return None
        proto_indices = np.asarray(proto_indices).reshape(-1)
        positions = np.asarray(prim.attr("positions") if prim.attr("positions") is not None else np.zeros((0, 3)), dtype=np.float32).reshape(-1, 3)
        scales = prim.attr("scales")
        scales = None if scales is None else np.asarray(scales, dtype=np.float32).reshape(-1, 3)
        quats = prim.attr("orientationsf") if prim.attr("orientationsf") is not None else prim.attr("orientations")
        if quats is not None:
            quats = np.asarray(quats, dtype=np.float32).reshape(-1, 4)
            if not getattr(prim, "quats_xyzw", False):
                quats = quats[:, [1, 2, 3, 0]]  # text form is (w, x, y, z); the crate stores x, y, z, w
        ids = prim.attr("ids")
        invisible = set(int(x) for x in (prim.attr("invisibleIds") if prim.attr("invisibleIds") is not None else []))
        placements = []
        for i, k in enumerate(proto_indices):
            if i >= positions.shape[0]:
                break
            ident = int(ids[i]) if (ids is not None and i < len(ids)) else i
            if ident in invisible:
                continue
            k = int(k)
            if k < 0 or k >= len(targets):
                continue
            sc = scales[i] if (scales is not None and i < scales.shape[0]) else np.ones(3, dtype=np.float32)
            q = quats[i] if (quats is not None and i < quats.shape[0]) else np.array([0, 0, 0, 1], dtype=np.float32)
            placements.append((k, _from_scale_rotation_translation(sc, q, positions[i])))
        return list(targets), placements

    def attach_proto_parts(parts, placement, name):
        """attach_proto_parts (usd_import.rs:1637-1671): one instance per part; non-invertible placements skipped."""
        for part in parts:
            m = _mul(placement, part["local"])
            if abs(float(_det4(m))) < 1e-12:
                continue  # a zero scale hides an instance
            desc.geoms.append(dict(kind="instance", proto=part["proto"], l2w=affine12(m), mask=part["mask"],
                                   material=part["material"], name=name))

    def emit_point_instancer(prim, world_xf):
        """emit_point_instancer (usd_import.rs:1832-1876): every placement attaches its prototype's parts, composed
        under the instancer's world transform."""
        layout = read_instancer(prim)
        if layout is None:
            return
        targets, placements = layout
        parts_of = [prototype_parts(by_path.get(t), 0) for t in targets]
        for k, xf in placements:
            attach_proto_parts(parts_of[k], _mul(world_xf, xf), prim.name)

    def visit(prim, parent_world):
        """Returns the prim's world matrix, or None when its subtree is not traversed."""
        local = local_matrix(prim)
        world = _mul(parent_world, local)
        t = prim.type
        if is_instance(prim):  # emit_native_instance (usd_import.rs:165-189, :1616-1631): never descend into the proxy subtree
            attach_proto_parts(prototype_parts(ref_target(prim), 0), world, prim.name)
            return None
        if t == "PointInstancer":
            emit_point_instancer(prim, world)
            return None  # prototypes are drawn through the instancer, never on their own (usd_import.rs:205-207)
        if t == "Mesh":  # emit_mesh (usd_import.rs:912-1003)
            mat = _material_of(prim, by_path)
            motion = prim.attr("crust:motion:translate")
            if abs(float(_det4(world))) < 1e-12:  # non-invertible placement: baked on the spot, motion ignored
                pts, counts, idx = prim.attr("points"), prim.attr("faceVertexCounts"), prim.attr("faceVertexIndices")
                if pts is not None and counts is not None and idx is not None:
                    pts = np.asarray(pts, dtype=np.float32).reshape(-1, 3)
                    tris = _triangulate(np.asarray(counts), np.asarray(idx), pts.shape[0])
                    if tris.shape[0]:
                        desc.geoms.append(dict(kind="mesh", verts=_xf_point(world, pts), idx=tris, mask=_ray_mask(prim),
                                               material=mat, name=prim.name))
                return world
            slot = intern_mesh(prim, mat)
            if slot is not None:
                slots[slot]["n_place"] += 1
                gid = len(desc.geoms)
                desc.geoms.append(dict(kind="pending", mask=_ray_mask(prim), material=mat, name=prim.name))
                pending.append((gid, slot, world, motion))
        elif t == "Sphere":  # emit_sphere (usd_import.rs:1235-1271)
            radius = f32(prim.attr("radius", 1.0))
            center = _xf_point(world, [(0.0, 0.0, 0.0)])[0]
            motion = prim.attr("crust:motion:translate")
            if motion is not None:  # a moving sphere rides an identity-placed instance whose end transform is the shutter translation
                desc.protos.append(dict(radius=float(radius), center=center))
                desc.geoms.append(dict(kind="instance", proto=len(desc.protos) - 1, l2w=affine12(_m4()),
                                       l2w_end=affine12(_translation(motion)), mask=_ray_mask(prim),
                                       material=_material_of(prim, by_path), name=prim.name))
            else:
                desc.geoms.append(dict(kind="sphere", center=center, radius=radius, mask=_ray_mask(prim),
                                       material=_material_of(prim, by_path), name=prim.name))
        elif t == "Camera":
            if desc.camera is None:
                desc.camera = _camera(prim, world, s)
        elif t == "SphereLight":  # usd_import.rs:2259-2296
            radius = f32(prim.attr("inputs:radius", 0.5))
            rad = _lux_emission(prim)
            pos = _xf_point(world, [(0.0, 0.0, 0.0)])[0]
            gid = len(desc.geoms)
            desc.geoms.append(dict(kind="sphere", center=pos, radius=radius, mask=_light_mask(prim),
                                   material={"_preset": "emissive", "emission_color": tuple(rad)}, name=prim.name))
            desc.lights.append(dict(kind="sphere", geom_id=gid, radiance=rad, center=pos, radius=radius))
        elif t == "RectLight":  # usd_import.rs:2298-2349
            w, h = f32(prim.attr("inputs:width", 1.0)), f32(prim.attr("inputs:height", 1.0))
            rad = _lux_emission(prim)
            origin = _xf_point(world, [(f32(-0.5) * w, f32(-0.5) * h, 0.0)])[0]
            eu, ev = _xf_vec(world, (w, 0, 0)), _xf_vec(world, (0, h, 0))
            nz = _xf_vec(world, (0, 0, -1))
            verts = np.stack([origin, origin + eu, (origin + eu + ev).astype(np.float32), origin + ev]).astype(np.float32)
            gid = len(desc.geoms)
            desc.geoms.append(dict(kind="mesh", verts=verts, idx=np.array([(0, 1, 2), (0, 2, 3)], np.uint32),
                                   mask=_light_mask(prim), material={"_preset": "emissive", "emission_color": tuple(rad)},
                                   name=prim.name))
            nn = nz / np.sqrt(f32(f32(nz[0] * nz[0] + nz[1] * nz[1]) + nz[2] * nz[2]))  # Vec3A::normalize (RectShape::new)
            desc.lights.append(dict(kind="rect", geom_id=gid, radiance=rad, origin=origin, edge_u=eu, edge_v=ev,
                                    normal=nn.astype(np.float32)))
        elif t == "DistantLight":  # usd_import.rs:2360-2377: the light travels down its local -Z
            light = distant_light(_xf_vec(world, (0, 0, -1)), _lux_emission(prim), prim.attr("inputs:angle", 0.53))
            if light is not None:
                desc.lights.append(light)
        elif t == "DomeLight":  # usd_import.rs:2389-2460
            if prim.attr("inputs:texture:file", None) is not None:
                # No AssetLoader stands behind this importer: as when the reference's host declines to decode the
                # image (usd_import.rs:2419-2426), the dome falls back to its uniform colour.
                import warnings
                warnings.warn(f"DomeLight {prim.name}: environment map not decoded, using the uniform colour")
            desc.lights.append(dome_light(_lux_emission(prim)))
        return world

    def traverse(root_prims, root_world):
        """traverse_into: explicit LIFO stack — children are pushed in authored order and popped reversed."""
        stack = [(p, root_world) for p in root_prims]
        while stack:
            prim, parent_world = stack.pop()
            if prim.spec == "class" or not is_active(prim):
                continue
            world = visit(prim, parent_world)
            if world is None:
                continue
            for c in composed_children(prim):
                stack.append((c, world))

    identity = _m4()
    # stream_roots (usd_import.rs:91-115): children of the single top-level prim, else the top-level prims;
    # fewer than 4 chunks -> one pass over the whole stage.
    chunks = roots[0].children if len(roots) == 1 else roots
    if len(chunks) < 4:
        traverse(list(roots), identity)  # pseudo-root's children pushed in order, popped in reverse
    else:
        if len(roots) == 1:
            top = roots[0]
            top_world = _mul(identity, local_matrix(top))
            for ch in chunks:  # each chunk under its own masked stage, in authored order
                traverse([ch], top_world)
        else:
            for ch in chunks:
                traverse([ch], identity)

    # flush_meshes (usd_import.rs:1034-1089): sole placement -> baked into world space; else instanced
    for gid, slot, world, motion in pending:
        sl = slots[slot]
        g = desc.geoms[gid]
        m3 = world[0:3, 0:3].astype(np.float64)
        if sl["n_place"] == 1 and not sl["committed"] and motion is None:
            idx = sl["idx"].copy()
            if np.linalg.det(m3) < 0.0:  # bake_indices: mirrored placement swaps the winding
                idx[:, [1, 2]] = idx[:, [2, 1]]
            g.update(kind="mesh", verts=_xf_point(world, sl["verts"]), idx=idx)
        else:
            if slot not in proto_of_slot:
                proto_of_slot[slot] = len(desc.protos)
                desc.protos.append(dict(verts=sl["verts"], idx=sl["idx"]))
            g.update(kind="instance", proto=proto_of_slot[slot], l2w=affine12(world))
            if motion is not None:  # transform_end = from_translation(v) * l2w
                g["l2w_end"] = affine12(_mul(_translation(motion), world))
    if desc.camera is None:
        raise ValueError("usda: stage has no Camera prim")
    return desc


def _camera(prim, world, settings):  # build_camera, usd_import.rs:2174-2220
    lookfrom = _xf_point(world, [(0.0, 0.0, 0.0)])[0]
    forward = _normalize(_xf_vec(world, (0.0, 0.0, -1.0)))
    up = _normalize(_xf_vec(world, (0.0, 1.0, 0.0)))
    focal = f32(prim.attr("focalLength", 50.0))
    h_ap = f32(prim.attr("horizontalAperture", 20.955))
    v_ap_auth = prim.attr("verticalAperture")
    f_stop = f32(prim.attr("fStop", 0.0))
    focus = f32(prim.attr("focusDistance", 10.0))
    w_f, h_f = f32(settings["width"]), f32(settings["height"])
    v_ap = f32(v_ap_auth) if v_ap_auth is not None else f32(h_ap * h_f / w_f)
    half = f32(math.atan(float(f32(v_ap / f32(f32(2.0) * focal)))))  # (vert_aperture / (2 * focal)).atan()
    vfov_deg = f32(f32(2.0) * f32(half * f32(57.29577951308232)))         # 2.0 * x.to_degrees() (f32: x * (180/PI))
    aperture = f32(focal / f_stop) if f_stop > 0 else f32(0.0)
    lookat = (lookfrom + forward * focus).astype(np.float32)
    return dict(lookfrom=lookfrom, lookat=lookat, vup=up, vfov_deg=vfov_deg, aspect=f32(w_f / h_f), aperture=aperture,
                focus_dist=focus)


def fill_material(m, overrides):
    """Applies a material dict (from SceneDesc) onto a default-initialised CrtMaterial/OraMaterial ctypes struct."""
    preset = overrides.get("_preset")
    if preset == "emissive":
        m.kind = 1
        m.emission_luminance = 1.0
    for k, v in overrides.items():
        if k.startswith("_"):
            continue
        if k == "thin_walled":
            m.thin_walled = 1 if v else 0
        elif isinstance(v, (tuple, list, np.ndarray)):
            getattr(m, k)[:] = [float(f32(x)) for x in v]
        else:
            setattr(m, k, float(f32(v)))
    return m


def build_world(desc, api, new_material):
    """Feeds a SceneDesc through a SceneBuilder-shaped API (WorldBuilder::attach_masked + commit,
    rt_world.rs:111-185). Returns (scene, materials[list of ctypes structs], protos)."""
    protos = []
    for p in desc.protos:  # MeshArena::committed_scene, usd_import.rs:891-909; local sphere parts :1462-1484
        b = api.SceneBuilder()
        if "radius" in p:
            b.attach_sphere(p.get("center", (0.0, 0.0, 0.0)), float(p["radius"]))
        elif "instances" in p:  # a nested instancer's sub-scene: earlier protos placed once per nested instance
            for it in p["instances"]:
                b.attach_instance(protos[it["proto"]], it["l2w"], None, mask=it["mask"])
        else:
            b.attach_triangles(p["verts"], p["idx"])
        protos.append(b.commit())
    b = api.SceneBuilder()
    materials = []
    for g in desc.geoms:
        if g["kind"] == "mesh":
            b.attach_triangles(g["verts"], g["idx"], mask=g["mask"])
        elif g["kind"] == "sphere":
            b.attach_sphere(g["center"], float(g["radius"]), mask=g["mask"])
        elif g["kind"] == "instance":
            b.attach_instance(protos[g["proto"]], g["l2w"], g.get("l2w_end"), mask=g["mask"])
        else:
            b.attach_empty(mask=g["mask"])
        materials.append(fill_material(new_material(), g["material"]))
    return b.commit(), materials, protos
