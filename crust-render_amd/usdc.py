"""Minimal USDC ("crate") reader — SURVEY §8 f3 — for single-layer stages such as samples/PointInstancedMedCity.usd.

Written from the published crate layout (bootstrap "PXR-USDC" + version + table of contents; LZ4-framed TOKENS,
integer-coded FIELDS / FIELDSETS / PATHS / SPECS; 64-bit value representations), versions 0.4.0 … 0.10.x. It produces
the same `Prim` trees as the text reader (usda.parse), so the importer above it is shared. Not supported: composition
arcs (references, payloads, variants, sublayers), dictionaries beyond skipping them, string-valued list ops.

The reference reads crates through the third-party `openusd 0.6` crate, absent here: decoding is unpinned against it.
What pins it instead is self-consistency on the real file — every token, path, type name and array length must make
sense for the stage to import at all, the 40 000 instance orientations are unit quaternions that put every
prototype's up axis on the stage's up axis exactly (tests/test_usdc.py) — and round trips of the LZ4 and integer
codings through encoders kept with the tests. Attributes that carry time samples resolve to their FIRST sample, others to their default (the file at hand
authors every array as a single time sample at its one frame, next to empty defaults for some of them). The
reference reads default values only (`attr.get()`); what `openusd` returns there for this file is unknown.
"""
import struct

import numpy as np

from .usda import Prim

MAGIC = b"PXR-USDC"
_ARRAY, _INLINED, _COMPRESSED = 1 << 63, 1 << 62, 1 << 61
_PAYLOAD = (1 << 48) - 1
(T_BOOL, T_UCHAR, T_INT, T_UINT, T_INT64, T_UINT64, T_HALF, T_FLOAT, T_DOUBLE, T_STRING, T_TOKEN, T_ASSET, T_M2D, T_M3D,
 T_M4D, T_QUATD, T_QUATF, T_QUATH, T_V2D, T_V2F, T_V2H, T_V2I, T_V3D, T_V3F, T_V3H, T_V3I, T_V4D, T_V4F, T_V4H, T_V4I,
 T_DICT, T_TOKLISTOP, T_STRLISTOP, T_PATHLISTOP, T_REFLISTOP, T_INTLISTOP, T_I64LISTOP, T_UINTLISTOP, T_U64LISTOP,
 T_PATHVEC, T_TOKVEC, T_SPECIFIER, T_PERMISSION, T_VARIABILITY, T_VARSEL, T_TIMESAMPLES, T_PAYLOAD, T_DOUBLEVEC,
 T_LAYEROFFVEC, T_STRVEC, T_VALUEBLOCK, T_VALUE, T_UNREG, T_UNREGLISTOP, T_PAYLOADLISTOP, T_TIMECODE) = range(1, 57)
# element dtype, components
_NUM = {T_BOOL: ("u1", 1), T_UCHAR: ("u1", 1), T_INT: ("<i4", 1), T_UINT: ("<u4", 1), T_INT64: ("<i8", 1),
        T_UINT64: ("<u8", 1), T_HALF: ("<f2", 1), T_FLOAT: ("<f4", 1), T_DOUBLE: ("<f8", 1), T_TIMECODE: ("<f8", 1),
        T_M2D: ("<f8", 4), T_M3D: ("<f8", 9), T_M4D: ("<f8", 16), T_QUATD: ("<f8", 4), T_QUATF: ("<f4", 4),
        T_QUATH: ("<f2", 4), T_V2D: ("<f8", 2), T_V2F: ("<f4", 2), T_V2H: ("<f2", 2), T_V2I: ("<i4", 2),
        T_V3D: ("<f8", 3), T_V3F: ("<f4", 3), T_V3H: ("<f2", 3), T_V3I: ("<i4", 3), T_V4D: ("<f8", 4),
        T_V4F: ("<f4", 4), T_V4H: ("<f2", 4), T_V4I: ("<i4", 4)}
SPEC_ATTRIBUTE, SPEC_PRIM, SPEC_PSEUDOROOT, SPEC_RELATIONSHIP = 1, 6, 7, 8


def lz4_block(src):
    """One raw LZ4 block (sequence of literal runs and back references)."""
    out, i, n = bytearray(), 0, len(src)
    while i < n:
        tok = src[i]; i += 1
        lit = tok >> 4
        if lit == 15:
            while True:
                x = src[i]; i += 1; lit += x
                if x != 255:
                    break
        out += src[i:i + lit]; i += lit
        if i >= n:
            break
        off = src[i] | (src[i + 1] << 8); i += 2
        ml = tok & 15
        if ml == 15:
            while True:
                x = src[i]; i += 1; ml += x
                if x != 255:
                    break
        ml += 4
        st = len(out) - off
        if off <= 0 or st < 0:
            raise ValueError("usdc: corrupt LZ4 stream")
        if off >= ml:
            out += out[st:st + ml]
        else:
            for k in range(ml):
                out.append(out[st + k])
    return bytes(out)


def fast_decompress(src):
    """TfFastCompression framing: a chunk-count byte (0 = one block), then [int32 size][block] per chunk."""
    if src[0] == 0:
        return lz4_block(src[1:])
    out, p = b"", 1
    for _ in range(src[0]):
        sz = struct.unpack_from("<i", src, p)[0]; p += 4
        out += lz4_block(src[p:p + sz]); p += sz
    return out


def decode_ints(buf, n, width=4):
    """Usd_IntegerCompression: most common delta + a 2-bit code per element (common / int8 / int16 / int32; for
    64-bit: common / int16 / int32 / int64), deltas against the running value, the whole thing LZ4-framed."""
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    enc = fast_decompress(buf)
    common = struct.unpack_from("<i" if width == 4 else "<q", enc, 0)[0]
    p = width + (n * 2 + 7) // 8
    codes = np.frombuffer(enc, dtype=np.uint8, count=(n * 2 + 7) // 8, offset=width)
    code = (np.repeat(codes, 4)[:n] >> (np.tile(np.arange(4, dtype=np.uint8) * 2, (n + 3) // 4)[:n])) & 3
    sizes = np.array([0, 1, 2, 4] if width == 4 else [0, 2, 4, 8])[code]
    offs = p + np.concatenate([[0], np.cumsum(sizes)[:-1]])
    deltas = np.full(n, common, dtype=np.int64)
    for c, fmt in ((1, "<i1"), (2, "<i2"), (3, "<i4")) if width == 4 else ((1, "<i2"), (2, "<i4"), (3, "<i8")):
        sel = np.nonzero(code == c)[0]
        if sel.size:
            w = np.dtype(fmt).itemsize
            raw = np.frombuffer(enc, dtype=np.uint8)
            idx = offs[sel][:, None] + np.arange(w)[None, :]
            deltas[sel] = raw[idx].copy().view(fmt).reshape(-1)
    vals = np.cumsum(deltas.astype(np.uint64))  # running value, wrapping like the unsigned type it is stored in
    if width == 4:
        vals &= np.uint64(0xFFFFFFFF)
        return vals.astype(np.int64)
    return vals  # uint64


def _signed(v, bits=32):
    if bits == 64:
        return np.asarray(v).astype(np.uint64).view(np.int64)
    v = np.asarray(v, dtype=np.int64)
    return np.where(v >= (1 << (bits - 1)), v - (1 << bits), v)


class Crate:
    def __init__(self, data):
        if data[:8] != MAGIC:
            raise ValueError("not a USDC crate")
        self.b = data
        self.version = tuple(data[8:11])
        if not ((0, 4, 0) <= self.version < (0, 11, 0)):
            raise NotImplementedError("usdc: crate version %d.%d.%d" % self.version)
        toc = struct.unpack_from("<q", data, 16)[0]
        n = struct.unpack_from("<Q", data, toc)[0]
        self.sec = {}
        for i in range(n):
            name, start, size = struct.unpack_from("<16sqq", data, toc + 8 + 32 * i)
            self.sec[name.rstrip(b"\0").decode()] = (start, size)
        self._tokens(); self._strings(); self._fields(); self._fieldsets(); self._paths(); self._specs()

    # ---- sections ----
    def _csection(self, p, n, width=4):
        cs = struct.unpack_from("<Q", self.b, p)[0]
        return decode_ints(self.b[p + 8:p + 8 + cs], n, width), p + 8 + cs

    def _tokens(self):
        s, _ = self.sec["TOKENS"]
        n, _usize, csize = struct.unpack_from("<QQQ", self.b, s)
        self.tokens = [t.decode("utf-8", "replace") for t in fast_decompress(self.b[s + 24:s + 24 + csize]).split(b"\0")[:n]]

    def _strings(self):
        s, _ = self.sec["STRINGS"]
        n = struct.unpack_from("<Q", self.b, s)[0]
        self.strings = list(struct.unpack_from("<%dI" % n, self.b, s + 8))

    def _fields(self):
        s, _ = self.sec["FIELDS"]
        n = struct.unpack_from("<Q", self.b, s)[0]
        self.field_tok, p = self._csection(s + 8, n)
        cs = struct.unpack_from("<Q", self.b, p)[0]
        self.field_rep = struct.unpack("<%dQ" % n, fast_decompress(self.b[p + 8:p + 8 + cs]))

    def _fieldsets(self):
        s, _ = self.sec["FIELDSETS"]
        n = struct.unpack_from("<Q", self.b, s)[0]
        self.fieldsets, _ = self._csection(s + 8, n)

    def _paths(self):
        s, _ = self.sec["PATHS"]
        n_paths, n_enc = struct.unpack_from("<QQ", self.b, s)
        p = s + 16
        pidx, p = self._csection(p, n_enc)
        etok, p = self._csection(p, n_enc)
        jumps, p = self._csection(p, n_enc)
        etok, jumps = _signed(etok), _signed(jumps)
        self.paths = [None] * n_paths   # (parent path index or -1, element name, is_property)
        self.path_str = [None] * n_paths
        stack = [(0, -1)]               # (encoded index to start at, parent path index)
        while stack:
            cur, parent = stack.pop()
            while True:
                this = cur; cur += 1
                me = int(pidx[this])
                if parent < 0:
                    self.paths[me] = (-1, "", False); self.path_str[me] = "/"
                else:
                    t = int(etok[this])
                    name, is_prop = self.tokens[abs(t)], t < 0
                    self.paths[me] = (parent, name, is_prop)
                    base = self.path_str[parent]
                    self.path_str[me] = base + ("." if is_prop else ("" if base.endswith("/") else "/")) + name
                j = int(jumps[this])
                has_child, has_sib = j > 0 or j == -1, j >= 0
                if has_child:
                    if has_sib:
                        stack.append((this + j, parent))
                    parent = me
                elif not has_sib:
                    break

    def _specs(self):
        s, _ = self.sec["SPECS"]
        n = struct.unpack_from("<Q", self.b, s)[0]
        p = s + 8
        self.spec_path, p = self._csection(p, n)
        self.spec_fset, p = self._csection(p, n)
        self.spec_type, p = self._csection(p, n)

    # ---- values ----
    def fields_of(self, spec):
        out, j = {}, int(self.spec_fset[spec])
        while int(self.fieldsets[j]) != 0xFFFFFFFF:
            fi = int(self.fieldsets[j])
            out[self.tokens[int(self.field_tok[fi])]] = self.field_rep[fi]
            j += 1
        return out

    def _array(self, ty, rep):
        off = rep & _PAYLOAD
        if off == 0:
            return np.zeros((0,), dtype=np.float32)
        b = self.b
        if self.version < (0, 7, 0):
            n = struct.unpack_from("<I", b, off + (4 if self.version < (0, 5, 0) else 0))[0]
            p = off + (8 if self.version < (0, 5, 0) else 4)
        else:
            n = struct.unpack_from("<Q", b, off)[0]; p = off + 8
        if ty == T_TOKEN:
            return [self.tokens[i] for i in struct.unpack_from("<%dI" % n, b, p)]
        if ty == T_STRING or ty == T_ASSET:
            idx = struct.unpack_from("<%dI" % n, b, p)
            return [self.tokens[self.strings[i]] if ty == T_STRING else self.tokens[i] for i in idx]
        dt, comps = _NUM[ty]
        if rep & _COMPRESSED:
            if ty in (T_INT, T_UINT, T_INT64, T_UINT64):
                w = 4 if ty in (T_INT, T_UINT) else 8
                cs = struct.unpack_from("<Q", b, p)[0]
                v = decode_ints(b[p + 8:p + 8 + cs], n, w)
                return (_signed(v, 8 * w) if ty in (T_INT, T_INT64) else v).astype(np.dtype(dt).newbyteorder("="))
            if ty in (T_HALF, T_FLOAT, T_DOUBLE):
                code = b[p:p + 1]
                if code == b"i":   # values that are all integers, stored as compressed int32
                    cs = struct.unpack_from("<Q", b, p + 1)[0]
                    return _signed(decode_ints(b[p + 9:p + 9 + cs], n)).astype(np.dtype(dt).newbyteorder("="))
                if code == b"t":   # lookup table + compressed indices
                    lut_n = struct.unpack_from("<I", b, p + 1)[0]
                    lut = np.frombuffer(b, dtype=dt, count=lut_n, offset=p + 5)
                    q = p + 5 + lut_n * np.dtype(dt).itemsize
                    cs = struct.unpack_from("<Q", b, q)[0]
                    return lut[decode_ints(b[q + 8:q + 8 + cs], n)].astype(np.dtype(dt).newbyteorder("="))
                raise ValueError("usdc: unknown float array code %r" % code)
            raise NotImplementedError("usdc: compressed array of type %d" % ty)
        a = np.frombuffer(b, dtype=dt, count=n * comps, offset=p)
        return a.reshape(n, comps) if comps > 1 else a

    def value(self, rep):
        ty = (rep >> 48) & 0xFF
        pay = rep & _PAYLOAD
        b = self.b
        if rep & _ARRAY:
            return self._array(ty, rep)
        if rep & _INLINED:
            if ty == T_BOOL:
                return bool(pay & 1)
            if ty in (T_INT, T_UINT, T_UCHAR):
                v = pay & 0xFFFFFFFF
                return int(v - (1 << 32)) if (ty == T_INT and v >= (1 << 31)) else int(v)
            if ty in (T_INT64, T_UINT64):   # stored as int32 when it fits
                return int(_signed(pay & 0xFFFFFFFF))
            if ty == T_FLOAT:
                return float(np.array([pay & 0xFFFFFFFF], dtype=np.uint32).view(np.float32)[0])
            if ty in (T_DOUBLE, T_TIMECODE):  # inlined doubles are stored as the float32 that equals them
                return float(np.array([pay & 0xFFFFFFFF], dtype=np.uint32).view(np.float32)[0])
            if ty == T_HALF:
                return float(np.array([pay & 0xFFFF], dtype=np.uint16).view(np.float16)[0])
            if ty == T_TOKEN:
                return self.tokens[pay]
            if ty == T_STRING:
                return self.tokens[self.strings[pay]]
            if ty == T_ASSET:
                return self.tokens[pay]
            if ty in (T_SPECIFIER, T_PERMISSION, T_VARIABILITY):
                return int(pay)
            if ty in _NUM:  # small-integer vectors / diagonal matrices packed as int8 components
                dt, comps = _NUM[ty]
                raw = np.frombuffer(struct.pack("<Q", pay), dtype=np.int8)
                if ty in (T_M2D, T_M3D, T_M4D):
                    k = {T_M2D: 2, T_M3D: 3, T_M4D: 4}[ty]
                    return np.diag(raw[:k].astype(np.float64))
                return raw[:comps].astype(np.dtype(dt).newbyteorder("="))
            if ty == T_VALUEBLOCK:
                return None
            if ty == T_DICT:
                return {}
            raise NotImplementedError("usdc: inlined value of type %d" % ty)
        if ty in _NUM:
            dt, comps = _NUM[ty]
            a = np.frombuffer(b, dtype=dt, count=comps, offset=pay)
            if ty in (T_M2D, T_M3D, T_M4D):
                k = {T_M2D: 2, T_M3D: 3, T_M4D: 4}[ty]
                return a.reshape(k, k).copy()
            return a[0].item() if comps == 1 else a.copy()
        if ty == T_TOKVEC:
            n = struct.unpack_from("<Q", b, pay)[0]
            return [self.tokens[i] for i in struct.unpack_from("<%dI" % n, b, pay + 8)]
        if ty == T_PATHVEC:
            n = struct.unpack_from("<Q", b, pay)[0]
            return [self.path_str[i] for i in struct.unpack_from("<%dI" % n, b, pay + 8)]
        if ty == T_DOUBLEVEC:
            n = struct.unpack_from("<Q", b, pay)[0]
            return np.frombuffer(b, dtype="<f8", count=n, offset=pay + 8).copy()
        if ty == T_PATHLISTOP or ty == T_TOKLISTOP:
            hdr = b[pay]
            p, lists = pay + 1, {}
            for bit, key in ((2, "explicit"), (4, "added"), (32, "prepended"), (64, "appended"), (8, "deleted"), (16, "ordered")):
                if hdr & bit:
                    n = struct.unpack_from("<Q", b, p)[0]
                    idx = struct.unpack_from("<%dI" % n, b, p + 8)
                    lists[key] = [self.path_str[i] if ty == T_PATHLISTOP else self.tokens[i] for i in idx]
                    p += 8 + 4 * n
            # the composed result of a single layer's list op: explicit, else prepended + added + appended
            return lists.get("explicit", lists.get("prepended", []) + lists.get("added", []) + lists.get("appended", []))
        if ty == T_TIMESAMPLES:
            p = pay
            p += struct.unpack_from("<q", b, p)[0]
            times = self.value(struct.unpack_from("<Q", b, p)[0])
            p += 8
            p += struct.unpack_from("<q", b, p)[0]
            n = struct.unpack_from("<Q", b, p)[0]
            reps = struct.unpack_from("<%dQ" % n, b, p + 8)
            return TimeSamples(np.atleast_1d(np.asarray(times, dtype=np.float64)), reps, self)
        if ty == T_DICT:
            return {}  # asset info and the like: nothing the importer reads
        if ty in (T_STRING, T_ASSET, T_TOKEN):
            return self.tokens[pay]
        raise NotImplementedError("usdc: value of type %d" % ty)


class TimeSamples:
    def __init__(self, times, reps, crate):
        self.times, self._reps, self._crate = times, reps, crate

    def first(self):
        return self._crate.value(self._reps[0]) if len(self._reps) else None


def _plain(v):
    """The Python shape the text reader produces for the same value."""
    if isinstance(v, np.ndarray):
        if v.dtype.kind == "f" and v.dtype.itemsize == 2:
            v = v.astype(np.float32)
        return v
    return v


def parse(data):
    """-> (layer metadata dict, [root Prims]) exactly as usda.parse: Prim.attrs hold resolved default values (or the
    first time sample), Prim.rels the composed target paths, children in `primChildren` order."""
    c = Crate(data)
    by_path = {c.path_str[int(p)]: k for k, p in enumerate(c.spec_path)}
    spec_types = {c.path_str[int(p)]: int(t) for p, t in zip(c.spec_path, c.spec_type)}

    def resolve(fields):
        # The stage is read at its first authored time sample where an attribute has any, else at its default.
        # (The file this reader exists for authors EVERY array as one time sample at its single frame and leaves
        # empty defaults next to some of them; reading defaults only would import an empty instancer.)
        if "timeSamples" in fields:
            v = c.value(fields["timeSamples"]).first()
            if v is not None:
                return _plain(v)
        if "default" in fields:
            return _plain(c.value(fields["default"]))
        return None

    def build(path):
        f = c.fields_of(by_path[path])
        spec = {0: "def", 1: "over", 2: "class"}[c.value(f["specifier"])] if "specifier" in f else "over"
        prim = Prim(spec, c.value(f["typeName"]) if "typeName" in f else "", path.rsplit("/", 1)[1])
        prim.quats_xyzw = True  # crate quaternions are stored imaginary part first; the text form is (w, x, y, z)
        for key in ("active", "kind", "instanceable"):
            if key in f:
                prim.meta[key] = c.value(f[key])
        if "active" in f:
            prim.attrs["active"] = prim.meta["active"]
        if "apiSchemas" in f:
            prim.meta["apiSchemas"] = c.value(f["apiSchemas"])
        for name in (c.value(f["properties"]) if "properties" in f else []):
            ppath = path + "." + name
            if ppath not in by_path:
                continue
            pf = c.fields_of(by_path[ppath])
            if spec_types[ppath] == SPEC_RELATIONSHIP:
                prim.rels[name] = c.value(pf["targetPaths"]) if "targetPaths" in pf else []
            else:
                v = resolve(pf)
                if v is not None:
                    prim.attrs[name] = v
        for child in (c.value(f["primChildren"]) if "primChildren" in f else []):
            cpath = (path if path != "/" else "") + "/" + child
            if cpath in by_path:
                prim.children.append(build(cpath))
        return prim

    root_fields = c.fields_of(by_path["/"])
    meta = {}
    for k, rep in root_fields.items():
        if k not in ("primChildren",):
            try:
                meta[k] = c.value(rep)
            except NotImplementedError:
                pass
    roots = [build("/" + n) for n in (c.value(root_fields["primChildren"]) if "primChildren" in root_fields else [])
             if "/" + n in by_path]
    return meta, roots
