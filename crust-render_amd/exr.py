"""Minimal OpenEXR I/O and the reference's image diff.

The reference writes its frames with `exr::prelude::write_rgb_file` (main.rs:598: three f32 channels, scanline
blocks, row 0 = top = buffer row height-1, buffer.rs:74-77) and compares two renders with
`examples/exr_diff.rs:45-86`. This module writes the same kind of file (uncompressed by default, which every EXR
reader accepts), reads single-part scanline files with NO / RLE / ZIPS / ZIP compression and FLOAT or HALF channels
(what the `exr` crate's writers produce), and restates exr_diff's metrics. Written from the OpenEXR file-layout
documentation; the `exr` crate is not available here, so interchange is unverified against it ("parity unpinned").
"""
import struct
import sys
import zlib

import numpy as np

MAGIC = 20000630
NO_COMPRESSION, RLE, ZIPS, ZIP = 0, 1, 2, 3
_LINES = {NO_COMPRESSION: 1, RLE: 1, ZIPS: 1, ZIP: 16}
HALF, FLOAT = 1, 2


def _attr(name, typ, data):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data


def _predict_interleave_encode(raw):
    a = np.frombuffer(raw, dtype=np.uint8)
    t = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)  # reorder: even bytes, then odd bytes
    d = t.copy()
    d[1:] = (t[1:] - t[:-1] + 128 + 256) % 256              # predictor
    return d.astype(np.uint8).tobytes()


def _predict_interleave_decode(buf):
    d = np.frombuffer(buf, dtype=np.uint8).astype(np.int64)
    if len(d) == 0:
        return b""
    t = (np.cumsum(np.concatenate([[d[0]], d[1:] - 128])) % 256).astype(np.uint8)  # t[i] = t[i-1] + d[i] - 128
    half = (len(t) + 1) // 2
    out = np.empty(len(t), dtype=np.uint8)
    out[0::2] = t[:half]
    out[1::2] = t[half:]
    return out.tobytes()


def _rle_encode(b):
    out, i, n = bytearray(), 0, len(b)
    while i < n:
        j = i + 1
        while j < n and b[j] == b[i] and j - i < 127:
            j += 1
        if j - i >= 3:  # run: count-1 (>= 0), value
            out += bytes([j - i - 1, b[i]])
            i = j
        else:  # literals until the next run of >= 3, at most 127
            j = i
            while j < n and j - i < 127 and not (j + 2 < n and b[j] == b[j + 1] == b[j + 2]):
                j += 1
            out += struct.pack("b", -(j - i)) + bytes(b[i:j])
            i = j
    return bytes(out)


def _rle_decode(b, expected):
    out, i = bytearray(), 0
    while i < len(b):
        c = struct.unpack_from("b", b, i)[0]
        i += 1
        if c < 0:
            out += b[i:i - c]
            i -= c
        else:
            out += bytes([b[i]]) * (c + 1)
            i += 1
    assert len(out) == expected, (len(out), expected)
    return bytes(out)


def write_exr(path, img, compression=NO_COMPRESSION):
    """img: [height, width, 3] float32, row 0 = TOP of the image (flip a film buffer with [::-1] first)."""
    img = np.ascontiguousarray(img, dtype=np.float32)
    h, w, _ = img.shape
    chans = b"".join(n + b"\0" + struct.pack("<iB3xii", FLOAT, 0, 1, 1) for n in (b"B", b"G", b"R")) + b"\0"
    box = struct.pack("<4i", 0, 0, w - 1, h - 1)
    header = (_attr("channels", "chlist", chans) + _attr("compression", "compression", bytes([compression])) +
              _attr("dataWindow", "box2i", box) + _attr("displayWindow", "box2i", box) +
              _attr("lineOrder", "lineOrder", b"\0") + _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) +
              _attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0)) +
              _attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    lines = _LINES[compression]
    blocks = []
    for y0 in range(0, h, lines):
        rows = img[y0:y0 + lines]
        raw = b"".join(rows[r, :, c].tobytes() for r in range(rows.shape[0]) for c in (2, 1, 0))  # B, G, R per line
        data = raw
        if compression == RLE:
            data = _rle_encode(_predict_interleave_encode(raw))
        elif compression in (ZIPS, ZIP):
            data = zlib.compress(_predict_interleave_encode(raw))
        if len(data) >= len(raw):
            data = raw  # the format stores a block uncompressed when compression does not shrink it
        blocks.append(struct.pack("<ii", y0, len(data)) + data)
    head = struct.pack("<ii", MAGIC, 2) + header
    table_at = len(head)
    offs, pos = [], table_at + 8 * len(blocks)
    for b in blocks:
        offs.append(pos)
        pos += len(b)
    with open(path, "wb") as f:
        f.write(head + struct.pack("<%dQ" % len(offs), *offs) + b"".join(blocks))


def read_exr(path):
    """-> [height, width, 3] float32 (R, G, B), row 0 = top. Single-part files: scanline blocks, or single-level
    tiles (the layout of the reference's own samples/sky_env.exr: 64x64 tiles, RLE, written by the `exr` crate)."""
    buf = open(path, "rb").read()
    magic, version = struct.unpack_from("<ii", buf, 0)
    if magic != MAGIC or (version & 0xff) != 2 or (version & 0x1800):
        raise ValueError("not a single-part OpenEXR file: %s" % path)
    tiled = bool(version & 0x200)
    pos, attrs = 8, {}
    while buf[pos] != 0:
        e = buf.index(b"\0", pos); name = buf[pos:e].decode(); pos = e + 1
        e = buf.index(b"\0", pos); typ = buf[pos:e].decode(); pos = e + 1
        n = struct.unpack_from("<i", buf, pos)[0]; pos += 4
        attrs[name] = (typ, buf[pos:pos + n]); pos += n
    pos += 1
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    comp = attrs["compression"][1][0]
    if comp not in _LINES:
        raise ValueError("unsupported EXR compression %d" % comp)
    chans, c, p = [], attrs["channels"][1], 0
    while c[p] != 0:
        e = c.index(b"\0", p); name = c[p:e].decode(); p = e + 1
        ptype, _lin, xs, ys = struct.unpack_from("<iB3xii", c, p); p += 16
        if (xs, ys) != (1, 1) or ptype not in (HALF, FLOAT):
            raise ValueError("unsupported channel %s" % name)
        chans.append((name, ptype))
    planes = {n: np.zeros((h, w), dtype=np.float32) for n, _ in chans}
    if tiled:
        tw, th, mode = struct.unpack("<IIB", attrs["tiles"][1])
        if mode & 0xf:
            raise ValueError("mip / rip mapped EXR tiles are not supported")
        nx, ny = (w + tw - 1) // tw, (h + th - 1) // th
        for off in struct.unpack_from("<%dQ" % (nx * ny), buf, pos):
            tx, ty, lx, ly, size = struct.unpack_from("<5i", buf, off)
            data = buf[off + 20:off + 20 + size]
            cols, rows = min(tw, w - tx * tw), min(th, h - ty * th)
            expected = rows * sum(cols * (2 if t == HALF else 4) for _, t in chans)
            if size != expected:
                packed = _rle_decode(data, expected) if comp == RLE else zlib.decompress(data)
                data = _predict_interleave_decode(packed)
            q = 0
            for r in range(rows):
                for name, t in chans:
                    planes[name][ty * th + r, tx * tw:tx * tw + cols] = np.frombuffer(
                        data, dtype=np.float16 if t == HALF else np.float32, count=cols, offset=q)
                    q += cols * (2 if t == HALF else 4)
        return np.stack([planes.get(k, np.zeros((h, w), np.float32)) for k in ("R", "G", "B")], axis=-1)
    line_bytes = sum(w * (2 if t == HALF else 4) for _, t in chans)
    lines = _LINES[comp]
    n_blocks = (h + lines - 1) // lines
    offs = struct.unpack_from("<%dQ" % n_blocks, buf, pos)
    for off in offs:
        y, size = struct.unpack_from("<ii", buf, off)
        data = buf[off + 8:off + 8 + size]
        rows = min(lines, h - (y - y0))
        expected = rows * line_bytes
        if size != expected:
            packed = _rle_decode(data, expected) if comp == RLE else zlib.decompress(data)
            data = _predict_interleave_decode(packed)
        q = 0
        for r in range(rows):
            for name, t in chans:
                nb = w * (2 if t == HALF else 4)
                planes[name][y - y0 + r] = np.frombuffer(data, dtype=np.float16 if t == HALF else np.float32, count=w, offset=q)
                q += nb
    return np.stack([planes.get(k, np.zeros((h, w), np.float32)) for k in ("R", "G", "B")], axis=-1)


def diff(a, b):
    """examples/exr_diff.rs:45-86 on two [h, w, 3] arrays: differing pixels, max abs, max rel, mean abs difference."""
    a, b = np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)
    if a.shape != b.shape:
        raise ValueError("resolutions differ")
    d = np.abs(a - b)
    scale = np.maximum(np.abs(a), np.abs(b))
    rel = np.where((d != 0) & (scale > 0), d / np.where(scale > 0, scale, 1), 0)
    return dict(differing_pixels=int((d != 0).any(axis=-1).sum()), total_pixels=int(a.shape[0] * a.shape[1]),
                max_abs=float(d.max(initial=0.0)), max_rel=float(rel.max(initial=0.0)),
                mean_abs=float(d.astype(np.float64).sum() / d.size))


if __name__ == "__main__":  # python crust-render_amd/exr.py a.exr b.exr
    if len(sys.argv) != 3:
        sys.exit("usage: exr.py <a.exr> <b.exr>")
    r = diff(read_exr(sys.argv[1]), read_exr(sys.argv[2]))
    print("differing pixels: %d/%d (%.4f%%)" % (r["differing_pixels"], r["total_pixels"], 100.0 * r["differing_pixels"] / r["total_pixels"]))
    print("max abs diff: %e   max rel diff: %e" % (r["max_abs"], r["max_rel"]))
    print("mean abs diff: %e" % r["mean_abs"])
