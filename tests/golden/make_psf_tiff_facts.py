"""Facts about the reference's own PSF stacks, read through THIS repository's TIFF reader.

The only data the reference holds for the path are the 18 measured PSF stacks `src/main/resources/Angle*.tif`, loaded by
`Tools.open(file, true)` (`Tools.java:297-307`: Bio-Formats read + `makeSquare`, `:315-349`) at
`SimulateMultiViewDataset.java:579` and `SimulateTileStitching.java:71`.  They are GPL data and stay where they are: this
script (build container only -- `/root/reference` does not exist on the GPU box) opens them with
`multiview-simulation_amd.tiffio.open_tiff` + `make_square` and writes FACTS, not pixels, to `tests/golden/psf_tiff_facts.json`:

  per file   size, sha256 of the file, byte order, IFD count, dims, first strip offset, ImageJ description keys,
             min / max / peak index, float64 sum (math.fsum), non-zero share, sha256 of the decoded little-endian float32
             array, second-moment sigmas per axis, rank-1 (separable) energy share,
             whether an INDEPENDENT decode (numpy on the raw strips, no tiffio) gives the same array,
             whether `tiffio.save_tiff(decoded, display_range from the description)` reproduces the file BYTE FOR BYTE
             (pins the writer = `Tools.save` / ImageJ's FileSaver layout on 18 real files), and whether makeSquare is the
             identity on it (the stacks are cubes already);
  overall    which files hold identical pixels (the survey counted 9 distinct contents).

`tests/test_host_logic.py::test_reference_psf_stacks_through_tiffio` re-derives all of it when `/root/reference` exists
and checks the JSON's self-consistency otherwise.

    python tests/golden/make_psf_tiff_facts.py            # rewrite the JSON
    python tests/golden/make_psf_tiff_facts.py --check    # compare with the committed JSON, exit 1 on a difference
"""
from __future__ import annotations

import glob
import hashlib
import importlib
import json
import math
import os
import re
import struct
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_DIR = "/root/reference/src/main/resources"
OUT = os.path.join(HERE, "psf_tiff_facts.json")


def _independent_decode(b: bytes):
    """No tiffio: walk the IFD chain by hand, take StripOffsets/StripByteCounts, decode with numpy."""
    bo = ">" if b[:2] == b"MM" else "<"
    off = struct.unpack_from(bo + "I", b, 4)[0]
    planes, ifds, first_strip, desc = [], 0, None, b""
    while off:
        n = struct.unpack_from(bo + "H", b, off)[0]
        ent = {}
        for i in range(n):
            tag, typ, cnt = struct.unpack_from(bo + "HHI", b, off + 2 + 12 * i)
            vo = off + 2 + 12 * i + 8
            ent[tag] = (typ, cnt, vo)
        def val(tag):
            typ, cnt, vo = ent[tag]
            fmt = {3: "H", 4: "I"}[typ]
            size = {3: 2, 4: 4}[typ] * cnt
            at = vo if size <= 4 else struct.unpack_from(bo + "I", b, vo)[0]
            return struct.unpack_from(bo + f"{cnt}{fmt}", b, at)
        w, h = val(256)[0], val(257)[0]
        so, sc = val(273), val(279)
        if first_strip is None:
            first_strip = so[0]
            if 270 in ent:
                typ, cnt, vo = ent[270]
                at = struct.unpack_from(bo + "I", b, vo)[0] if cnt > 4 else vo
                desc = b[at:at + cnt]
        raw = b"".join(b[o:o + c] for o, c in zip(so, sc))
        planes.append(np.frombuffer(raw, np.dtype(bo + "f4"), count=w * h).reshape(h, w))
        ifds += 1
        off = struct.unpack_from(bo + "I", b, off + 2 + 12 * n)[0]
    return np.stack(planes).astype(np.float32), bo, ifds, first_strip, desc


def _sigmas(a: np.ndarray):
    """Second-moment widths about the centroid, background (the stack's minimum) removed; order (x, y, z)."""
    w = a.astype(np.float64) - float(a.min())
    tot = w.sum()
    out = []
    for axis in (2, 1, 0):
        idx = np.arange(a.shape[axis], dtype=np.float64)
        prof = w.sum(axis=tuple(i for i in range(3) if i != axis))
        m = (prof * idx).sum() / tot
        out.append(math.sqrt(max((prof * (idx - m) ** 2).sum() / tot, 0.0)))
    return out


def _rank1_energy(a: np.ndarray) -> float:
    """Energy share of the best separable (x) x (y) x (z) approximation found by alternating power iterations (HOSVD start)."""
    t = a.astype(np.float64)
    u = [np.linalg.svd(np.moveaxis(t, ax, 0).reshape(t.shape[ax], -1), full_matrices=False)[0][:, 0] for ax in range(3)]
    for _ in range(20):
        u[0] = np.einsum("zyx,y,x->z", t, u[1], u[2]); u[0] /= np.linalg.norm(u[0])
        u[1] = np.einsum("zyx,z,x->y", t, u[0], u[2]); u[1] /= np.linalg.norm(u[1])
        u[2] = np.einsum("zyx,z,y->x", t, u[0], u[1]); u[2] /= np.linalg.norm(u[2])
    s = np.einsum("zyx,z,y,x->", t, u[0], u[1], u[2])
    return float(s * s / (t * t).sum())


def facts_of(path: str, tiffio) -> dict:
    b = open(path, "rb").read()
    img = tiffio.open_tiff(path)                       # the reader under test
    ind, bo, ifds, first_strip, desc = _independent_decode(b)
    keys = {}
    for line in desc.rstrip(b"\x00").split(b"\n"):
        if b"=" in line:
            k, v = line.split(b"=", 1)
            keys[k.decode()] = v.decode()
    sq = tiffio.make_square(img)
    peak = np.unravel_index(int(np.argmax(img)), img.shape)
    rewritten = None
    if "min" in keys and "max" in keys:
        with tempfile.NamedTemporaryFile(suffix=".tif", delete=False) as tf:
            tmp = tf.name
        try:
            tiffio.save_tiff(img, tmp, display_range=(float(keys["min"]), float(keys["max"])), imagej_version=keys.get("ImageJ", "1.48o"))
            rewritten = open(tmp, "rb").read() == b
        finally:
            os.remove(tmp)
    return {
        "file_bytes": len(b),
        "file_sha256": hashlib.sha256(b).hexdigest(),
        "byte_order": "big" if bo == ">" else "little",
        "ifds": ifds,
        "first_strip_offset": int(first_strip),
        "description": keys,
        "dims_xyz": [int(img.shape[2]), int(img.shape[1]), int(img.shape[0])],
        "min": float(img.min()),
        "max": float(img.max()),
        "peak_index_xyz": [int(peak[2]), int(peak[1]), int(peak[0])],
        "sum_f64": math.fsum(float(v) for v in img.ravel()),
        "nonzero_share": float(np.count_nonzero(img)) / img.size,
        "pixels_sha256": hashlib.sha256(np.ascontiguousarray(img, "<f4").tobytes()).hexdigest(),
        "sigma_xyz": [round(s, 6) for s in _sigmas(img)],
        "rank1_energy": round(_rank1_energy(img), 6),
        "independent_decode_equal": bool(np.array_equal(ind, img)),
        "save_tiff_reproduces_file": rewritten,
        "make_square_is_identity": bool(sq.shape == img.shape and np.array_equal(sq, img)),
    }


def collect() -> dict:
    sys.path.insert(0, ROOT)
    tiffio = importlib.import_module("multiview-simulation_amd.tiffio")
    files = sorted(glob.glob(os.path.join(REF_DIR, "Angle*.tif")), key=lambda p: int(re.search(r"Angle(\d+)", p).group(1)))
    per = {os.path.basename(p): facts_of(p, tiffio) for p in files}
    groups = {}
    for name, f in per.items():
        groups.setdefault(f["pixels_sha256"], []).append(name)
    synth = importlib.import_module("multiview-simulation_amd.synthetic")
    g = synth.measured_like_psf(51)
    stand_in = {
        "what": "multiview-simulation_amd.synthetic.measured_like_psf(51): the stack the tests and examples use in place of the files",
        "max": float(g.max()), "min": float(g.min()),
        "peak_index_xyz": [int(i) for i in np.unravel_index(int(np.argmax(g)), g.shape)[::-1]],
        "sum_f64": math.fsum(float(v) for v in g.ravel()),
        "nonzero_share": float(np.count_nonzero(g)) / g.size,
        "sigma_xyz": [round(s, 6) for s in _sigmas(g)],
        "rank1_energy": round(_rank1_energy(g), 6),
    }
    stat = lambda key, i=None: [f[key] if i is None else f[key][i] for f in per.values()]
    ranges = {
        "sum_f64": [min(stat("sum_f64")), max(stat("sum_f64"))],
        "nonzero_share": [min(stat("nonzero_share")), max(stat("nonzero_share"))],
        "rank1_energy": [min(stat("rank1_energy")), max(stat("rank1_energy"))],
        "sigma_x": [min(stat("sigma_xyz", 0)), max(stat("sigma_xyz", 0))],
        "sigma_y": [min(stat("sigma_xyz", 1)), max(stat("sigma_xyz", 1))],
        "sigma_z": [min(stat("sigma_xyz", 2)), max(stat("sigma_xyz", 2))],
    }
    return {
        "stand_in": stand_in,
        "ranges_over_the_files": ranges,
        "source": "src/main/resources/Angle*.tif of PreibischLab/multiview-simulation (GPL-2+; facts only, no pixel data)",
        "reader": "multiview-simulation_amd.tiffio.open_tiff + make_square (Tools.java:297-349)",
        "files": per,
        "identical_pixel_groups": sorted(groups.values(), key=lambda g: int(re.search(r"(\d+)", g[0]).group(1))),
        "distinct_contents": len(groups),
    }


def main() -> int:
    if not os.path.isdir(REF_DIR):
        print(f"{REF_DIR} not present: nothing to derive", file=sys.stderr)
        return 2
    rec = collect()
    if "--check" in sys.argv:
        old = json.load(open(OUT))
        if old != json.loads(json.dumps(rec)):
            print("psf_tiff_facts.json differs from what the reference's files give", file=sys.stderr)
            return 1
        print("psf_tiff_facts.json matches")
        return 0
    with open(OUT, "w") as fh:
        json.dump(rec, fh, indent=1, sort_keys=True)
        fh.write("\n")
    print(f"wrote {OUT}: {len(rec['files'])} files, {rec['distinct_contents']} distinct contents")
    return 0


if __name__ == "__main__":
    sys.exit(main())
