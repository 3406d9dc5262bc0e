"""TIFF stacks as the reference reads and writes them (host-side, no GPU involved).

    Tools.open   Tools.java:162-232  (Bio-Formats reader: float32 only, either byte order, one plane per z)
    Tools.save   Tools.java:88-105   (ImageJ FileSaver.saveAsTiffStack: big-endian float32 ImageJ stack)
    makeSquare   Tools.java:313-349  (pad to the largest dimension with the minimum intensity, centred)

The writer reproduces ImageJ's stack layout byte for byte (checked against the reference's shipped `Angle0.tif`):
header, first IFD at offset 8 (11 entries), the ImageJ description, all planes contiguously, then one IFD per
remaining plane.  The reader accepts that layout and ordinary multi-IFD uncompressed float32 TIFFs.
Images are numpy float32 arrays shaped (Nz, Ny, Nx), as everywhere in this package.
"""
from __future__ import annotations

import struct

import numpy as np

_TYPE_SIZE = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 6: 1, 7: 1, 8: 2, 9: 4, 10: 8, 11: 4, 12: 8, 16: 8}


def _read_ifd(b: bytes, off: int, bo: str):
    n = struct.unpack_from(bo + "H", b, off)[0]
    tags = {}
    for i in range(n):
        tag, typ, cnt, raw = struct.unpack_from(bo + "HHI4s", b, off + 2 + 12 * i)
        size = _TYPE_SIZE.get(typ, 1) * cnt
        data = raw[:size] if size <= 4 else b[struct.unpack(bo + "I", raw)[0]: struct.unpack(bo + "I", raw)[0] + size]
        if typ == 3:
            val = list(struct.unpack(bo + f"{cnt}H", data))
        elif typ == 4:
            val = list(struct.unpack(bo + f"{cnt}I", data))
        elif typ == 2:
            val = data
        else:
            val = data
        tags[tag] = val
    nxt = struct.unpack_from(bo + "I", b, off + 2 + 12 * n)[0]
    return tags, nxt


def _one(tags, tag, default=None):
    v = tags.get(tag)
    if v is None:
        return default
    return v[0] if isinstance(v, list) else v


def open_tiff(path: str) -> np.ndarray:
    """Tools.open: float32 stack -> (Nz, Ny, Nx).  Anything but 32-bit float samples is rejected (the reference
    prints "PixelType ... not supported" and returns null)."""
    b = open(path, "rb").read()
    if b[:2] == b"MM":
        bo = ">"
    elif b[:2] == b"II":
        bo = "<"
    else:
        raise ValueError(f"{path}: not a TIFF file")
    if struct.unpack_from(bo + "H", b, 2)[0] != 42:
        raise ValueError(f"{path}: not a classic TIFF (BigTIFF is not supported)")
    off = struct.unpack_from(bo + "I", b, 4)[0]
    planes = []
    first = None
    while off:
        tags, off = _read_ifd(b, off, bo)
        if first is None:
            first = tags
        w, h = _one(tags, 256), _one(tags, 257)
        bits, fmt = _one(tags, 258, 1), _one(tags, 339, 1)
        if _one(tags, 259, 1) != 1:
            raise ValueError(f"{path}: compressed TIFFs are not supported")
        if bits != 32 or fmt != 3 or _one(tags, 277, 1) != 1:
            raise ValueError(f"{path}: PixelType not supported (only single-channel 32-bit float)")
        offs, cnts = tags[273], tags.get(279)
        if cnts is None:
            cnts = [w * h * 4]
        planes.append((w, h, offs, cnts))
    w, h = planes[0][0], planes[0][1]
    n_images = len(planes)
    desc = first.get(270, b"")
    if isinstance(desc, bytes) and desc.startswith(b"ImageJ="):
        for line in desc.split(b"\n"):
            if line.startswith(b"images="):
                n_images = max(n_images, int(line[7:].strip(b"\x00 ")))
    dt = np.dtype(bo + "f4")
    if n_images > len(planes):
        # ImageJ writes stacks beyond 4 GB with a single IFD: the planes follow each other from the first strip
        start = planes[0][2][0]
        out = np.frombuffer(b, dt, count=n_images * w * h, offset=start).reshape(n_images, h, w)
        return out.astype(np.float32)
    out = np.empty((len(planes), h, w), np.float32)
    for z, (pw, ph, offs, cnts) in enumerate(planes):
        if (pw, ph) != (w, h):
            raise ValueError(f"{path}: planes of different sizes")
        raw = b"".join(b[o:o + c] for o, c in zip(offs, cnts))
        out[z] = np.frombuffer(raw, dt, count=w * h).reshape(h, w)
    return out


def _jdouble(v: float) -> str:
    """Double.toString for the values that show up as display ranges."""
    v = float(v)
    if v == int(v) and abs(v) < 1e7:
        return f"{int(v)}.0"
    return repr(v)


def save_tiff(img, path: str, display_range=None, imagej_version: str = "1.48o") -> None:
    """Tools.save: ImageJ big-endian float32 stack (3-D) or single image (2-D).  display_range: the (min, max)
    ImageJ records in the description; default = the data range."""
    a = np.asarray(img, dtype=np.float32)
    if a.ndim == 2:
        a = a[None]
    if a.ndim != 3:
        raise ValueError("expected a 2-D or 3-D image")
    nz, h, w = a.shape
    lo, hi = display_range if display_range is not None else (float(a.min()), float(a.max()))
    desc = f"ImageJ={imagej_version}\n"
    if nz > 1:
        desc += f"images={nz}\nslices={nz}\nloop=false\n"
    desc += f"min={_jdouble(lo)}\nmax={_jdouble(hi)}\n"
    desc_b = desc.encode("ascii") + b"\x00"
    n_ent = 11
    ifd_size = 2 + 12 * n_ent + 4
    desc_off = 8 + ifd_size
    data_off = desc_off + len(desc_b)
    plane = w * h * 4
    if data_off + nz * plane + (nz - 1) * ifd_size >= 1 << 32:
        raise ValueError("image too large for a classic TIFF")

    def ifd(strip_off: int, next_off: int) -> bytes:
        e = [(254, 4, 1, 0), (256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, 32 << 16), (262, 3, 1, 1 << 16),
             (270, 2, len(desc_b), desc_off), (273, 4, 1, strip_off), (277, 3, 1, 1 << 16), (278, 3, 1, h << 16),
             (279, 4, 1, plane), (339, 3, 1, 3 << 16)]
        out = struct.pack(">H", n_ent)
        for tag, typ, cnt, val in e:
            out += struct.pack(">HHII", tag, typ, cnt, val)
        return out + struct.pack(">I", next_off)

    tail = data_off + nz * plane
    with open(path, "wb") as f:
        f.write(b"MM\x00\x2a" + struct.pack(">I", 8))
        f.write(ifd(data_off, tail if nz > 1 else 0))
        f.write(desc_b)
        f.write(a.astype(">f4").tobytes())
        for z in range(1, nz):
            f.write(ifd(data_off + z * plane, tail + z * ifd_size if z + 1 < nz else 0))


def make_square(img) -> np.ndarray:
    """Tools.makeSquare: every dimension padded to the largest one with the minimum intensity; source voxel
    i lands at i + S/2 - N/2 (integer divisions as in the reference)."""
    a = np.asarray(img, dtype=np.float32)
    s = max(a.shape)
    out = np.full((s,) * a.ndim, a.min(), np.float32)
    sl = tuple(slice(s // 2 - n // 2, s // 2 - n // 2 + n) for n in a.shape)
    out[sl] = a
    return out
