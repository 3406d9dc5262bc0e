"""Host-side logic and the C-ABI surface -- runs without a GPU (no compute calls)."""
import ctypes
import importlib
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mvsim.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mvsim_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(mvs):
    lib = ctypes.CDLL(mvs._lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mvsim.h but not exported"
    # and the ctypes table covers the whole header (no silently unbound entry point)
    assert set(names) == set(mvs._lib.SIGNATURES)


def test_version_and_error_channel(mvs):
    assert "gfx950" in mvs.version()
    L = mvs._lib.load()
    assert L.mvsim_extract_nz(512, 3) == 171 and L.mvsim_extract_nz(512, 4) == 128 and L.mvsim_extract_nz(512, 1) == 512
    assert L.mvsim_extract_nz(10, 0) == -1
    assert L.mvsim_isotropic_nz(171, 3) == 511
    assert L.mvsim_poisson_mul(25.0) == 124.99999999999997
    # EINVAL path sets the thread-local message and maps to ValueError (Java: IllegalArgumentException)
    with pytest.raises(ValueError, match="axis"):
        mvs.SimulateMultiViewDataset.axisRotation((8, 8, 8), 3, 10)
    with pytest.raises(ValueError):
        mvs.shard_views(8, 0, 0)


def test_axis_rotation_matches_oracle(mvs, orc):
    for dims in ((512, 512, 512), (289, 289, 289), (10, 14, 12)):
        for axis in (0, 1, 2):
            for deg in (0, 15, 60, 90, -52, 327):
                a = mvs.SimulateMultiViewDataset.axisRotation(dims, axis, deg)
                b = orc.axis_rotation(dims, axis, deg)
                assert np.array_equal(a, b), (dims, axis, deg)


def test_java_random_mirror_matches_jdk_vectors(mvs, orc):
    r = mvs.JavaRandom(464232194)
    assert [r.nextDouble() for _ in range(4)] == [0.4143130143281428, 0.9731632560980291, 0.6356592534797139,
                                                  0.45751024578762167]
    r = mvs.JavaRandom(464232194)
    assert [r.nextInt(20) for _ in range(6)] == [14, 6, 19, 4, 11, 9]
    assert mvs.JavaRandom(0).nextInt() == -1155484576
    a, b = mvs.JavaRandom(12345), orc.JRandom(12345)
    for _ in range(50):
        assert a.nextLong() == b.nextLong()
        assert a.nextInt(1000) == b.nextInt(1000)
        assert a.nextInt(64) == b.nextInt(64)


def test_shard_views(mvs):
    assert mvs.shard_views(8, 1, 0) == list(range(8))
    assert mvs.shard_views(8, 8, 3) == [3]
    assert mvs.shard_views(6, 4, 1) == [1, 5] and mvs.shard_views(6, 4, 3) == [3]
    allv = sorted(v for r in range(4) for v in mvs.shard_views(12, 4, r))
    assert allv == list(range(12))
    assert mvs.shard_views(0, 2, 1) == []


def test_synthetic_generators(synth):
    v = synth.sphere_phantom(32)
    assert v.shape == (32, 32, 32) and v.dtype == np.float32 and v.min() == 0 and 0 < v.max() < 1
    assert np.array_equal(v, synth.sphere_phantom(32))
    p = synth.gaussian_psf(15, sigma=(2, 2, 2))
    assert p.shape == (15, 15, 15) and p[7, 7, 7] == 1.0 and p.max() == 1.0
    p = synth.gaussian_psf(31, 31, 63, sigma=(2, 2.2, 12))
    assert p.shape == (63, 31, 31)
    assert synth.view_angles(8) == [15 + 45 * k for k in range(8)]
    h = synth.hourglass_psf(21)
    assert h.shape == (21, 21, 21) and h.min() >= 0


def test_no_gpu_means_loud_failure(mvs):
    """The product has no CPU fallback: without a usable gfx950 device context creation raises."""
    n = ctypes.c_int(0)
    rc = mvs._lib.load().mvsim_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present; covered by the gpu tests")
    with pytest.raises(mvs.MvsimNoDeviceError):
        mvs.Context(0)
    with pytest.raises(mvs.MvsimNoDeviceError):
        mvs.SimulateMultiViewDataset.rotateAroundAxis(np.zeros((4, 4, 4), np.float32), 0, 10)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "multiview-simulation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in text and "from oracle" not in text and "mvsim_oracle" not in text, f
                assert "liborc" not in text, f


# ------------------------------------------------------------------------------------------------ TIFF I/O (8f rank 4)
def test_tiff_reader_and_imagej_writer_round_trip(mvs, synth, tmp_path):
    """Tools.open / Tools.save (Tools.java:88-105,162-232).  ImageJ's FileSaver writes a stack as: header, first IFD at
    offset 8 (11 entries), the "ImageJ=" description, every plane contiguously, then one IFD per further plane.  The
    expected bytes of a small stack are assembled here entry by entry (TIFF 6.0 tags, big-endian) and compared with
    what the writer produces; the reader must return the planes from either."""
    import struct
    tiffio = importlib.import_module("multiview-simulation_amd.tiffio")
    img = (np.arange(3 * 2 * 3, dtype=np.float32).reshape(3, 2, 3) / 16).astype(np.float32)
    nz, h, w = img.shape
    desc = b"ImageJ=1.48o\nimages=3\nslices=3\nloop=false\nmin=0.0\nmax=1.0\n\x00"
    ifd_len = 2 + 11 * 12 + 4
    desc_off, plane = 8 + ifd_len, w * h * 4
    data_off = desc_off + len(desc)
    tail = data_off + nz * plane

    def ifd(strip, nxt):
        short = lambda v: v << 16                                             # a SHORT value sits left-justified in its field
        ent = [(254, 4, 1, 0),                # NewSubfileType
               (256, 4, 1, w), (257, 4, 1, h),
               (258, 3, 1, short(32)),        # BitsPerSample
               (262, 3, 1, short(1)),         # PhotometricInterpretation: BlackIsZero
               (270, 2, len(desc), desc_off), # ImageDescription
               (273, 4, 1, strip),            # StripOffsets
               (277, 3, 1, short(1)),         # SamplesPerPixel
               (278, 3, 1, short(h)),         # RowsPerStrip
               (279, 4, 1, plane),            # StripByteCounts
               (339, 3, 1, short(3))]         # SampleFormat: IEEE float
        return struct.pack(">H", 11) + b"".join(struct.pack(">HHII", *e) for e in ent) + struct.pack(">I", nxt)
    want = b"MM\x00\x2a" + struct.pack(">I", 8) + ifd(data_off, tail) + desc + img.astype(">f4").tobytes()
    want += ifd(data_off + plane, tail + ifd_len) + ifd(data_off + 2 * plane, 0)
    out = str(tmp_path / "stack.tif")
    tiffio.save_tiff(img, out, display_range=(0.0, 1.0))
    assert open(out, "rb").read() == want
    assert np.array_equal(mvs.Tools.open(out), img)
    # a stack with only the first IFD (what ImageJ writes beyond 4 GB): the planes follow the first strip
    single = str(tmp_path / "single_ifd.tif")
    open(single, "wb").write(b"MM\x00\x2a" + struct.pack(">I", 8) + ifd(data_off, 0) + desc + img.astype(">f4").tobytes())
    assert np.array_equal(mvs.Tools.open(single), img)
    # the PSF-sized stack the reference's callers load (51 planes of 51 x 51), default description = data range
    psf = synth.measured_like_psf(51)
    assert psf.shape == (51, 51, 51) and psf.max() == np.float32(0.99) and psf.min() == 0.0
    assert np.unravel_index(np.argmax(psf), psf.shape) == (25, 25, 25)          # the PSF peaks at its centre K/2
    assert 0.01 < float((psf > 0).mean()) < 0.12
    mvs.Tools.save(psf, out)
    assert np.array_equal(mvs.Tools.open(out), psf)
    plane_f = str(tmp_path / "plane.tif")
    mvs.Tools.save(psf[7], plane_f)                                              # 2-D image -> single-plane TIFF
    assert np.array_equal(mvs.Tools.open(plane_f)[0], psf[7])


def test_reference_psf_stacks_through_tiffio(synth):
    """The only data the reference holds for the path: the 18 measured PSF stacks `src/main/resources/Angle*.tif`, loaded through
    `Tools.open(file, true)` = read + makeSquare (Tools.java:297-349) at SimulateMultiViewDataset.java:579.  The files are GPL data
    and stay in /root/reference; `tests/golden/psf_tiff_facts.json` holds FACTS about them derived through this repository's reader
    (`tests/golden/make_psf_tiff_facts.py`).  Where the reference is present (the build container) every fact is re-derived from the
    files and must equal the JSON -- including that `tiffio.save_tiff` reproduces all 18 files BYTE FOR BYTE from the decoded pixels
    and the description's display range (the writer = Tools.save, Tools.java:88-105) and that an independent numpy decode agrees with
    `tiffio.open_tiff`.  Everywhere (the GPU box has no /root/reference) the JSON must be self-consistent, and the synthetic stand-in
    the tests use instead of the files must lie inside the range the real stacks span."""
    import json
    gold = os.path.join(ROOT, "tests", "golden")
    rec = json.load(open(os.path.join(gold, "psf_tiff_facts.json")))
    files = rec["files"]
    assert len(files) == 18 and rec["distinct_contents"] == len(rec["identical_pixel_groups"]) == 8
    assert sorted(n for g in rec["identical_pixel_groups"] for n in g) == sorted(files)
    for g in rec["identical_pixel_groups"]:
        assert len({files[n]["pixels_sha256"] for n in g}) == 1
    assert len({f["pixels_sha256"] for f in files.values()}) == 8
    for name, f in files.items():
        assert f["dims_xyz"] == [51, 51, 51] and f["byte_order"] == "big" and f["ifds"] == 51, name
        assert f["peak_index_xyz"] == [25, 25, 25] and f["max"] == float(np.float32(0.99)) and f["min"] == 0.0, name   # peak at K/2
        assert f["independent_decode_equal"] and f["save_tiff_reproduces_file"] and f["make_square_is_identity"], name
        # ImageJ's stack layout: header 8 + first IFD (2 + 11 * 12 + 4) + description, then the planes back to back
        desc = "".join(f"{k}={f['description'][k]}\n" for k in ("ImageJ", "images", "slices", "loop", "min", "max"))
        assert f["first_strip_offset"] == 8 + 138 + len(desc) + 1, name
        assert f["file_bytes"] == f["first_strip_offset"] + 51 * 51 * 51 * 4 + 50 * 138, name
    r, s = rec["ranges_over_the_files"], rec["stand_in"]
    for key, vals in (("sum_f64", [f["sum_f64"] for f in files.values()]), ("rank1_energy", [f["rank1_energy"] for f in files.values()])):
        assert r[key] == [min(vals), max(vals)]
    # the stand-in the tests, the bench's `small_views` leg and the examples run on, against the real stacks
    g = synth.measured_like_psf(51)
    sys.path.insert(0, gold)
    facts = importlib.import_module("make_psf_tiff_facts")
    sig = facts._sigmas(g)
    assert g.max() == np.float32(0.99) and g.min() == 0.0 and np.unravel_index(np.argmax(g), g.shape) == (25, 25, 25)
    assert abs(float(g.sum(dtype=np.float64)) - s["sum_f64"]) < 1e-6 * s["sum_f64"]
    assert r["sum_f64"][0] <= s["sum_f64"] <= r["sum_f64"][1]
    assert r["nonzero_share"][0] <= np.count_nonzero(g) / g.size <= r["nonzero_share"][1]
    assert r["rank1_energy"][0] <= facts._rank1_energy(g) <= r["rank1_energy"][1]
    for ax, key in enumerate(("sigma_x", "sigma_y", "sigma_z")):
        assert r[key][0] * 0.99 <= sig[ax] <= r[key][1] * 1.01, (key, sig[ax], r[key])
    if os.path.isdir(facts.REF_DIR):
        assert json.loads(json.dumps(facts.collect())) == rec          # every fact again, from the reference's own files


def test_tiff_reader_little_endian_multi_strip_and_rejections(mvs, tmp_path):
    import struct
    rng = np.random.default_rng(3)
    img = rng.random((3, 5, 4), dtype=np.float32)
    # hand-built little-endian TIFF, two strips per plane, IFDs in front of the data
    n_ent, planes = 9, []
    ifd_size = 2 + 12 * n_ent + 4
    head = 8
    arrays_off = head + 3 * ifd_size                    # per plane: 2 strip offsets + 2 strip byte counts
    data_off = arrays_off + 3 * 16
    body = b""
    for z in range(3):
        rows0 = img[z, :3].astype("<f4").tobytes()
        rows1 = img[z, 3:].astype("<f4").tobytes()
        o0 = data_off + len(body)
        body += rows0 + rows1
        planes.append((o0, len(rows0), o0 + len(rows0), len(rows1)))
    f = b"II\x2a\x00" + struct.pack("<I", head)
    arrays = b""
    for z, (o0, c0, o1, c1) in enumerate(planes):
        a_off = arrays_off + 16 * z
        ent = [(256, 4, 1, 4), (257, 4, 1, 5), (258, 3, 1, 32), (259, 3, 1, 1), (273, 4, 2, a_off), (277, 3, 1, 1),
               (278, 4, 1, 3), (279, 4, 2, a_off + 8), (339, 3, 1, 3)]
        f += struct.pack("<H", n_ent) + b"".join(struct.pack("<HHII", *e) for e in ent)
        f += struct.pack("<I", head + (z + 1) * ifd_size if z < 2 else 0)
        arrays += struct.pack("<IIII", o0, o1, c0, c1)
    path = str(tmp_path / "le.tif")
    open(path, "wb").write(f + arrays + body)
    assert np.array_equal(mvs.Tools.open(path), img)
    # 16-bit samples: "PixelType not supported"
    bad = bytearray(f + arrays + body)
    bad[head + 2 + 12 * 2 + 8] = 16
    open(path, "wb").write(bytes(bad))
    with pytest.raises(ValueError):
        mvs.Tools.open(path)
    open(path, "wb").write(b"not a tiff at all")
    with pytest.raises(ValueError):
        mvs.Tools.open(path)


def test_make_square(mvs):
    """Tools.java:313-349: pad to the largest dimension with the minimum, source index i -> i + S/2 - N/2."""
    a = np.arange(2 * 3 * 5, dtype=np.float32).reshape(2, 3, 5) + 1
    sq = mvs.Tools.makeSquare(a)
    assert sq.shape == (5, 5, 5)
    assert np.array_equal(sq[1:3, 1:4, :], a)           # offsets 5//2 - 2//2 = 1, 5//2 - 3//2 = 1, 0
    mask = np.ones_like(sq, bool)
    mask[1:3, 1:4, :] = False
    assert np.all(sq[mask] == 1.0)
    cube = np.ones((4, 4, 4), np.float32)
    assert np.array_equal(mvs.Tools.makeSquare(cube), cube)


def _bench_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                              "MVSIM_BENCH_SELF_LAUNCHED")}


def test_bench_self_launcher_starts_its_ranks():
    """`python bench.py --gpus 2 --backend gloo` without torchrun: the parent starts two children (RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT) and forwards rank 0's line; --dry-run-launch stops after the rendezvous
    and an all-reduce of 1 (no GPU here).  A child that fails makes the parent fail; --gpus that contradicts WORLD_SIZE is
    an error, also for WORLD_SIZE=1 (round 2 ran one rank and printed n_gpus 1 there)."""
    import json
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--backend", "gloo", "--dry-run-launch"], capture_output=True, text=True,
                       timeout=300, env=_bench_env())
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d == {"dry_run_launch": True, "n_gpus": 2, "ranks_seen": 2, "backend": "gloo", "self_launched": True}
    bad = subprocess.run([sys.executable, bench, "--gpus", "2", "--backend", "no_such_backend", "--dry-run-launch"], capture_output=True,
                         text=True, timeout=300, env=_bench_env())
    assert bad.returncode != 0 and "exited with status" in bad.stderr
    mism = subprocess.run([sys.executable, bench, "--gpus", "8", "--dry-run-launch"], capture_output=True, text=True, timeout=120,
                          env=dict(_bench_env(), WORLD_SIZE="1", RANK="0"))
    assert mism.returncode != 0 and "WORLD_SIZE=1" in mism.stderr


def test_bench_control_plane_defaults_to_gloo_and_the_fastest_broadcast_wins():
    """VERDICT r5 next #7.  (a) `python bench.py --gpus 2` with NO --backend: torch.distributed is the control plane only and comes up
    on gloo (one RCCL instance per rank: the C ABI's own communicator carries the data).  (b) the selection among the forms of the
    ground-truth broadcast that one N > 1 invocation times back to back (`pick_broadcast`): the smallest ms_per_step wins, a form that
    failed or has no number cannot, ties and an empty table go to the form the line was asked for."""
    import json
    import subprocess
    import sys
    bench_py = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench_py, "--gpus", "2", "--dry-run-launch"], capture_output=True, text=True, timeout=300, env=_bench_env())
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["backend"] == "gloo" and d["ranks_seen"] == 2 and d["n_gpus"] == 2
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    pick = bench.pick_broadcast
    assert bench.BROADCAST_FORMS == ("scatter_allgather", "pipelined", "peer_copy")
    assert pick({"scatter_allgather": {"ms_per_step": 2.3}, "pipelined": {"ms_per_step": 1.9}, "peer_copy": {"ms_per_step": 2.0}}, "scatter_allgather") == "pipelined"
    assert pick({"scatter_allgather": {"ms_per_step": 2.3}, "pipelined": {"failed": "RuntimeError('x')"}, "peer_copy": {"ms_per_step": 2.0}}, "scatter_allgather") == "peer_copy"
    assert pick({"scatter_allgather": {"ms_per_step": 2.3}, "pipelined": {"ms_per_step": 2.3}}, "scatter_allgather") == "scatter_allgather"       # a tie stays
    assert pick({"scatter_allgather": {"ms_per_step": 2.3}, "pipelined": {"ms_per_step": 1.0, "failed": "late"}}, "scatter_allgather") == "scatter_allgather"
    assert pick({"scatter_allgather": {"failed": "x"}, "peer_copy": {"ms_per_step": 5.0}}, "scatter_allgather") == "peer_copy"
    assert pick({}, "ring") == "ring" and pick({"ring": {"ms_per_step": 3.0}, "pipelined": {"ms_per_step": None}}, "ring") == "ring"
    # the line leads with what a reader of its first 200 characters needs
    line = bench.ordered_line({"metric": "m" * 90, "value": 74000.123, "unit": "Mvoxel/s", "value_dense": 63000.5, "steps": 5,
                               "roofline": {"frac": 0.14159, "whole_view": {"frac": 0.29911}}})
    head = json.loads(line)
    assert list(head)[:5] == ["value", "unit", "value_dense", "whole_view_frac", "roofline_frac"]
    assert '"value_dense"' in line[:200] and '"whole_view_frac": 0.2991' in line[:200] and '"roofline_frac": 0.1416' in line[:200]


def test_slab_range_partitions_the_planes(mvs):
    """mvsim_slab_range (host only): contiguous, balanced, covering partition of [0, Nz)."""
    L = importlib.import_module("multiview-simulation_amd._lib").load()
    for nz, ranks in ((1024, 8), (513, 4), (7, 3), (5, 5), (100, 1)):
        edges = []
        for r in range(ranks):
            z0, z1 = ctypes.c_int64(), ctypes.c_int64()
            assert L.mvsim_slab_range(nz, ranks, r, ctypes.byref(z0), ctypes.byref(z1)) == 0
            edges.append((z0.value, z1.value))
        assert edges[0][0] == 0 and edges[-1][1] == nz
        assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
        sizes = [b - a for a, b in edges]
        assert max(sizes) - min(sizes) <= 1
    z0, z1 = ctypes.c_int64(), ctypes.c_int64()
    assert L.mvsim_slab_range(10, 2, 2, ctypes.byref(z0), ctypes.byref(z1)) != 0


# ------------------------------------------------------------------------------------------------ plain-C consumer
def _build_c_smoke(tmp_path):
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "multiview-simulation_amd")
    exe = str(tmp_path / "c_abi_smoke")
    cmd = [shutil.which("gcc") or "gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"),
           os.path.join(root, "tests", "c_abi", "smoke.c"), "-L" + pkg, "-lmvsim", "-Wl,-rpath," + pkg,
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lm", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_c_abi_compiles_as_plain_c_and_fails_loudly_without_gpu(tmp_path):
    """include/mvsim.h is a C header (gcc -std=c99 -Werror), the library links without Python or torch, and a host
    without a GPU gets MVSIM_ENODEV -- not a CPU fallback."""
    import subprocess
    import torch
    exe = _build_c_smoke(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    if torch.cuda.is_available():
        assert r.returncode == 0, r.stderr
    else:
        assert r.returncode == 3 and "no HIP device" in r.stderr


# ------------------------------------------------------------------------------------------------ sanitizers (CPU side)
def test_oracle_under_address_and_ub_sanitizer():
    """`make -C oracle asan`: the C restatement under -fsanitize=address,undefined on exactly sized heap buffers."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    b = subprocess.run(["make", "-C", os.path.join(root, "oracle"), "asan"], capture_output=True, text=True)
    assert b.returncode == 0, b.stdout + b.stderr
    r = subprocess.run([os.path.join(root, "oracle", "asan_check")], capture_output=True, text=True)
    assert r.returncode == 0 and "oracle sanitizer run ok" in r.stdout, r.stdout + r.stderr


def test_c_abi_host_only_leg_under_sanitizers(tmp_path):
    """tests/c_abi/host_only.c built with ASan + UBSan against libmvsim.so: the entry points that need no GPU."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "multiview-simulation_amd")
    exe = str(tmp_path / "c_abi_host_only")
    cmd = [shutil.which("gcc") or "gcc", "-std=c99", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "c_abi", "host_only.c"),
           "-L" + pkg, "-lmvsim", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lm", "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True)
    assert b.returncode == 0, b.stdout + b.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0")   # the ROCm runtime keeps process-lifetime blocks
    r = subprocess.run([exe], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "host-only sanitizer run ok" in r.stdout, r.stdout + r.stderr


def test_jni_shim_covers_every_native_method():
    """Source-only Java layer (no JDK in the image): at least keep MvsimNative.java and java/jni/mvsim_jni.cpp in step --
    every `static native` method has exactly one JNI_FN definition, and the shim only calls C-ABI symbols the header
    declares."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    java = open(os.path.join(root, "java/src/main/java/net/preibisch/simulation/gpu/MvsimNative.java")).read()
    cpp = open(os.path.join(root, "java/jni/mvsim_jni.cpp")).read()
    header = open(os.path.join(root, "include/mvsim.h")).read()
    natives = re.findall(r"static\s+native\s+[\w.\[\]<>]+\s+(\w+)\s*\(", java)
    defined = [d for d in re.findall(r"JNI_FN\((\w+)\)", cpp) if d != "name"]      # "name" is the macro's own parameter
    assert natives and sorted(natives) == sorted(defined), (set(natives) ^ set(defined))
    declared = set(re.findall(r"\b(mvsim_\w+)\s*\(", header))
    for sym in set(re.findall(r"\b(mvsim_[a-z0-9_]+)\s*\(", cpp)):
        assert sym in declared, sym


def test_jni_shim_syntax_checks_against_a_jni_h_subset():
    """java/jni/mvsim_jni.cpp through `g++ -fsyntax-only` against tests/jni_stub/jni.h -- a hand-written SUBSET of the JNI C++
    surface (the image has no JDK).  This is a compile check of the shim against include/mvsim.h and the JNI signatures it
    uses; it links nothing, runs nothing and pins nothing about the shim's behaviour on a JVM."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    assert gxx, "g++ is part of the image"
    r = subprocess.run([gxx, "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "tests", "jni_stub"),
                        "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "java", "jni", "mvsim_jni.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # the stub declares only JNIEnv members the specification has, each used by the shim (no invented conveniences)
    stub = open(os.path.join(ROOT, "tests", "jni_stub", "jni.h")).read()
    cpp = open(os.path.join(ROOT, "java", "jni", "mvsim_jni.cpp")).read()
    for member in re.findall(r"^\s+\w[\w\s\*]*?\b(\w+)\(", stub.split("struct JNIEnv_ {")[1].split("};")[0], flags=re.M):
        assert "env->" + member + "(" in cpp, member


def test_java_facade_passes_a_fresh_generation_per_dataset():
    """simulateViews passed gt_generation 0 for every dataset, and the staging block of a second dataset of
    the same size comes from the pool at the same host address -- the library then skipped the upload and simulated the
    previous dataset.  Source-level check (no JDK): the facade hands the block's own generation to the native call, the
    generation is drawn when a block is filled, and nothing passes a constant."""
    base = os.path.join(ROOT, "java/src/main/java/net/preibisch/simulation/gpu")
    facade = open(os.path.join(base, "SimulateMultiViewDatasetGPU.java")).read()
    buffers = open(os.path.join(base, "Buffers.java")).read()
    assert "simulateViewAsync( ctx, gt.floats, gt.generation, d," in facade and "gt.floats, 0L" not in facade
    assert "GENERATION.getAndIncrement()" in buffers and "AtomicLong" in buffers
    # every per-stage operator of the reference's call sites goes through the z-slab natives: no 2^29-voxel cap on them
    for op in ("rotateAroundAxisSlabs", "attenuate3dSlabs", "convolveSlabs", "extractSlicesSlabs"):
        assert "MvsimNative." + op + "(" in facade, op
    assert "public static void main( final String[] args )" in facade
    for name in ("rendered.tif", "groundtruth.tif", "rot_view_", "att_view_", "con_view_", "acq_view_", "iso_view_", "aligned_view_",
                 "aligned_view_psf_", "aligned_view_weights", "sum_weights.tif"):
        assert name in facade, name


def test_committed_pmc_traffic_belongs_to_this_build():
    """bench.py prints roofline.traffic only while the SHA of the kernel sources equals the one the PMC record under profiles/ was
    collected on (tools/profile_all.sh).  A mismatch is not an error of the build -- the record has to be collected again on a GPU box --
    so it is reported as a skip, with what bench.py will do about it."""
    import importlib, json
    build = importlib.import_module("multiview-simulation_amd.build")
    for name in ("r06_traffic.json", "r06_traffic_1024.json"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            pytest.skip(f"profiles/{name} has not been collected yet (bash tools/profile_all.sh TAG on a GPU box): bench.py prints roofline.traffic = null")
        rec = json.load(open(path))
        assert rec["views_profiled"] >= 1 and rec["per_view_bytes"]["convolve"] > 0
    path = os.path.join(ROOT, "profiles", "r06_traffic.json")
    rec = json.load(open(path))
    if rec["kernel_sha"] != build.source_sha():
        pytest.skip(f"{os.path.relpath(path, ROOT)} was collected on kernel sources {rec['kernel_sha']}, this tree is {build.source_sha()}: "
                    "bench.py will print roofline.traffic = null until `bash tools/profile_all.sh TAG` has been run again")
    assert rec["views_profiled"] >= 1 and rec["per_view_bytes"]["convolve"] > 0


@pytest.mark.parametrize("nranks,root,count,pieces", [(2, 0, 1000, 8), (2, 1, 100003, 3), (3, 0, 100003, 8), (4, 2, 134217728, 8),
                                                        (8, 0, 134217728, 8), (8, 5, 1 << 20, 4), (8, 0, 100, 8), (5, 4, 17, 8), (8, 3, 4097, 1)])
def test_pipelined_broadcast_plan_is_complete_and_matched(mvs, nranks, root, count, pieces):
    """mvsim_comm_broadcast_plan (the schedule of option broadcast=pipelined, host only): simulated over every rank of the job --
    in each stage every send has exactly one matching receive at its peer (same range) and vice versa (a group of ncclSend /
    ncclRecv cannot deadlock then), a rank only sends floats it already holds at the START of the stage (stages follow each other in
    stream order; inside a stage nothing is ordered), and after the last stage every rank holds every float.  The root's outbound
    traffic is the volume once per peer-chunk, not twice."""
    plans = [mvs.broadcast_plan(nranks, r, root, count, pieces) for r in range(nranks)]
    have = [np.zeros(count, bool) for _ in range(nranks)]
    have[root][:] = True
    nstage = 1 + max([op[0] for p in plans for op in p], default=-1)
    sent_by_root = 0
    for st in range(nstage):
        sends = sorted((r, op[2], op[3], op[4]) for r, p in enumerate(plans) for op in p if op[0] == st and op[1] == "send")
        recvs = sorted((op[2], r, op[3], op[4]) for r, p in enumerate(plans) for op in p if op[0] == st and op[1] == "recv")
        assert sends == recvs, (st, sends[:3], recvs[:3])                       # (from, to, first, count) on both sides
        assert len(set((a, b) for a, b, _, _ in sends)) == len(sends)          # at most one transfer per ordered pair and stage
        for a, b, first, n in sends:
            assert n > 0 and first >= 0 and first + n <= count and a != b
            assert have[a][first:first + n].all(), (st, a, b, first, n)        # the source holds what it sends before the stage starts
        for a, b, first, n in sends:
            have[b][first:first + n] = True
            sent_by_root += n if a == root else 0
        for p in plans:                                                         # stages are issued in order
            stages = [op[0] for op in p]
            assert stages == sorted(stages)
    assert all(h.all() for h in have)
    assert sent_by_root == count * (1 if nranks > 1 else 0) + (count - (count // (nranks - 1) & ~15) * (nranks - 1)) * (nranks - 2 if nranks > 1 else 0)

