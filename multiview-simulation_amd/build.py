"""Build recipe for libmvsim.so (hipcc, gfx950 only, in-tree so the .so travels with the repo)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmvsim.so")
SOURCES = ["api.cpp", "comm.cpp", "kernels.hip", "fftconv.hip", "fft_kernels.hip", "rotate_fft.hip", "stencil.hip", "phantom.hip"]
HEADERS = ["common.h", "poisson_dev.h", "fft_dev.h", "../../include/mvsim.h"]
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _hipcc() -> str:
    for cand in (os.path.join(ROCM, "bin", "hipcc"), shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libmvsim.so)")


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def source_sha() -> str:
    """First 16 hex digits of the SHA-256 over the kernel / host sources of libmvsim.so (and the header): the identity
    the PMC traffic record of profiles/ is keyed by, so that bench.py never prints counters collected on other kernels."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(sources() + [os.path.normpath(os.path.join(CSRC, x)) for x in HEADERS])
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _common_flags():
    return [
        "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
        *os.environ.get("MVSIM_EXTRA_CFLAGS", "").split(),
        "-Wall", "-Wno-unused-result", "-I" + os.path.join(ROCM, "include"),
    ]


def is_current() -> bool:
    """The library exists, is newer than every source and was built with the flags this process would use (an experiment
    build with MVSIM_EXTRA_CFLAGS never passes for the product build)."""
    if not os.path.exists(LIB) or not os.path.exists(LIB + ".flags"):
        return False
    with open(LIB + ".flags") as fh:
        if fh.read() != " ".join(_common_flags()):
            return False
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS] + [os.path.abspath(__file__)]
    return all(os.path.getmtime(d) <= t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every translation unit for gfx950 and link libmvsim.so.

    -ffp-contract=off: the parity-relevant kernels state their rounding points explicitly
    (fmaf() where a fused op is wanted); the compiler must not fuse behind our back.
    """
    if not force and is_current():
        return LIB
    hipcc = _hipcc()
    objs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    common = _common_flags()
    # per-object rebuild: an object is kept when it is newer than its source, every header and this recipe, and was built
    # with the same flags (recorded beside it) -- fft_kernels.hip alone takes ~1.5 min
    flags_id = " ".join(common)
    hdr_time = max(os.path.getmtime(os.path.normpath(os.path.join(CSRC, h))) for h in HEADERS)
    hdr_time = max(hdr_time, os.path.getmtime(os.path.abspath(__file__)))
    procs = []
    for src in sources():
        obj = os.path.join(HERE, "build", os.path.basename(src) + ".o")
        objs.append(obj)
        stamp = obj + ".flags"
        fresh = (not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == flags_id
                 and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_time))
        if fresh:
            continue
        cmd = [hipcc, "-x", "hip", "-c", src, "-o", obj] + common
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, stamp, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, stamp, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            if os.path.exists(stamp):
                os.remove(stamp)
            sys.stderr.write(f"--- {src} ---\n{out}\n")
        else:
            with open(stamp, "w") as fh:
                fh.write(flags_id)
            if verbose and out.strip():
                sys.stderr.write(out)
    if failed:
        raise RuntimeError("hipcc failed")
    link = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs + [
        "-L" + os.path.join(ROCM, "lib"), "-lrocfft", "-lrccl", "-lroctx64", "-Wl,-rpath," + os.path.join(ROCM, "lib"),
    ]
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.check_call(link)
    with open(LIB + ".flags", "w") as fh:
        fh.write(flags_id)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
