"""Which kernels wait for a global load the moment they have asked for it?

    python tools/isa_wait_scan.py [source.hip ...]        (default: the three kernel files of the convolution / rotate / sampler path)

Compiles each file to gfx950 assembly with the product's flags (device only, nothing is linked or run: works without a GPU) and lists, per
kernel, the `global_load` instructions that are followed at once by `s_waitcnt vmcnt(0)` -- a load whose data the wave sits and waits for
before it asks for anything else.  One such pair at the head of a tile's loads means the block pays a whole trip to memory before its other
loads are even under way.  Where it comes from in this code base: `v = 0; if (wanted) v = load(...)` becomes a branch
whose merge of loaded and zero registers costs the compiler a register copy of the loaded value, hence the wait; a select instead of the
branch is turned back into one (CodeGenPrepare sinks a load that only a select uses).  What holds: an unconditional load from a clamped
address and an AND with a mask (fft_kernels.hip: zconv_keep4).  Dependent loads (an item fetched by a ticket, a table read behind a scalar
load) show up here as well and are what they are; the list is a place to look, not a verdict.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multiview-simulation_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-I/opt/rocm/include",
         "-I" + os.path.join(ROOT, "include"), "--offload-device-only", "-S"]


def scan(path):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-o", out, path], capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit(r.stderr)
        lines = open(out).read().split("\n")
    cur, loads, hits = None, {}, {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\S+):", l)
        if m:
            cur = m.group(1)
        if cur and "global_load" in l:
            loads[cur] = loads.get(cur, 0) + 1
            nxt = [x for x in lines[i + 1:i + 4] if x.strip() and not x.strip().startswith(";")]
            if nxt and "s_waitcnt vmcnt(0)" in nxt[0]:
                hits[cur] = hits.get(cur, 0) + 1
    return loads, hits


def demangle(names):
    try:
        r = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True)
        return dict(zip(names, r.stdout.split("\n")))
    except OSError:
        return {n: n for n in names}


if __name__ == "__main__":
    files = sys.argv[1:] or [os.path.join(CSRC, f) for f in ("fft_kernels.hip", "rotate_fft.hip", "kernels.hip")]
    for f in files:
        loads, hits = scan(f)
        names = demangle(sorted(hits, key=lambda k: -hits[k]))
        print(f"== {os.path.relpath(f, ROOT)}: {len(loads)} kernels with global loads, {len(hits)} of them wait at once behind one")
        for k, pretty in names.items():
            print(f"   {hits[k]:3d} of {loads[k]:3d} loads   {pretty[:150]}")
