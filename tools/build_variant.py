"""Build an experiment variant of the library beside the product build (never instead of it):
    python tools/build_variant.py NAME "-DMACRO=... -DOTHER"  [file.hip ...]
writes multiview-simulation_amd/libmvsim_NAME.so (objects in multiview-simulation_amd/build_NAME/); the listed translation units are
compiled with the extra flags, every other object is taken from the product build (multiview-simulation_amd/build/), which must be
current.  tools/ab_lib.sh / ab_lib_sizes.sh then A/B the variants on one box."""
import importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
b = importlib.import_module("multiview-simulation_amd.build")
name, extra = sys.argv[1], sys.argv[2].split()
only = set(sys.argv[3:])
b.build()
out_dir = os.path.join(b.HERE, "build_" + name)
os.makedirs(out_dir, exist_ok=True)
objs, procs = [], []
for src in b.sources():
    base = os.path.basename(src)
    if only and base not in only:
        objs.append(os.path.join(b.HERE, "build", base + ".o"))
        continue
    obj = os.path.join(out_dir, base + ".o")
    objs.append(obj)
    procs.append((src, subprocess.Popen([b._hipcc(), "-x", "hip", "-c", src, "-o", obj] + b._common_flags() + extra,
                                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
for src, p in procs:
    out, _ = p.communicate()
    if p.returncode:
        sys.exit(f"--- {src} ---\n{out}")
lib = os.path.join(b.HERE, f"libmvsim_{name}.so")
subprocess.check_call([b._hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib] + objs +
                      ["-L" + os.path.join(b.ROCM, "lib"), "-lrocfft", "-lrccl", "-lroctx64", "-Wl,-rpath," + os.path.join(b.ROCM, "lib")])
print(lib)
