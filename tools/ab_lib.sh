# A/B of BUILDS on one box: bash tools/ab_lib.sh A B A B   (multiview-simulation_amd/libmvsim_<name>.so, built beforehand and shipped with the snapshot)
set -e
cp multiview-simulation_amd/libmvsim.so gpurun_out/libmvsim_keep.so
for v in "$@"; do
  cp multiview-simulation_amd/libmvsim_$v.so multiview-simulation_amd/libmvsim.so
  python bench.py --no-cpu-baseline --no-end-to-end --no-size-1024 --no-two-streams --no-compact-queue-leg --no-small-views --no-main-iteration > gpurun_out/ab_lib_$v.log 2>&1 || { tail -5 gpurun_out/ab_lib_$v.log; exit 1; }
  python - "$v" gpurun_out/ab_lib_$v.log <<'PY'
import json, sys
for l in open(sys.argv[2]):
    if l.startswith("{"):
        d = json.loads(l); s = d["roofline"]["stage_ms"]
        print(f"[{sys.argv[1]}] {d['value']:.0f} Mvox/s dense {d['value_dense']:.0f}  total {s['total_ms']:.3f}  rot {s['rotate_ms']:.3f} (dense {d['no_empty_space']['rotate_attenuate_ms']:.3f}) conv {s['convolve_ms']:.3f} (B {s['pass_b_ms']:.3f} C {s['pass_c_ms']:.3f} D {s['pass_d_ms']:.3f} E {s['pass_e_ms']:.3f}) extract {s['extract_ms']:.3f}")
PY
done
cp gpurun_out/libmvsim_keep.so multiview-simulation_amd/libmvsim.so
