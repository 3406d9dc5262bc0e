# A/B harness for experiments on the GPU box: bash tools/ab.sh "<pytest -k expr>" "<opt string 1>" "<opt string 2>" ...
set -e
K="$1"; shift
if [ -n "$K" ]; then
  python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$K" > gpurun_out/ab_tests.log 2>&1 || { tail -40 gpurun_out/ab_tests.log; exit 1; }
  tail -1 gpurun_out/ab_tests.log
fi
i=0
for o in "$@"; do
  i=$((i+1))
  MVSIM_OPTIONS="$o" python bench.py --no-cpu-baseline --no-end-to-end --no-size-1024 --no-two-streams --no-compact-queue-leg > gpurun_out/ab_$i.log 2>&1 || { tail -5 gpurun_out/ab_$i.log; exit 1; }
  python - "$o" gpurun_out/ab_$i.log <<'PY'
import json, sys
for l in open(sys.argv[2]):
    if l.startswith("{"):
        d = json.loads(l); s = d["roofline"]["stage_ms"]
        print(f"[{sys.argv[1]}] {d['value']:.0f} Mvox/s  total {s['total_ms']:.3f}  rot {s['rotate_ms']:.3f} conv {s['convolve_ms']:.3f} (A {s['pass_a_ms']:.3f} B {s['pass_b_ms']:.3f} C {s['pass_c_ms']:.3f} D {s['pass_d_ms']:.3f} E {s['pass_e_ms']:.3f}) extract {s['extract_ms']:.3f}")
PY
done
