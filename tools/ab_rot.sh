set -e
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused or rotate or attenuate or golden or config0 or full_width or slab or options or odd_dims" > gpurun_out/r02_t3.log 2>&1 || { tail -30 gpurun_out/r02_t3.log; exit 1; }
tail -2 gpurun_out/r02_t3.log
for v in 2 1; do
  MVSIM_OPTIONS="fused_rotate=$v" python bench.py --no-cpu-baseline --no-end-to-end --no-size-1024 > gpurun_out/r02_b3_rot$v.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/r02_b3_rot$v.log"):
    if l.startswith("{"):
        d=json.loads(l); print("fused_rotate=$v", round(d["value"]), d["roofline"]["stage_ms"])
PY
done
