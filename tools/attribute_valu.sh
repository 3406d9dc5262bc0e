# VALU instructions per wave of the Poisson kernels for experiment builds: bash tools/attribute_valu.sh "<cflags 1>" "<cflags 2>" ...
set -e
for f in "$@"; do
  MVSIM_EXTRA_CFLAGS="-DMVSIM_DEV_ATTRIBUTION $f" python -c "import importlib; b = importlib.import_module('multiview-simulation_amd.build'); b.build(force=True)"
  echo "[$f]"
  bash tools/pmc_i.sh "" > gpurun_out/pi.txt 2>&1 || { tail -5 gpurun_out/pi.txt; exit 1; }
  grep -i "extract\|resolve" gpurun_out/pi.txt | cut -c1-40,88-200
  python bench.py --no-cpu-baseline --no-end-to-end --no-size-1024 --no-two-streams 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        s=json.loads(l)['roofline']['stage_ms']; print('   extract_ms', s['extract_ms'], 'total', s['total_ms'])
"
done
# leave the product build behind, not the last experiment
python -c "import importlib; b = importlib.import_module('multiview-simulation_amd.build'); b.build(force=True)"
