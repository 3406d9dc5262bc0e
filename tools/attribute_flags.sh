# stage times of the serial bench leg for experiment builds: bash tools/attribute_flags.sh "<cflags 1>" "<cflags 2>" ...
# e.g. the fused rotate + attenuate + x transform kernel:
#   bash tools/attribute_flags.sh "-DMVSIM_EXP_ROTFFT_NOFFT" "-DMVSIM_EXP_ROTFFT_NOBLEND" "-DMVSIM_EXP_ROTFFT_NOFFT -DMVSIM_EXP_ROTFFT_NOBLEND" "-DMVSIM_ROTFFT_PREFETCH=4"
set -e
for f in "$@"; do
  MVSIM_EXTRA_CFLAGS="-DMVSIM_DEV_ATTRIBUTION $f" python -c "import importlib; b = importlib.import_module('multiview-simulation_amd.build'); b.build(force=True)"
  for r in 1 2; do
  python bench.py --serial --no-cpu-baseline --no-end-to-end --no-size-1024 --no-two-streams --no-dense-leg 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['roofline']['stage_ms']; print('[$f]', round(d['value']), 'A %.3f B %.3f C %.3f D %.3f E %.3f conv %.3f rot %.3f ext %.3f' % (s['pass_a_ms'], s['pass_b_ms'], s['pass_c_ms'], s['pass_d_ms'], s['pass_e_ms'], s['convolve_ms'], s['rotate_ms'], s['extract_ms']))
"
  done
done
python -c "import importlib; b = importlib.import_module('multiview-simulation_amd.build'); b.build(force=True)"
