# A/B on environment settings with the default bench line: bash tools/ab_env.sh "VAR=1 OTHER=2" "VAR=0" ...
set -e
for e in "$@"; do
  env $e python bench.py --no-cpu-baseline --no-end-to-end --no-size-1024 --no-two-streams 2>gpurun_out/ab_env.err | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['roofline']['stage_ms']; print('[$e]', round(d['value']), 'Mvox/s  rot', s['rotate_ms'], 'conv', s['convolve_ms'], '(A %.3f B %.3f C %.3f D %.3f E %.3f)' % (s['pass_a_ms'], s['pass_b_ms'], s['pass_c_ms'], s['pass_d_ms'], s['pass_e_ms']), 'extract', s['extract_ms'], 'view', round(d['roofline']['whole_view']['ms'],3))
"
done
