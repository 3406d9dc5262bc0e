#!/bin/bash
# end-to-end (page-locked host buffers in and out) against the number of host threads that widen the uint16 counts: bash tools/e2e_host_threads.sh 8 16 32 64
for T in "$@"; do
  MVSIM_OPTIONS="host_threads=$T" python bench.py --no-cpu-baseline --no-size-1024 --no-two-streams --no-dense-leg --no-main-iteration --no-small-views --no-compact-queue-leg --steps 5 --warmup 2 2>/dev/null | python -c "
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); e = d['end_to_end']
print('host_threads=$T', 'same gt', round(e['same_ground_truth']['ms_per_view'], 2), 'ms/view; fresh gt', round(e['fresh_ground_truth_per_view']['ms_per_view'], 2), '; f32 transfer', round(e['same_ground_truth_float32_transfer']['ms_per_view'], 2))"
done
