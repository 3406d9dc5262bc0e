#!/bin/bash
# round 6: the two 8-column tiles of a 128-byte line on one XCD (k_fft_lines) against the plain grid order (exp=4), same box
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out
for r in 1 2; do
  for W in "1024 1024 1024 31 31 31 1 gt=phantom2x" "1024 1024 1024 31 31 63 4" "2048 2048 512 63 63 63 3" "2048 2048 512 63 63 63 1"; do
    python3 tools/view_time.py $W exp=4
    python3 tools/view_time.py $W
  done
done
