#!/bin/bash
# round 6: y lines as two half-length transforms (k_fft_lines_split, default) against the whole-length 8-column tiles (exp=8), same box:
# lines of 2160 / 2048 points (one block per CU otherwise) and of 1024 .. 1152 points (64-byte rows otherwise)
set -e
for r in 1 2; do
  for W in "1024 1024 1024 31 31 31 1 gt=phantom2x" "1024 1024 1024 31 31 63 4" "2048 2048 512 63 63 63 3" "2048 2048 512 63 63 63 1"; do
    python3 tools/view_time.py $W exp=8
    python3 tools/view_time.py $W
  done
done
