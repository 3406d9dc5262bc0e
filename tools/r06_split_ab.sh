#!/bin/bash
# round 6: y lines of one block per CU (L >= 1280) as two half-length transforms (k_fft_lines_split, default) against the one-block form (exp=8), same box
set -e
for r in 1 2; do
  for W in "2048 2048 512 63 63 63 3" "2048 2048 512 63 63 63 1" "2048 2048 256 31 31 31 1" "1536 1536 512 31 31 31 1"; do
    python3 tools/view_time.py $W exp=8
    python3 tools/view_time.py $W
  done
done
