#!/bin/bash
# round 6: segments per resolver block (option poisson_resolve_group) A/B on one box: 512^3 phantom and dense, 31^3 PSF, inc 1
set -e
for r in 1 2; do
  for G in 1 2 4 8; do
    for K in phantom dense; do
      python3 tools/view_time.py 512 512 512 31 31 31 1 gt=$K poisson_resolve_group=$G
    done
  done
done
