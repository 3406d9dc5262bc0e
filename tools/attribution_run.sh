#!/bin/bash
# Stage times (HIP events, one 512^3 view, serial) of experiment builds that compile parts of a kernel's work out -- the attribution
# figures quoted in DESIGN.md section 4.  The variants are built beforehand (MVSIM_EXTRA_CFLAGS="-DMVSIM_DEV_ATTRIBUTION -DMVSIM_DEV_SIZES <macro>", copied with
# the Python package into wt_exp_<n>/ together with flags.txt); this script only runs them:   bash tools/attribution_run.sh > profiles/r04_attribution.txt
echo "# experiment builds (each removes one part of a kernel's work, and with it the correctness of the results): HIP-event stage times of one 512^3 view, two runs each"
for d in wt_exp_*; do
  for r in 1 2; do
    echo -n "[$(cat $d/flags.txt)] "
    python3 tools/ab_extract.py $d 2>/dev/null | sed 's/^wt_exp_[0-9]* //'
  done
done
