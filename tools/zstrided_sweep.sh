#!/bin/bash
# The z pass of compact views (inc > 1), one box: every plane (zconv_strided=0: k_zconv or the inline FFT), the default (cost rule of
# zconv_strided_chunk) and k_zconv_strided forced (exp=2).    bash tools/zstrided_sweep.sh > profiles/r04_zstrided.txt
echo "# tools/view_time.py (HIP-event stage times of one device-resident view, overlaps off); pass_c_ms is the z pass"
for g in "1024 1024 1024 31 31 63 4" "1024 1024 1024 15 15 41 4" "1024 1024 1024 31 31 31 4" "2048 2048 512 63 63 63 3" "512 512 512 31 31 31 2" "512 512 512 31 31 31 3" "512 512 512 31 31 31 4" "512 512 512 31 31 63 3" "289 289 289 51 51 51 3"; do
  for o in "zconv_strided=0" "" "exp=2"; do
    python3 tools/view_time.py $g $o | sed -e 's/rotate_ms.*pass_c_ms/pass_c_ms/' -e 's/ pass_d_ms.*//'
  done
done
