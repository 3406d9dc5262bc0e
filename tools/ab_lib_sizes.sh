# A/B of BUILDS at the larger geometries (one view at a time, HIP-event stage times): bash tools/ab_lib_sizes.sh K L1 K L1
set -e
cp multiview-simulation_amd/libmvsim.so gpurun_out/libmvsim_keep.so
for v in "$@"; do
  cp multiview-simulation_amd/libmvsim_$v.so multiview-simulation_amd/libmvsim.so
  echo "[$v]"
  python tools/view_time.py 1024 1024 1024 31 31 31 1 gt=phantom2x 2>/dev/null | cut -c1-260
  python tools/view_time.py 1024 1024 1024 31 31 63 4 2>/dev/null | cut -c1-260
  python tools/view_time.py 2048 2048 512 63 63 63 3 2>/dev/null | cut -c1-260
done
cp gpurun_out/libmvsim_keep.so multiview-simulation_amd/libmvsim.so
