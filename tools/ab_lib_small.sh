# A/B of BUILDS on the reference's own sizes (views stacked): bash tools/ab_lib_small.sh K A2 K A2
set -e
cp multiview-simulation_amd/libmvsim.so gpurun_out/libmvsim_keep.so
for v in "$@"; do
  cp multiview-simulation_amd/libmvsim_$v.so multiview-simulation_amd/libmvsim.so
  echo "[$v]"; python tools/small_views.py c0 ref 256 2>/dev/null | grep -E "stacked|Gvoxel" | head -12
done
cp gpurun_out/libmvsim_keep.so multiview-simulation_amd/libmvsim.so
