# per-kernel average durations (rocprofv3 --kernel-trace --stats, serial bench leg) of BUILDS on one box: bash tools/ktrace_libs.sh A B
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
cp multiview-simulation_amd/libmvsim.so gpurun_out/libmvsim_keep.so
for v in "$@"; do
  cp multiview-simulation_amd/libmvsim_$v.so multiview-simulation_amd/libmvsim.so
  rm -rf gpurun_out/kt && mkdir -p gpurun_out/kt
  rocprofv3 --kernel-trace --stats -d gpurun_out/kt -o run -- python3 bench.py --serial --no-cpu-baseline --no-end-to-end --no-size-1024 --no-two-streams --no-dense-leg --no-main-iteration --no-small-views --no-compact-queue-leg > gpurun_out/kt_$v.log 2>&1
  echo "[$v]"
  python3 tools/kstats.py gpurun_out/kt 12
  rm -rf gpurun_out/kt
done
cp gpurun_out/libmvsim_keep.so multiview-simulation_amd/libmvsim.so
