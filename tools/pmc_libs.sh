# instruction and wait counters of the sampler's kernels for BUILDS on one box: bash tools/pmc_libs.sh A B
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
cp multiview-simulation_amd/libmvsim.so gpurun_out/libmvsim_keep.so
B="bench.py --no-cpu-baseline --no-end-to-end --no-size-1024 --no-two-streams --no-dense-leg --no-main-iteration --no-small-views --no-compact-queue-leg --serial --steps 1 --warmup 1"
for v in "$@"; do
  cp multiview-simulation_amd/libmvsim_$v.so multiview-simulation_amd/libmvsim.so
  echo "[$v]"
  rm -rf gpurun_out/pi && mkdir -p gpurun_out/pi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d gpurun_out/pi -o run -- python3 $B > gpurun_out/pi.log 2>&1
  python3 tools/pmc_insts.py gpurun_out/pi | grep -i "kernel\|extract\|resolve"
  rm -rf gpurun_out/pi && mkdir -p gpurun_out/pi
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d gpurun_out/pi -o run -- python3 $B > gpurun_out/pi.log 2>&1
  python3 tools/pmc_sq_report.py gpurun_out/pi 2>/dev/null | grep -i "kernel\|extract\|resolve" || true
  rm -rf gpurun_out/pi
done
cp gpurun_out/libmvsim_keep.so multiview-simulation_amd/libmvsim.so
