set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export MVSIM_OPTIONS="$1"
rm -rf gpurun_out/pi && mkdir -p gpurun_out/pi
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d gpurun_out/pi -o run -- python3 bench.py --no-cpu-baseline --no-end-to-end --no-size-1024 --no-two-streams --no-dense-leg --serial --steps 1 --warmup 1 > gpurun_out/pi.log 2>&1
python3 tools/pmc_insts.py gpurun_out/pi
rm -rf gpurun_out/pi
