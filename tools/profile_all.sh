#!/bin/bash
# Collect the rocprofv3 artefacts of profiles/ for the current build (run on the GPU box through gpurun):
#   bash tools/profile_all.sh r03_x
# kernel trace + stats, then the PMC passes in runs of their own (no trace domains beside --pmc), as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass.
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/prof_$TAG
rm -rf $O && mkdir -p $O
# --serial: the library's default overlaps off, one kernel at a time, so that per-kernel durations add up to the stage times
BENCH="bench.py --serial --no-cpu-baseline --no-end-to-end --no-size-1024 --no-two-streams --no-dense-leg"
python3 bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/trace -o run -- python3 $BENCH > $O/trace.log 2>&1
python3 tools/kstats.py $O/trace 20 $O/kernel_stats.csv > $O/kernel_stats.txt
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o run -- python3 $BENCH --steps 1 --warmup 1 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o run -- python3 $BENCH --steps 1 --warmup 1 > $O/write.log 2>&1
python3 tools/pmc_traffic.py $O/fetch $O/write 16 --json $O/traffic.json > $O/pmc_hbm_traffic.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/insts -o run -- python3 $BENCH --steps 1 --warmup 1 > $O/insts.log 2>&1
python3 tools/pmc_insts.py $O/insts > $O/pmc_instructions.txt
rm -rf $O/trace $O/fetch $O/write $O/insts
cat $O/kernel_stats.txt; cat $O/pmc_hbm_traffic.txt; cat $O/pmc_instructions.txt
