#!/bin/bash
# Collect the rocprofv3 artefacts of profiles/ for the current build (run on the GPU box through gpurun):
#   bash tools/profile_all.sh r06_a
# Order matters: the PMC traffic passes come FIRST and write traffic.json keyed by the SHA of the kernel sources;
# bench.py runs afterwards with MVSIM_TRAFFIC_JSON pointing at it, so that the bench line of record carries roofline.traffic of
# the very build it measured.  Counter passes run on their own (no trace domains beside --pmc), FETCH_SIZE and WRITE_SIZE in
# separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
set -e
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/prof_$TAG
rm -rf $O && mkdir -p $O
# --serial: the library's default overlaps off, one kernel at a time, so that per-kernel durations add up to the stage times
BENCH="bench.py --serial --no-cpu-baseline --no-end-to-end --no-size-1024 --no-two-streams --no-dense-leg --no-main-iteration --no-small-views --no-compact-queue-leg"
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o run -- python3 $BENCH --steps 1 --warmup 1 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o run -- python3 $BENCH --steps 1 --warmup 1 > $O/write.log 2>&1
python3 tools/pmc_traffic.py $O/fetch $O/write 16 --json $O/traffic.json > $O/pmc_hbm_traffic.txt
echo "traffic done"
# the same two counter passes for one 1024^3 view on the data bench.py's size_1024 record runs on (6 launches of every kernel: 2 warm-up + 4 timed)
V1024="tools/view_time.py 1024 1024 1024 31 31 31 1 gt=phantom2x"
rocprofv3 --pmc FETCH_SIZE -d $O/fetch1k -o run -- python3 $V1024 > $O/fetch1k.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write1k -o run -- python3 $V1024 > $O/write1k.log 2>&1
python3 tools/pmc_traffic.py $O/fetch1k $O/write1k 6 --json $O/traffic_1024.json --size 1024 --psf 31 --inc 1 --workload "1024^3, one view at a time (6 launches), 31^3 PSF, inc 1, the 512^3 phantom up-sampled 2x" --command "python3 $V1024" > $O/pmc_hbm_traffic_1024.txt
rm -rf $O/fetch1k $O/write1k
echo "traffic 1024 done"
MVSIM_TRAFFIC_JSON=$O/traffic.json MVSIM_TRAFFIC_JSON_1024=$O/traffic_1024.json python3 bench.py > $O/bench.json 2> $O/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats -d $O/trace -o run -- python3 $BENCH > $O/trace.log 2>&1
python3 tools/kstats.py $O/trace 20 $O/kernel_stats.csv > $O/kernel_stats.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/insts -o run -- python3 $BENCH --steps 1 --warmup 1 > $O/insts.log 2>&1
python3 tools/pmc_insts.py $O/insts > $O/pmc_instructions.txt
echo "counters done"
# SQ wait / issue counters of the convolution passes, the fused rotate kernel and the sampler
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS -d $O/sq -o run -- python3 $BENCH --steps 1 --warmup 1 > $O/sq.log 2>&1
python3 tools/pmc_sq_report.py $O/sq > $O/sq_counters.txt
# the whole iteration of main's loop (mvsim_simulate_iteration_dev): the rotate-back kernels and makeIsotropic
rocprofv3 --kernel-trace --stats -d $O/trace_it -o run -- python3 bench.py --serial --no-cpu-baseline --no-end-to-end --no-size-1024 --no-two-streams --no-dense-leg --no-small-views --no-compact-queue-leg > $O/trace_it.log 2>&1
python3 tools/kstats.py $O/trace_it 24 $O/main_iteration_kernel_stats.csv > $O/main_iteration_kernel_stats.txt
echo "iteration trace done"
# HBM traffic of the views of BASELINE configs[3] and configs[4]: same two counter passes, one view at a time
{
  for W in "1024 1024 1024 31 31 63 4" "2048 2048 512 63 63 63 3"; do
    rocprofv3 --pmc FETCH_SIZE -d $O/fx -o run -- python3 tools/view_time.py $W > $O/fx.log 2>&1
    rocprofv3 --pmc WRITE_SIZE -d $O/wx -o run -- python3 tools/view_time.py $W > $O/wx.log 2>&1
    python3 tools/pmc_traffic.py $O/fx $O/wx 6 --workload "$W (Nx Ny Nz Kx Ky Kz inc), one view at a time (6 launches), compactly supported volume" --command "python3 tools/view_time.py $W"
    rm -rf $O/fx $O/wx
    echo
  done
} > $O/other_sizes_traffic.txt 2>/dev/null
python3 tools/small_views.py c0 ref 256 64 lanes=1,4 > $O/small_views.txt 2>/dev/null
python3 tools/queue_share.py 512 > $O/queue_share.txt 2>/dev/null
{
  python3 tools/view_time.py 128 128 128 15 15 15 1
  python3 tools/view_time.py 289 289 289 51 51 51 3
  python3 tools/view_time.py 1024 1024 1024 31 31 63 4
  python3 tools/view_time.py 1024 1024 1024 15 15 41 4
  python3 tools/view_time.py 2048 2048 512 63 63 63 3
  python3 tools/view_time.py 2048 2048 512 63 63 63 1
} > $O/other_sizes.txt 2>/dev/null
rm -rf $O/trace $O/fetch $O/write $O/insts $O/sq $O/trace_it
cat $O/kernel_stats.txt; cat $O/pmc_hbm_traffic.txt; cat $O/pmc_instructions.txt; cat $O/other_sizes.txt
