# per-kernel average durations of the bench under rocprofv3 --kernel-trace --stats: bash tools/ktrace.sh "<MVSIM_OPTIONS>"
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export MVSIM_OPTIONS="$1"
rm -rf gpurun_out/kt && mkdir -p gpurun_out/kt
rocprofv3 --kernel-trace --stats -d gpurun_out/kt -o run -- python3 bench.py --no-cpu-baseline --no-end-to-end --no-size-1024 --no-two-streams > gpurun_out/kt.log 2>&1
python3 tools/kstats.py gpurun_out/kt 14
rm -rf gpurun_out/kt
