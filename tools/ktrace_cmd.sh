# per-kernel average durations of any python tool under rocprofv3 --kernel-trace --stats:
#   bash tools/ktrace_cmd.sh TAG tools/small_views.py ref lanes=1      -> gpurun_out/TAG_kernel_stats.csv, printed top 24
set -e
tag="$1"; shift
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/kt_$tag && mkdir -p gpurun_out/kt_$tag
rocprofv3 --kernel-trace --stats -d gpurun_out/kt_$tag -o run -- python3 "$@" > gpurun_out/${tag}_run.log 2>&1
python3 tools/kstats.py gpurun_out/kt_$tag 24 gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/kt_$tag
