# Direct stencil against its fp32 roofline, with rocprofv3 kernel stats: bash tools/stencil_profile.sh
# Writes gpurun_out/r03_stencil_*: the K sweep (tools/stencil_bench.py), and for 15^3 / 31^3 (512^3 volume) and 63^3 (256^3
# sub-volume) the bench.py --conv-method 2 line plus the kernel-trace summary of that same command.
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 tools/stencil_bench.py --json gpurun_out/r03_stencil_bench.json > gpurun_out/r03_stencil_bench.txt 2>&1
for cfg in "15 512" "31 512" "63 256"; do
  set -- $cfg
  k=$1; n=$2
  rm -rf gpurun_out/st_$k
  rocprofv3 --kernel-trace --stats -d gpurun_out/st_$k -o run -- python3 bench.py --conv-method 2 --psf $k --size $n --steps 2 --warmup 1 --serial --no-cpu-baseline > gpurun_out/r03_stencil_bench_K$k.json 2> gpurun_out/r03_stencil_K$k.err
  python3 tools/kstats.py gpurun_out/st_$k 8 gpurun_out/r03_stencil_K${k}_kernel_stats.csv > gpurun_out/r03_stencil_K${k}_kernel_stats.txt
  rm -rf gpurun_out/st_$k
done
