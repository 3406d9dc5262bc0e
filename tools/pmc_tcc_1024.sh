set -e
mkdir -p gpurun_out
for SET in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_READ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE"; do
  echo "### $SET"
  bash tools/pmc_view.sh "$SET" 1024 1024 1024 31 31 31 1 gt=phantom2x | grep -v "rocclr\|plane_flags\|reduce_partials\|r2c"
done > gpurun_out/pmc1024_tcc.txt 2>&1
python3 tools/view_time.py 1024 1024 1024 31 31 31 1 gt=phantom2x >> gpurun_out/pmc1024_tcc.txt
