#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration for the y passes' access shape (tools/microbench/fetch_calib.hip); run on the GPU box:
#   bash tools/fetch_calib.sh > gpurun_out/fetch_calib.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
B=tools/microbench/fetch_calib
[ -x $B ] || hipcc --offload-arch=gfx950 -O3 $B.hip -o $B
O=gpurun_out/fcal; rm -rf $O; mkdir -p $O
echo "## plain run (HIP events, 3 launches each)"
$B 3
echo
echo "## counters available that look at the L2's memory side"
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_REQ[A-Z0-9_]*\|TCC_HIT[A-Z0-9_]*\|TCC_MISS[A-Z0-9_]*\|FETCH_SIZE\|WRITE_SIZE\|TCC_BUBBLE[A-Z0-9_]*\|TCC_READ[A-Z0-9_]*" | sort -u | tr '\n' ' '
echo
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_READ_sum"; do
  D=$O/$(echo $SET | tr ' ' '_')
  if rocprofv3 --pmc $SET -d $D -o run -- $B 1 > $D.log 2>&1; then
    echo "## --pmc $SET  (two launches per line of the plain run, in its order; value per launch)"
    python3 - "$D" <<'PY'
import glob, sqlite3, sys
dbs = sorted(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True))
rows = {}
for did, name, cn, v in sqlite3.connect(dbs[-1]).execute("select dispatch_id, kernel_name, counter_name, value from counters_collection order by dispatch_id"):
    rows.setdefault(did, [name[:40], {}])[1][cn] = rows.get(did, [None, {}])[1].get(cn, 0.0) + float(v)
for did in sorted(rows):
    name, c = rows[did]
    print(f"{did:4d} {name:40s} " + "  ".join(f"{k}={v:.0f}" for k, v in sorted(c.items())))
PY
  else
    echo "## --pmc $SET: not collected"; tail -3 $D.log
  fi
done
