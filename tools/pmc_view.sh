#!/bin/bash
# any PMC counters of one view at an arbitrary geometry: bash tools/pmc_view.sh "CNT1 CNT2" 512 512 512 31 31 31 1 [options...]
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
C="$1"; shift
O=gpurun_out/pmcv; rm -rf $O; mkdir -p $O
rocprofv3 --pmc $C -d $O/p -o run -- python3 tools/view_time.py "$@" > $O/log 2>&1
python3 - "$O" <<'PY'
import glob, sqlite3, sys
dbs = sorted(glob.glob(sys.argv[1] + "/p/**/*_results.db", recursive=True))
acc = {}
for name, cn, v in sqlite3.connect(dbs[-1]).execute("select kernel_name, counter_name, value from counters_collection"):
    a = acc.setdefault((name[:60], cn), [0, 0.0]); a[0] += 1; a[1] += float(v)
for (k, cn), (n, v) in sorted(acc.items()):
    print(f"{k:60s} {cn:36s} launches {n:3d}  mean {v / n:14.1f}")
PY
