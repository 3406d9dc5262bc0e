# SQ wait/issue counters of the kernels matching a pattern: bash tools/pmc_sq.sh "<MVSIM_OPTIONS>" <pattern>
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export MVSIM_OPTIONS="$1"
rm -rf gpurun_out/psq && mkdir -p gpurun_out/psq
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS -d gpurun_out/psq -o run -- python3 bench.py --no-cpu-baseline --no-end-to-end --no-size-1024 --no-two-streams --no-dense-leg --serial --steps 1 --warmup 1 > gpurun_out/psq.log 2>&1
python3 - "$2" <<'PY'
import glob, sqlite3, sys, collections
db = sorted(glob.glob("gpurun_out/psq/**/*_results.db", recursive=True))[-1]
acc = collections.OrderedDict()
for name, cn, v in sqlite3.connect(db).execute("select kernel_name, counter_name, value from counters_collection order by dispatch_id"):
    if sys.argv[1] not in name: continue
    d = acc.setdefault(name[:60], collections.Counter()); d[cn] += v; d["_n_" + cn] += 1
for k, d in acc.items():
    n = d["_n_SQ_WAVES"]
    w = d["SQ_WAVES"] / n
    print(k, f"launches {n} waves {w:.0f}")
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT"):
        print(f"   {c:24s} per wave {d[c] / d['SQ_WAVES']:12.0f}")
PY
rm -rf gpurun_out/psq
