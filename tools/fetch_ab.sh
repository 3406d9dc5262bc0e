cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/fab; rm -rf $O; mkdir -p $O
for e in 0 1; do
rocprofv3 --pmc FETCH_SIZE -d $O/f$e -o run -- python3 tools/view_time.py 512 512 512 31 31 31 1 exp=$e > $O/f$e.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/g$e -o run -- python3 tools/view_time.py 1024 1024 1024 31 31 63 4 exp=$e > $O/g$e.log 2>&1
done
python3 - <<'PY'
import glob, sqlite3
for d in ("f0","f1","g0","g1"):
    dbs = sorted(glob.glob(f"gpurun_out/fab/{d}/**/*_results.db", recursive=True))
    acc = {}
    for name, v in sqlite3.connect(dbs[-1]).execute("select kernel_name, value from counters_collection where counter_name = 'FETCH_SIZE'"):
        a = acc.setdefault(name[:70], [0, 0.0]); a[0] += 1; a[1] += float(v)
    for k, (n, v) in acc.items():
        if "zconv" in k or "k_fft_lines" in k: print(d, k, n, f"{2 * v / n * 1024 / 1e9:.3f} GB/launch")
PY
