"""Aggregate two rocprofv3 counter passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, run separately and without
any trace domain) into the per-kernel HBM traffic table committed under profiles/.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <views_profiled> > profiles/rNN_pmc_hbm_traffic.txt

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are
in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream, so the read column is multiplied by 2;
WRITE_SIZE is exact.  Values are means per launch.
"""
import csv, glob, sys
from collections import OrderedDict


def load(d, counter):
    acc = OrderedDict()
    dbs = sorted(glob.glob(d + "/**/*_results.db", recursive=True))
    if dbs:  # rocprofv3's default rocpd (sqlite) output
        import sqlite3
        q = "select kernel_name, value from counters_collection where counter_name = ? order by dispatch_id"
        for name, v in sqlite3.connect(dbs[-1]).execute(q, (counter,)):
            a = acc.setdefault(name, [0, 0.0])
            a[0] += 1
            a[1] += float(v)
        return acc
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        a = acc.setdefault(r["Kernel_Name"], [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


def main():
    fd, wd, views = sys.argv[1], sys.argv[2], int(sys.argv[3])
    rd, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    print("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes; no trace domains), "
          "python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline")
    print("# workload: 512^3 x 8 views, 31^3 PSF, inc 1.  Counter unit KiB; gfx950: FETCH_SIZE reports 1/2 of a wide "
          "coalesced read stream")
    print("# (MI355X_MICROARCH.md, HBM section) -> x2 correction applied to the read column.  Values are means per launch.")
    print(f"{'kernel':100s} {'launches':>8s} {'read_GB':>9s} {'write_GB':>9s}")
    total = 0.0
    for k, (n, v) in rd.items():
        r = 2.0 * v * 1024 / n / 1e9
        wn, wv = wr.get(k, [1, 0.0])
        w = wv * 1024 / max(wn, 1) / 1e9
        total += (r + w) * n
        print(f"{k[:100]:100s} {n:8d} {r:9.3f} {w:9.3f}")
    print(f"# total HBM traffic per view (all launches, {views} views profiled): {total / views:.2f} GB")


if __name__ == "__main__":
    main()
