"""Aggregate two rocprofv3 counter passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, run separately and without
any trace domain) into the per-kernel HBM traffic table committed under profiles/.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <views_profiled> [--json profiles/rNN_traffic.json
                                 --size 512 --psf 31 --inc 1] > profiles/rNN_pmc_hbm_traffic.txt

With --json the per-view, per-stage byte totals are also written as the record bench.py reads (`roofline.traffic`,
`roofline.hbm_measured`), keyed by the workload and by the SHA of the kernel sources they were measured on.

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


def stage_of(kernel: str) -> str:
    k = kernel
    if "k_rotate" in k or "k_attenuate" in k:
        return "rotate_attenuate"
    if "k_extract" in k or "k_poisson" in k:
        return "extract_poisson"
    if "fft::" in k or "k_reduce_partials" in k or "k_adjust" in k or "k_sum" in k or "k_stencil" in k or "k_pad" in k or "k_cmul" in k or "k_crop" in k or "k_psf" in k:
        return "convolve"
    return "other"


def main():
    argv = sys.argv[1:]
    opts = {}
    while len(argv) > 3:
        opts[argv[3].lstrip("-")] = argv[4]
        del argv[3:5]
    fd, wd, views = argv[0], argv[1], int(argv[2])
    rd, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    print("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes; no trace domains), "
          + opts.get("command", "python3 bench.py --serial --steps 1 --warmup 1 --no-cpu-baseline"))
    print("# workload: " + opts.get("workload", "512^3 x 8 views, 31^3 PSF, inc 1") + ".  Counter unit KiB; gfx950: FETCH_SIZE reports 1/2 of a wide "
          "coalesced read stream")
    print("# (MI355X_MICROARCH.md, HBM section) -> x2 correction applied to the read column.  Values are means per launch.")
    print(f"{'kernel':100s} {'launches':>8s} {'read_GB':>9s} {'write_GB':>9s}")
    total = 0.0
    stages, per_kernel = {}, {}
    for k, (n, v) in rd.items():
        r = 2.0 * v * 1024 / n / 1e9
        wn, wv = wr.get(k, [1, 0.0])
        w = wv * 1024 / max(wn, 1) / 1e9
        total += (r + w) * n
        stages[stage_of(k)] = stages.get(stage_of(k), 0.0) + (r + w) * n * 1e9 / views
        per_kernel[k[:120]] = {"launches": n, "read_bytes": r * 1e9, "write_bytes": w * 1e9}
        print(f"{k[:100]:100s} {n:8d} {r:9.3f} {w:9.3f}")
    print(f"# total HBM traffic per view (all launches, {views} views profiled): {total / views:.2f} GB")
    if "json" in opts:
        import importlib, json, os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        sha = importlib.import_module("multiview-simulation_amd.build").source_sha()
        rec = {"kernel_sha": sha, "workload": {"size": int(opts.get("size", 512)), "psf": int(opts.get("psf", 31)), "inc": int(opts.get("inc", 1))},
               "views_profiled": views, "per_view_bytes": {k: v for k, v in stages.items() if k != "other"},
               "per_kernel": per_kernel,
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs (no trace domains) of `"
                         + opts.get("command", "python3 bench.py --serial --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end --no-size-1024 --no-two-streams") + "`; counter unit "
                         "KiB; FETCH_SIZE x2 (gfx950 reports half of a wide coalesced read stream, MI355X_MICROARCH.md)"}
        json.dump(rec, open(opts["json"], "w"), indent=1)
        print(f"# wrote {opts['json']} (kernel_sha {sha})")


if __name__ == "__main__":
    main()
