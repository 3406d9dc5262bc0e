"""Print the rocprofv3 --kernel-trace --stats summary found under a directory (csv or rocpd .db output);
with a third argument, also write it as a kernel_stats csv (for profiles/)."""
import csv, glob, sys

d = sys.argv[1]
rows = []
dbs = sorted(glob.glob(d + "/**/*_results.db", recursive=True))
if dbs:
    import sqlite3
    q = ("select name, total_calls, total_duration, average, percentage from top_kernels")
    for n, c, t, a, p in sqlite3.connect(dbs[-1]).execute(q):  # durations in microseconds
        rows.append({"Name": n, "Calls": c, "TotalDurationNs": int(t * 1e3), "AverageNs": a * 1e3, "Percentage": p})
else:
    f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(f"{r['Name'][:96]:96s} n={int(r['Calls']):>4d} avg_us={float(r['AverageNs'])/1e3:9.1f} "
          f"{float(r['Percentage']):6.2f}%")
if len(sys.argv) > 3:
    with open(sys.argv[3], "w", newline="") as o:
        w = csv.DictWriter(o, fieldnames=["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"],
                           quoting=csv.QUOTE_NONNUMERIC, extrasaction="ignore")
        w.writeheader()
        w.writerows(rows)
