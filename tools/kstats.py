"""Print the rocprofv3 kernel_stats.csv found under a directory (development aid)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(f"{r['Name'][:96]:96s} n={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} {float(r['Percentage']):6.2f}%")
