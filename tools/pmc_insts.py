"""Per-kernel dynamic instruction counts from one rocprofv3 counter pass, e.g.

    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d DIR -o run \
        -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python tools/pmc_insts.py DIR > profiles/rNN_pmc_instructions.txt

Per wave averages; "valu_ms" = waves x VALU instructions x 4 cycles / (1024 SIMDs x 2.4 GHz): the time the chip needs
just to issue the kernel's vector instructions (a kernel whose duration is close to it is issue-bound, not HBM-bound).
"""
import collections, glob, sqlite3, sys

db = sorted(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True))[-1]
acc = collections.OrderedDict()
q = "select kernel_name, counter_name, value from counters_collection order by dispatch_id"
for name, cn, v in sqlite3.connect(db).execute(q):
    d = acc.setdefault(name, collections.Counter())
    d[cn] += v
    d["_n_" + cn] += 1
print(f"{'kernel':88s} {'launches':>8s} {'waves':>9s} {'VALU':>7s} {'SALU':>7s} {'LDS':>6s} {'VMEM_RD':>8s} {'VMEM_WR':>8s} {'valu_ms':>8s}")
for k, d in acc.items():
    n = d["_n_SQ_WAVES"] or 1
    w = d["SQ_WAVES"] / n
    if w < 1:
        continue
    per = lambda c: d[c] / d["SQ_WAVES"]
    valu_ms = w * per("SQ_INSTS_VALU") * 4 / (1024 * 2.4e9) * 1e3
    print(f"{k[:88]:88s} {n:8d} {w:9.0f} {per('SQ_INSTS_VALU'):7.0f} {per('SQ_INSTS_SALU'):7.0f} {per('SQ_INSTS_LDS'):6.0f} "
          f"{per('SQ_INSTS_VMEM_RD'):8.0f} {per('SQ_INSTS_VMEM_WR'):8.0f} {valu_ms:8.3f}")
