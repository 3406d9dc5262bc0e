"""Per-kernel SQ wait / issue counters from one rocprofv3 pass (tools/profile_all.sh):
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS ...
    python tools/pmc_sq_report.py DIR > profiles/rNN_sq_counters.txt
wait_any = share of the waves' cycles spent waiting for anything (memory, LDS, barriers); active = share with an instruction in flight."""
import collections, glob, sqlite3, sys
db = sorted(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True))[-1]
acc = collections.OrderedDict()
for name, cn, v in sqlite3.connect(db).execute("select kernel_name, counter_name, value from counters_collection order by dispatch_id"):
    d = acc.setdefault(name, collections.Counter()); d[cn] += v; d["_n_" + cn] += 1
print(f"{'kernel':84s} {'launches':>8s} {'waves':>8s} {'cycles/wave':>11s} {'wait_any':>8s} {'wait_inst':>9s} {'active':>7s} {'VALU/wave':>9s} {'LDS/wave':>8s} {'bank_conf/LDS':>13s}")
for k, d in acc.items():
    n = d["_n_SQ_WAVES"] or 1
    w = d["SQ_WAVES"]
    if w / n < 16:
        continue
    cyc = d["SQ_WAVE_CYCLES"] or 1
    print(f"{k[:84]:84s} {n:8d} {w / n:8.0f} {cyc / w:11.0f} {d['SQ_WAIT_ANY'] / cyc:8.2f} {d['SQ_WAIT_INST_ANY'] / cyc:9.2f} {d['SQ_ACTIVE_INST_ANY'] / cyc:7.2f} "
          f"{d['SQ_INSTS_VALU'] / w:9.0f} {d['SQ_INSTS_LDS'] / w:8.0f} {d['SQ_LDS_BANK_CONFLICT'] / max(1, d['SQ_INSTS_LDS']):13.2f}")
