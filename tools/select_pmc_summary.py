"""`python tools/select_pmc_summary.py DIR [DIR ...]`: the SQ counters of the selection kernels from
rocprofv3 --pmc passes over tools/select_bench.py (DIR/pmc*/**/counter_collection.csv): per kernel
the largest dispatch's counters and the instructions per 64 ELEMENTS of the block (select_bench.py's
default block: 64 x 2048 x 1024 elements, three differencing orders, six jobs per row)."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(d + "/pmc*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "select_hist" not in k and "select_gather" not in k and "select_small" not in k:
                continue
            name = k.split("fr::")[1].split("(")[0]
            c = r["Counter_Name"]
            agg[name][c] = max(agg[name][c], float(r["Counter_Value"]))
    print(f"== {d}")
    for name, c in sorted(agg.items()):
        w = 64 * 2048 * 1024 / 64 if "small" not in name else (c.get("SQ_WAVES", 0) or 1)
        unit = "per 64 elements" if "small" not in name else "per wave"
        print(f"{name:30s} waves {int(c.get('SQ_WAVES', 0)):7d}  {unit}: VALU {c.get('SQ_INSTS_VALU', 0) / w:7.1f} "
              f"SALU {c.get('SQ_INSTS_SALU', 0) / w:7.1f} LDS {c.get('SQ_INSTS_LDS', 0) / w:5.1f} "
              f"VMEM_RD {c.get('SQ_INSTS_VMEM_RD', 0) / w:4.1f}  LDS_BANK_CONFLICT {c.get('SQ_LDS_BANK_CONFLICT', 0):.3g} "
              f"WAIT_INST_LDS {c.get('SQ_WAIT_INST_LDS', 0):.3g} ACTIVE_INST_VALU {c.get('SQ_ACTIVE_INST_VALU', 0):.3g} "
              f"WAVE_CYCLES {c.get('SQ_WAVE_CYCLES', 0):.3g}")
