"""Condenses a `rocprofv3 --kernel-trace` run: per (kernel name, grid) the dispatches, average /
min duration and the share of the whole; plus the span from the first start to the last end of
one repetition (python tools/trace_dispatches.py DIR [skip_first_n])."""
import collections, csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
acc = collections.OrderedDict()
for r in rows:
    key = (r["Kernel_Name"][:90], r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("VGPR_Count", r.get("Arch_VGPR_Count", "?")))
    acc.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in acc.values())
for (name, grid, vg), v in acc.items():
    print(f"{name:90s} grid {grid:>10s} vgpr {vg:>4s} n {len(v):4d} avg {sum(v)/len(v)/1e3:10.1f} us min {min(v)/1e3:10.1f} "
          f"share {100.0*sum(v)/tot:5.1f} %")
