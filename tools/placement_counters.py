"""Groups the per-dispatch counters of a `rocprofv3 --pmc ... -- python tools/placement_probe.py`
run by output buffer: python tools/placement_counters.py DIR buffers launches [kernel substring]."""
import collections, csv, glob, sys
d, n_buf, n_launch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sub = sys.argv[4] if len(sys.argv) > 4 else "iss_walk"
rows = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
disp = collections.OrderedDict()
for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
    if sub in r["Kernel_Name"]:
        e = disp.setdefault(r["Dispatch_Id"], {"us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = list(disp)[-n_buf * n_launch:]
names = sorted({k for i in ids for k in disp[i] if k != "us"})
print("buffer  us(median)  " + "  ".join(names))
for b in range(n_buf):
    grp = [disp[i] for i in ids[b * n_launch:(b + 1) * n_launch]]
    us = sorted(g["us"] for g in grp)[len(grp) // 2]
    print(f"{b:6d}  {us:10.1f}  " + "  ".join(f"{sum(g.get(k, 0.0) for g in grp) / len(grp):.4g}" for k in names))
