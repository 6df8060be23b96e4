"""The idle gaps of the device during one Fruit.fit: python tools/fit_gaps.py DIR (a rocprofv3
--kernel-trace of tools/fit_profile.py) - the last fit's kernels, its span, busy time and the
largest gaps with the kernels on either side."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
bursts, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) > 5_000_000:
        bursts.append(cur); cur = []
    cur.append(b)
bursts.append(cur)
full = [b for b in bursts if len(b) > 100]
bu = full[-2] if len(full) > 1 else full[-1]
span = (int(bu[-1]["End_Timestamp"]) - int(bu[0]["Start_Timestamp"])) / 1e6
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in bu) / 1e6
print(f"kernels {len(bu)} span {span:.1f} ms busy {busy:.1f} ms idle {span - busy:.1f} ms")
gaps = []
for a, b in zip(bu, bu[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e6
    gaps.append((g, a["Kernel_Name"][:46], b["Kernel_Name"][:46], (int(a["End_Timestamp"]) - int(bu[0]["Start_Timestamp"])) / 1e6))
small = sum(g for g, *_ in gaps if g < 0.05)
print(f"gaps < 50 us: {sum(1 for g, *_ in gaps if g < 0.05)} totalling {small:.2f} ms")
for g, a, b, at in sorted(gaps, reverse=True)[:14]:
    print(f"  {g:6.2f} ms at {at:6.1f} ms  after {a:46s} before {b}")
if len(sys.argv) > 2:
    t0 = int(bu[0]["Start_Timestamp"])
    for r in bu[:int(sys.argv[2])]:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(st - t0) / 1e6:7.2f} ms +{(en - st) / 1e3:8.1f} us  {r['Kernel_Name'][:70]}")
