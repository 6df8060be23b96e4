"""Condenses a rocprofv3 --kernel-trace of tools/fit_profile.py: the dispatches of the last
fr_select_ranks call and the per-kernel totals."""
import csv, glob, sys, collections
path = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"].split("(")[0].replace("void ", "")[:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
        r["Grid_Size_Y"]) for r in rows]
last = max(i for i, s in enumerate(seq) if "coswiss" in s[0] or "iss_" in s[0])
for s in seq[last:]:
    print(f"{s[0]:62s} {s[1]:9.1f} us  groups {s[2]}")
tot = collections.Counter()
for s in seq:
    tot[s[0]] += s[1]
print()
for k, v in tot.most_common(12):
    print(f"{k:62s} {v / 1e3:9.2f} ms total")
