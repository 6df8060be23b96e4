"""Summarises rocprofv3 --pmc csv output: mean per dispatch of each counter for kernels matching a pattern."""
import csv, glob, sys, collections
root, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if pat in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:28s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
