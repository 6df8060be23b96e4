export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
for b in 0 16384 8192 4096 2048; do echo -n "bpj_total $b: "; FRUITS_SEL_BPJ=$b python tools/select_bench.py 64 10 2>&1 | tail -1; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 tools/select_bench.py 64 5 > $O/trace.log 2>&1
python3 - <<'PY' $O
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "select_" in r["Kernel_Name"]:
            d[r["Kernel_Name"].split("fr::")[1].split("(")[0][:40]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v.sort()
    print(f"{k:42s} n={len(v):4d} median {v[len(v)//2]:8.1f} us  max {v[-1]:8.1f}  sum/call {sum(v)/7:8.1f}")
PY
