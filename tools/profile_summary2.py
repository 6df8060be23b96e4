"""Condenses a tools/profile_round2.sh run into gpurun_out/profiles_<tag>/ (copied to profiles/)."""
import collections, csv, glob, json, os, sys
root, tag = sys.argv[1], sys.argv[2]
out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "profiles_" + tag)
os.makedirs(out_dir, exist_ok=True)
KERNELS = ("iss_walk", "iss_fused", "coswiss")

def stats(sub):
    f = glob.glob(f"{root}/{sub}/**/*kernel_stats.csv", recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []

def pmc(sub):
    """Counter means per LAUNCH: a plan in pieces is several dispatches per launch (one per piece
    type, then gather_row_blocks - whose dispatches count the launches); else one.  Of a fused
    target (cfg3 / 4 / 5) only the fused walk's dispatches count (WalkCfg<..., TEAM 4, MODE 1, ...>):
    the fit of the bench pipeline materialises the fit sample's iterated sums with another
    instantiation of the same kernel (MODE 2: 1.4 GB written once for config 4) - averaged in,
    as the round-3 summaries did, it made the launches look 20 % lighter in instructions and
    2.4 x heavier in written bytes than they are."""
    acc = collections.defaultdict(list)
    launches = collections.Counter()
    fused_target = sub.startswith(("cfg3", "cfg4", "cfg5"))
    for f in glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if fused_target and "iss_fused" in r["Kernel_Name"] and ", 4, 1, " not in r["Kernel_Name"]:
                continue
            if any(k in r["Kernel_Name"] for k in KERNELS):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "gather_row_blocks" in r["Kernel_Name"]:
                launches[r["Counter_Name"]] += 1
    per = {k: (launches[k] if launches[k] else len(v)) for k, v in acc.items()}
    return {k: sum(v) / per[k] for k, v in acc.items()}, per

# headline: the driver's command
rows = stats("bench_trace")
summary = {"round": tag}
if rows:
    with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows)
    walk = [r for r in rows if "iss_walk" in r["Name"]]
    if walk:
        r = max(walk, key=lambda r: int(r["Calls"]))
        summary.update({"iss_walk_kernel": r["Name"], "iss_walk_calls": int(r["Calls"]),
                        "iss_walk_avg_ns": float(r["AverageNs"]), "iss_walk_min_ns": float(r["MinNs"]),
                        "iss_walk_max_ns": float(r["MaxNs"])})
fetch, nf = pmc("bench_fetch"); write, nw = pmc("bench_write")
if "FETCH_SIZE" in fetch and "WRITE_SIZE" in write:
    f_kb, w_kb = fetch["FETCH_SIZE"], write["WRITE_SIZE"]
    summary.update({"FETCH_SIZE_KiB_per_launch": f_kb, "WRITE_SIZE_KiB_per_launch": w_kb,
                    "read_bytes_per_launch_corrected": 2 * f_kb * 1024, "write_bytes_per_launch": w_kb * 1024,
                    "iss_walk_bytes_per_launch": 2 * f_kb * 1024 + w_kb * 1024,
                    "dispatches_sampled": min(nf["FETCH_SIZE"], nw["WRITE_SIZE"]),
                    "correction": "gfx950: FETCH_SIZE x2 for 16 B/lane coalesced reads (MI355X_MICROARCH.md, "
                                  "HBM); separate --pmc passes"})
with open(os.path.join(out_dir, "traffic.json"), "w") as f:
    json.dump(summary, f, indent=1)
print(json.dumps(summary, indent=1))

# the other kernels: per target one text summary
for t in ("cfg2", "cfg3", "cfg4", "cfg5", "cos1", "cos2"):
    rows = stats(f"{t}_trace")
    if not rows:
        continue
    lines = [f"# {t}: rocprofv3 --kernel-trace --stats and separate --pmc passes of `python tools/run_kernels.py {t}`"]
    lines.append("# kernel stats (Name, Calls, AverageNs, MinNs, MaxNs, Percentage)")
    main = None
    for r in rows:
        lines.append(f"{r['Name'][:110]:110s} calls {int(r['Calls']):4d} avg_ns {float(r['AverageNs']):12.0f} "
                     f"min {float(r['MinNs']):12.0f} max {float(r['MaxNs']):12.0f} pct {float(r['Percentage']):6.2f}")
        if (any(k in r["Name"] for k in KERNELS) and not ("iss_fused" in r["Name"] and t.startswith("cfg") and t != "cfg2"
                                                           and ", 4, 1, " not in r["Name"])
                and (main is None or float(r["TotalDurationNs"]) > float(main["TotalDurationNs"]))):
            main = r
    c = {}
    for i in range(1, 5):
        m, _ = pmc(f"{t}_pmc{i}")
        c.update(m)
    lines.append("# PMC means per dispatch of the walk / coswiss kernels")
    for k in sorted(c):
        lines.append(f"{k:28s} {c[k]:18.1f}")
    pieces = [r for r in rows if "gather_row_blocks" in r["Name"]]
    if pieces:
        n_launch = int(pieces[0]["Calls"])
        fused = [r for r in rows if any(k in r["Name"] for k in KERNELS)]
        # (the fit's materialising launch is of another instantiation: MODE 2)
        fused = [r for r in fused if ", 4, 1, " in r["Name"]]
        launch_ns = sum(float(r["TotalDurationNs"]) for r in fused) / n_launch
        lines.append(f"# a plan in pieces: {sum(int(r['Calls']) for r in fused) // n_launch} dispatches per launch, "
                     f"{launch_ns / 1e3:.1f} us of kernels per launch ({n_launch} launches)")
    if main is not None and "SQ_INSTS_VALU" in c:
        dur = launch_ns * 1e-9 if pieces else float(main["AverageNs"]) * 1e-9
        # issue roofline: one VALU instruction per SIMD per 4-cycle issue turn; 256 CUs x 4 SIMDs
        clk = 2.4e9
        valu_rate = c["SQ_INSTS_VALU"] / dur
        peak = 256 * 4 * clk / 4
        lines.append(f"# issue roofline: SQ_INSTS_VALU {c['SQ_INSTS_VALU']:.3e} per launch / {dur*1e6:.1f} us = "
                     f"{valu_rate:.3e} wave-instr/s; peak 256 CUs x 4 SIMDs x {clk/1e9:.1f} GHz / 4 cycles = {peak:.3e}; "
                     f"VALU issue fraction {valu_rate / peak:.3f}")
        if "SQ_INSTS_SALU" in c:
            lines.append(f"# SALU/VALU {c['SQ_INSTS_SALU'] / c['SQ_INSTS_VALU']:.2f}; "
                         f"SALU issue fraction (1 per cycle per CU) {c['SQ_INSTS_SALU'] / dur / (256 * clk):.3f}")
        if "SQ_WAVE_CYCLES" in c:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA"):
                if k in c:
                    lines.append(f"# {k} / SQ_WAVE_CYCLES = {c[k] / c['SQ_WAVE_CYCLES']:.3f}")
        if "GRBM_GUI_ACTIVE" in c:
            lines.append(f"# clock: GRBM_GUI_ACTIVE {c['GRBM_GUI_ACTIVE']:.3e} / 8 XCDs / {dur*1e6:.1f} us = "
                         f"{c['GRBM_GUI_ACTIVE'] / 8 / dur / 1e9:.2f} GHz while the kernels run")
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            lines.append(f"# HBM traffic per launch: read {2 * c['FETCH_SIZE'] * 1024 / 1e6:.1f} MB (FETCH_SIZE x2), "
                         f"write {c['WRITE_SIZE'] * 1024 / 1e6:.1f} MB")
    if t in ("cfg3", "cfg4", "cfg5") and main is not None and "SQ_INSTS_VALU" in c:
        issue = {}
        ip = os.path.join(out_dir, "fused_issue.json")
        if os.path.exists(ip):
            issue = json.load(open(ip))
        issue[t] = round(c["SQ_INSTS_VALU"] / dur / (256 * 4 * 2.4e9 / 4), 3)
        issue["source"] = (f"rocprofv3 --pmc SQ_INSTS_VALU over tools/run_kernels.py, {tag}: wave-instructions per "
                           "launch / kernel time / (256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles)")
        json.dump(issue, open(ip, "w"), indent=1)
    with open(os.path.join(out_dir, f"{tag}_{t}_summary.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines[-8:]))
