"""Compiler resource usage (VGPRs, SGPRs, spills, scratch, occupancy) of every kernel instance of
one translation unit:  python tools/resource_usage.py walk_inst.hip -DWALK_MODE=1 -DWALK_LV=6"""
import re, subprocess, sys, shutil
src = sys.argv[1]
flags = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
       "-Rpass-analysis=kernel-resource-usage", "-x", "hip", "-c", "fruits_amd/csrc/" + src,
       "-o", "/tmp/_ru.o"] + flags
run = subprocess.run(cmd, capture_output=True, text=True)
err = run.stderr
if run.returncode != 0:
    sys.exit("compilation failed:\n" + "\n".join(l for l in err.splitlines() if "error" in l)[:4000])
filt = shutil.which("c++filt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key, short in (("TotalSGPRs", "sgpr"), ("VGPRs", "vgpr"), (r"ScratchSize \[bytes/lane\]", "scratch"),
                       (r"Occupancy \[waves/SIMD\]", "occ"), ("SGPRs Spill", "sspill"), ("VGPRs Spill", "vspill")):
        m = re.search(r"\s" + key + r": (\d+)", line)
        if m and cur is not None and short not in cur:
            cur[short] = int(m.group(1))
for r in rows:
    n = subprocess.run([filt, r["name"]], capture_output=True, text=True).stdout.strip()
    n = re.sub(r"^void fr::", "", n)
    n = re.sub(r"\(fr::IssArgs\)$", "", n)
    print(f"{n:90s} vgpr {r.get('vgpr'):4d} sgpr {r.get('sgpr'):4d} sspill {r.get('sspill'):3d} "
          f"vspill {r.get('vspill'):3d} scratch {r.get('scratch'):4d} occ {r.get('occ')}")
