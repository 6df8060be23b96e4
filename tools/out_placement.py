"""`python tools/out_placement.py`: how much the time of a materialising launch depends on WHERE the
driver places its output - the same kernel on four fresh allocations of the output (a dummy allocation
in between shifts the next one), with the plane stride padded by 0 / 4 KB / 32 KB + 256 B / 512 KB + 4 KB.
Large outputs (the 48-word launch: 805 MB, of_weight(4,2): 1.9 GB) move by 10-25 % between placements
whatever the padding; the 18-plane headline (302 MB) does not."""
import sys, os
sys.path.insert(0, ".")
import numpy as np, torch
import fruits_amd as fr
import bench
N, T = 2048, 1024
Xd = bench._device_batch(torch, (N, 3, T), 0)
w15 = fr.words.of_weight(2, dim=3)
cases = (("w48", fr.ISS([w15[i % 15] for i in range(48)]), True), ("w4", fr.ISS(fr.words.of_weight(4, dim=2), mode=fr.ISSMode.EXTENDED), False),
         ("w2 headline", fr.ISS(w15, mode=fr.ISSMode.EXTENDED), True))
for name, iss, prep in cases:
    plan = iss._plan(0, len(iss.words))
    if prep:
        plan.prepare(N, T)
    K = plan.rows
    for pad in (0, 512, 4096 + 32, 65536 + 544):
        ts = []
        keep = []
        for trial in range(4):
            flat = torch.empty(K * (N * T + pad), dtype=torch.float64, device="cuda")
            t = bench._event_time_us(torch, lambda: plan.run(Xd, None, out=flat, strides=(N * T + pad, T)))
            ts.append(round(t, 1))
            del flat
            torch.cuda.empty_cache()
            keep.append(torch.empty((trial + 1) * 37_000_001, dtype=torch.uint8, device="cuda"))
        del keep
        torch.cuda.empty_cache()
        print(name, "K", K, "pad elems", pad, ts, flush=True)
