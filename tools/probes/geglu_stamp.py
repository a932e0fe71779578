"""Cycle stamps of the GEGLU projection GEMM (8192 x 2560 x 320, value|gate packed weights, planes output) -- the largest single kernel
family of the step -- per workgroup 0 / 100, consumer waves 0 / 3 (profiling build, see ws_stamp.py)."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from mvdfusion_amd import hip

hip.LIB_PATHS["f16"] = os.path.join(ROOT, "tools", "probes", "libmvd_hip_stamp.so")
NAMES = ["prologue", "first k-tile wait", "k-loop", "end barrier", "acc -> LDS", "GEGLU chunks", "-", "-", "exit"]
g = torch.Generator().manual_seed(0)
for (M, C) in ((8192, 320), (2048, 640)):
    A = hip.split_planes(torch.randn(M, C, generator=g).cuda())
    W = hip.pack_linear((torch.randn(8 * C, C, generator=g) / math.sqrt(C)).cuda(), torch.zeros(8 * C).cuda(), geglu=True)
    planes = torch.empty(M, 2 * 4 * C, dtype=torch.int16, device="cuda")
    ws = torch.zeros(1 << 22, device="cuda")
    for cfg, label in ((17, "128x128 plain (2 WG/CU)"), (19, "128x128 pipe"), (21, "128x128 staggered3"), (25, "128x128 ring4"), (1, "64x64 plain"), (9, "64x64 ring4")):
        try:
            for rep in range(3):
                ws.zero_()
                hip.gemm(A, W, None, prec=4, workspace=ws, cfg=cfg, splitk=1, epi=hip.EPI_GEGLU, out_planes=planes)
        except Exception as e:
            print("skip", cfg, str(e)[:80])
            continue
        torch.cuda.synchronize()
        d = ws.view(torch.int64)[:128].cpu().tolist()
        for blk in (0, 1):
            t = d[blk * 64: blk * 64 + 10]
            if t[0] == 0 or t[8] == 0:
                continue
            seq = [t[0], t[1], t[2], t[3], t[4], t[5], t[9], t[6], t[7], t[8]]
            ph = [seq[i + 1] - seq[i] for i in range(9)]
            print(f"GEGLU {M}x{8 * C}x{C} {label:24s} wg {blk * 100}: total {t[8] - t[0]:6d} ({C // 32} k-tiles, {ph[2] / (C // 32):5.0f} / k-tile) | " +
                  " | ".join(f"{n} {v}" for n, v in zip(NAMES, ph) if n != "-"), flush=True)
