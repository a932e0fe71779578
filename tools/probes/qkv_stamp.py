"""Cycle stamps of the QKV projection GEMM (LayerNorm folded, q / k planes + V^T planes out) and, for comparison, the GEGLU projection of the
same level: where one 128x128 tile of the plain loop spends its life (profiling build: tools/probes/stamp.sh).
    python tools/probes/qkv_stamp.py"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.nn as nn
from mvdfusion_amd import hip

hip.LIB_PATHS["f16"] = os.path.join(ROOT, "tools", "probes", "libmvd_hip_stamp.so")
NAMES = ["prologue", "first k-tile wait", "k-loop", "end barrier", "acc -> LDS", "epilogue chunks", "-", "-", "exit"]
g = torch.Generator().manual_seed(0)
ws = torch.zeros(1 << 24, device="cuda")
for (M, C, L) in ((8192, 320, 1024), (2048, 640, 256), (512, 1280, 64)):
    heads = 8
    x = torch.randn(M, C, generator=g).cuda()
    xp = hip.split_planes(x)
    rs = hip.RowStats(M, C, "cuda")
    Wl = hip.pack_linear((torch.randn(C, C, generator=g) / math.sqrt(C)).cuda(), None)
    tp, tt = hip.planes_like(M, C, "cuda"), torch.empty(M, C, device="cuda")
    hip.gemm(xp, Wl, tt, out_planes=tp, row_stats=rs, workspace=ws)
    norm = nn.LayerNorm(C).cuda()
    for kind in ("qkv", "geglu"):
        if kind == "qkv":
            fold = hip.LnFold((torch.randn(3 * C, C, generator=g) / math.sqrt(C)).cuda(), None, norm)
            planes = hip.alloc_attn_planes(M // L, heads, L, C // heads, "cuda")
            kw = dict(epi=hip.EPI_QKV, qkv=dict(planes=planes, heads=heads, dhead=C // heads, L=L), ln=(rs, fold))
        else:
            fold = hip.LnFold((torch.randn(8 * C, C, generator=g) / math.sqrt(C)).cuda(), torch.zeros(8 * C, device="cuda"), norm, geglu=True)
            kw = dict(epi=hip.EPI_GEGLU, out_planes=hip.planes_like(M, 4 * C, "cuda"), ln=(rs, fold))
        for cfg, label in ((hip.make_cfg(1, 0), "128x128 plain"), (hip.make_cfg(0, 4), "64x64 ring4")):
            for rep in range(3):
                ws.zero_()
                hip.gemm(tp, fold.w, None, prec=3, workspace=ws, cfg=cfg, splitk=1, **kw)
            torch.cuda.synchronize()
            d = ws.view(torch.int64)[:128].cpu().tolist()
            for blk in (0, 1):
                t = d[blk * 64: blk * 64 + 10]
                if t[0] == 0 or t[8] == 0:
                    continue
                seq = [t[0], t[1], t[2], t[3], t[4], t[5], t[9], t[6], t[7], t[8]]
                ph = [seq[i + 1] - seq[i] for i in range(9)]
                print(f"{kind:5s} M={M} C={C} {label:14s} wg {blk * 100}: total {t[8] - t[0]:6d} ({C // 32} k-tiles) | " +
                      " | ".join(f"{n} {v}" for n, v in zip(NAMES, ph) if n != "-"), flush=True)
