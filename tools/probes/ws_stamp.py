"""Cycle stamps inside gemm_ws_kernel (profiling build: tools/probes/stamp.sh compiles csrc/gemm.hip with -DMVD_STAMP into
tools/probes/libmvd_hip_stamp.so).  Prints, per consumer wavefront of workgroups 0 and 100, the cycles spent in each phase of ONE launch:
launch -> prologue issued -> first k-tile landed -> k-loop -> barrier -> accumulators in LDS -> epilogue stores issued -> stores
acknowledged -> statistics passes."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from mvdfusion_amd import hip

hip.LIB_PATHS["f16"] = os.path.join(ROOT, "tools", "probes", "libmvd_hip_stamp.so")
NAMES = ["prologue", "first k-tile wait", "k-loop", "end barrier", "acc -> LDS", "pass 1 + pass 2 (stores issued)", "stores acknowledged",
         "statistics + exit"]
g = torch.Generator().manual_seed(0)
CASES = {"conv32": (8, 32, 320, 320, True), "proj32": (8, 32, 320, 320, False), "ff2_32": (8, 32, 1280, 320, False)}
for name, (B, H, Cin, N, conv) in CASES.items():
    M = B * H * H
    A = hip.split_planes(torch.randn(M, Cin, generator=g).cuda())
    if conv:
        W = hip.pack_conv3x3((torch.randn(N, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).cuda(), torch.zeros(N).cuda())
        kw = dict(conv=dict(B=B, Hin=H, Win=H, Cin=Cin, Hout=H, Wout=H, stride=1, upsample=0))
    else:
        W = hip.pack_linear((torch.randn(N, Cin, generator=g) / math.sqrt(Cin)).cuda(), torch.zeros(N).cuda())
        kw = {}
    out = torch.empty(M, N, device="cuda")
    R = torch.randn(M, N, generator=g).cuda()
    ws = torch.zeros(1 << 20, device="cuda")
    for label, extra in (("out + res", dict(res=R)), ("out only", dict())):
        for cfg in (47, 79):
            for rep in range(3):
                ws.zero_()
                hip.gemm(A, W, out, prec=4, workspace=ws, cfg=cfg, splitk=1, **extra, **kw)
            torch.cuda.synchronize()
            d = ws.view(torch.int64)[:128].cpu().tolist()
            for blk in (0, 1):
                for w in (0, 3):
                    t = d[blk * 64 + w * 16: blk * 64 + w * 16 + 10]
                    if t[0] == 0:
                        continue
                    ph = [t[i + 1] - t[i] for i in range(8)]
                    print(f"{name} [{label}] cfg {cfg} wg {blk * 100} wave {w}: total {t[8] - t[0]:6d} (pass 1 {t[9] - t[5]}, pass 2 {t[6] - t[9]}) | " +
                          " | ".join(f"{n} {v}" for n, v in zip(NAMES, ph)), flush=True)
