"""Cycle stamps inside gemm_kernel / gemm_ws_kernel for the LOW-RESOLUTION problems of the step (profiling build, see ws_stamp.py).
    python tools/probes/gemm_stamp.py        # fixed list of (shape, cfg) below, split-K off"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from mvdfusion_amd import hip

hip.LIB_PATHS["f16"] = os.path.join(ROOT, "tools", "probes", "libmvd_hip_stamp.so")
NAMES = ["prologue", "first k-tile wait", "k-loop", "end barrier", "acc -> LDS", "pass 1", "pass 2", "stores acknowledged", "statistics + exit"]
g = torch.Generator().manual_seed(0)
# name: (B, H, Cin, N, conv)
CASES = {"dense 2048x640x640": (8, 16, 640, 640, False), "dense 512x1280x1280": (8, 8, 1280, 1280, False),
         "dense 8192x320x320": (8, 32, 320, 320, False), "conv 2048x640x5760": (8, 16, 640, 640, True),
         "conv 512x1280x11520": (8, 8, 1280, 1280, True)}
CFGS = [(9, "64x64 ring4"), (3, "64x64 pipe"), (25, "128x128 ring4"), (41, "128x80 ring4"), (47, "128x80 ws")]
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
    for cfg, label in CFGS:
        try:
            for rep in range(3):
                ws.zero_()
                hip.gemm(A, W, out, prec=4, workspace=ws, cfg=cfg, splitk=1, res=R, **kw)
        except Exception as e:
            continue
        torch.cuda.synchronize()
        d = ws.view(torch.int64)[:128].cpu().tolist()
        t = d[0:10]
        if t[0] == 0 or t[8] == 0:
            continue
        seq = [t[0], t[1], t[2], t[3], t[4], t[5], t[9], t[6], t[7], t[8]]
        ph = [seq[i + 1] - seq[i] for i in range(9)]
        nkt = (9 * Cin if conv else Cin) // 32
        print(f"{name:22s} {label:14s}: total {t[8] - t[0]:6d} ({nkt} k-tiles, {ph[2] / nkt:5.0f} / k-tile) | " +
              " | ".join(f"{n} {v}" for n, v in zip(NAMES, ph)), flush=True)
