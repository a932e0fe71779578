"""mvd_gemm with WARM weights (same packed weight every launch: Infinity-Cache resident) vs COLD weights (launches rotate
over enough distinct copies of the packed weight that every launch streams it from HBM, as in a real DDIM step whose 4 GB
weight set never stays cached).  Per kernel configuration (include/mvd_hip.h: mvd_gemm_desc.cfg).

    python tools/gemm_cold.py [name-substring] [cfgs=0,5,6,...]
"""
import copy
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvdfusion_amd import hip

SHAPES = [  # (name, M, N, K, conv(B,H,Cin) or None)
    ("proj32", 8192, 320, 320, None), ("qkv32", 8192, 960, 320, None), ("ff1_32", 8192, 2560, 320, None),
    ("ff2_32", 8192, 320, 1280, None), ("conv32", 8192, 320, 2880, (8, 32, 320)),
    ("proj16", 2048, 640, 640, None), ("ff1_16", 2048, 5120, 640, None), ("ff2_16", 2048, 640, 2560, None),
    ("conv16", 2048, 640, 5760, (8, 16, 640)),
    ("proj8", 512, 1280, 1280, None), ("qkv8", 512, 3840, 1280, None), ("ff1_8", 512, 10240, 1280, None),
    ("ff2_8", 512, 1280, 5120, None), ("conv8", 512, 1280, 11520, (8, 8, 1280)),
    ("conv4", 128, 1280, 11520, (8, 4, 1280)), ("proj4", 128, 1280, 1280, None),
]
COLD_BYTES = 640 << 20


def bench(fn, reps):
    g = hip.Graph()
    with g:
        for i in range(reps):
            fn(i)
    g.launch()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = hip.Event(), hip.Event()
        e0.record()
        g.launch()
        e1.record()
        best = min(best, e0.elapsed_ms(e1) / reps)
    return best * 1e3


def main():
    only = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("cfgs=") else None
    cfgs = [0] + list(hip.GEMM_CONFIGS)
    for a in sys.argv[1:]:
        if a.startswith("cfgs="):
            cfgs = [int(c) for c in a[5:].split(",")]
    ws = torch.empty(64 * 1024 * 1024, device="cuda")
    print(f"{'shape':8s} {'cfg':>3s} {'warm us':>9s} {'cold us':>9s} {'cold/warm':>9s} {'cold TF/s':>9s}")
    for name, M, N, K, conv in SHAPES:
        if only and only not in name:
            continue
        g = torch.Generator().manual_seed(0)
        if conv:
            B, H, Cin = conv
            A = hip.split_planes(torch.randn(B * H * H, Cin, generator=g).cuda())
            W = hip.pack_conv3x3((torch.randn(N, Cin, 3, 3, generator=g) / math.sqrt(K)).cuda(), torch.zeros(N).cuda())
            kw = dict(conv=dict(B=B, Hin=H, Win=H, Cin=Cin, Hout=H, Wout=H, stride=1, upsample=0))
        else:
            A = hip.split_planes(torch.randn(M, K, generator=g).cuda())
            W = hip.pack_linear((torch.randn(N, K, generator=g) / math.sqrt(K)).cuda(), torch.zeros(N).cuda())
            kw = {}
        wbytes = W.data.numel()
        ncopy = max(2, min(64, COLD_BYTES // wbytes + 1))
        Ws = []
        for _ in range(ncopy):
            w2 = copy.copy(W)
            w2.data = W.data.clone()
            Ws.append(w2)
        out = torch.empty(M, N, device="cuda")
        R = torch.randn(M, N, generator=g).cuda()
        reps = max(20, ncopy)
        fl = 2.0 * M * N * K
        for cfg in cfgs:
            warm = bench(lambda i: hip.gemm(A, W, out, prec=4, res=R, workspace=ws, cfg=cfg, **kw), reps)
            cold = bench(lambda i: hip.gemm(A, Ws[i % ncopy], out, prec=4, res=R, workspace=ws, cfg=cfg, **kw), reps)
            print(f"{name:8s} {cfg:3d} {warm:9.1f} {cold:9.1f} {cold / warm:9.2f} {fl / cold / 1e6:9.1f}", flush=True)


if __name__ == "__main__":
    main()
