"""One GEMM shape, a list of (cfg, splitk) variants: HIP-event timing (graph of 20 launches, warm and cold weights) or -- under
rocprofv3 (MVD_PROBE_PLAIN=1) -- 8 plain launches per variant for the PMC summary of tools/gemm_probe_pmc.py.

    python tools/gemm_probe.py conv32 1:1,5:1,9:1,17:1,17:2
"""
import copy
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvdfusion_amd import hip
from gemm_cold import SHAPES, bench


def main():
    name = sys.argv[1]
    variants = [tuple(int(t) for t in v.split(":")) for v in sys.argv[2].split(",")]
    (M, N, K, conv), = [(m, n, k, c) for nm, m, n, k, c in SHAPES if nm == name]
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
    ws = torch.empty(64 * 1024 * 1024, device="cuda")
    out = torch.empty(M, N, device="cuda")
    R = torch.randn(M, N, generator=g).cuda()
    ncopy = max(2, min(64, (640 << 20) // W.data.numel() + 1))
    Ws = []
    for _ in range(ncopy):
        w2 = copy.copy(W)
        w2.data = W.data.clone()
        Ws.append(w2)
    fl = 2.0 * M * N * K
    plain = bool(os.environ.get("MVD_PROBE_PLAIN"))
    for cfg, sk in variants:
        if plain:
            for i in range(8):
                hip.gemm(A, Ws[i % ncopy], out, prec=4, res=R, workspace=ws, cfg=cfg, splitk=sk, **kw)
            torch.cuda.synchronize()
            continue
        reps = max(20, ncopy)
        warm = bench(lambda i: hip.gemm(A, W, out, prec=4, res=R, workspace=ws, cfg=cfg, splitk=sk, **kw), reps)
        cold = bench(lambda i: hip.gemm(A, Ws[i % ncopy], out, prec=4, res=R, workspace=ws, cfg=cfg, splitk=sk, **kw), reps)
        print(f"{name:8s} {hip.kernel_symbol(cfg, 4, bool(conv)):40s} order {'m' if (cfg - 1) & 1 else 'n'} splitk {sk:2d}  "
              f"warm {warm:7.1f} us  cold {cold:7.1f} us  {fl / cold / 1e6:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
