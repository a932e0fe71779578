"""Does the ROW STRIDE of the A operand matter?  A split-planes row is 4 K bytes; with K = 320 / 640 / 1280 / 2560 the rows of a tile -- and the
same rows of every other tile -- start 10 / 20 / 40 / 80 cache lines apart, i.e. on 8 / 4 / 2 / 1 of 16 L2 channels if the channel is a
plain function of the low line-address bits, and all CUs walk k in lockstep.  Same GEMM, A with ld = K (as the model allocates it) and with
ld = K + 32 / K + 96 (rows 1 / 3 lines further apart); best tuner candidate per variant, cold weights, graph of >= 20 launches.

    python tools/probes/stride_probe.py > gpurun_out/stride_probe.log
"""
import copy
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from mvdfusion_amd import hip
from gemm_cold import bench

SHAPES = [("proj32", 8192, 320, 320), ("ff1_32", 8192, 2560, 320), ("ff2_32", 8192, 320, 1280), ("proj16", 2048, 640, 640),
          ("ff2_16", 2048, 640, 2560), ("proj8", 512, 1280, 1280), ("ff2_8", 512, 1280, 5120)]


def main():
    g = torch.Generator().manual_seed(0)
    ws = torch.empty(64 * 1024 * 1024, device="cuda")
    for name, M, N, K in SHAPES:
        W = hip.pack_linear((torch.randn(N, K, generator=g) / math.sqrt(K)).cuda(), torch.zeros(N).cuda())
        ncopy = max(2, min(64, (640 << 20) // W.data.numel() + 1))
        Ws = []
        for _ in range(ncopy):
            w2 = copy.copy(W)
            w2.data = W.data.clone()
            Ws.append(w2)
        x = torch.randn(M, K, generator=g).cuda()
        out = torch.empty(M, N, device="cuda")
        R = torch.randn(M, N, generator=g).cuda()
        fl = 2.0 * M * N * K
        reps = max(20, ncopy)
        for pad in (0, 32, 96):
            A = hip.split_planes(x, ldp=K + pad)                      # (M, 2 * (K + pad)) planes, zero-padded columns
            d = hip.GemmDesc()
            best = (1e9, 0, 0)
            for cfg in hip.gemm_configs(hip.EPI_STORE):
                for sk in (1, 0):
                    try:
                        t = bench(lambda i: hip.gemm(A, Ws[i % ncopy], out, prec=3, res=R, workspace=ws, cfg=cfg, splitk=sk, lda=K + pad, M=M), reps)
                    except Exception:
                        continue
                    if t < best[0]:
                        best = (t, cfg, sk)
            t, cfg, sk = best
            print(f"{name:8s} M={M:5d} N={N:5d} K={K:5d}  ld = K + {pad:2d} (row stride {4 * (K + pad):6d} B = {(4 * (K + pad)) // 128:3d} lines): best {t:7.1f} us  "
                  f"{fl / t / 1e6:6.1f} TF/s  {hip.kernel_symbol(cfg, 3, False)} splitk {sk}", flush=True)


if __name__ == "__main__":
    main()
