"""Micro-benchmark of mvd_gemm on the shapes of one DDIM step (HIP events on the launch stream)."""
import math
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvdfusion_amd import hip

SHAPES = [  # (name, M, N, K, conv(B,H,Cin) or None)
    ("proj 32^2", 8192, 320, 320, None), ("qkv 32^2", 8192, 960, 320, None), ("ff1 32^2", 8192, 2560, 320, None),
    ("ff2 32^2", 8192, 320, 1280, None), ("conv 32^2 320", 8192, 320, 2880, (8, 32, 320)),
    ("conv 32^2 640in", 8192, 320, 5760, (8, 32, 640)), ("conv 32^2 960in", 8192, 320, 8640, (8, 32, 960)),
    ("proj 16^2", 2048, 640, 640, None), ("ff1 16^2", 2048, 5120, 640, None), ("ff2 16^2", 2048, 640, 2560, None),
    ("conv 16^2 640", 2048, 640, 5760, (8, 16, 640)), ("conv 16^2 1920in", 2048, 640, 17280, (8, 16, 1920)),
    ("proj 8^2", 512, 1280, 1280, None), ("ff1 8^2", 512, 10240, 1280, None), ("ff2 8^2", 512, 1280, 5120, None),
    ("conv 8^2 1280", 512, 1280, 11520, (8, 8, 1280)), ("conv 8^2 2560in", 512, 1280, 23040, (8, 8, 2560)),
    ("conv 4^2 1280", 128, 1280, 11520, (8, 4, 1280)), ("conv 4^2 2560in", 128, 1280, 23040, (8, 4, 2560)),
    ("ga pre", 16384, 256, 736, None), ("ga qkv", 16384, 768, 256, None), ("ga fc1", 16384, 512, 256, None),
    ("ga fc2", 16384, 256, 512, None),
]


def main():
    prec = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    only = sys.argv[2] if len(sys.argv) > 2 else None
    ws = torch.empty(64 * 1024 * 1024, device="cuda")
    hip.AUTOTUNE = bool(os.environ.get("MVD_BENCH_TUNE"))
    force = int(os.environ["MVD_BENCH_CFG"]) if os.environ.get("MVD_BENCH_CFG") else None   # explicit kernel configuration
    tot_ms, tot_fl = 0.0, 0.0
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
        out = torch.empty(M, N, device="cuda")
        R = torch.randn(M, N, generator=g).cuda()
        for _ in range(3):
            hip.gemm(A, W, out, prec=prec, res=R, workspace=ws, cfg=force, **kw)
        torch.cuda.synchronize()
        reps = 20
        class _Direct:                # MVD_BENCH_NOGRAPH=1: plain launches (for rocprofv3 counter passes)
            def launch(self):
                for _ in range(reps):
                    hip.gemm(A, W, out, prec=prec, res=R, workspace=ws, cfg=force, **kw)
        if os.environ.get("MVD_BENCH_NOGRAPH"):
            graph = _Direct()
        else:
            graph = hip.Graph()          # replayed graph: no host launch overhead between the kernels
            with graph:
                for _ in range(reps):
                    hip.gemm(A, W, out, prec=prec, res=R, workspace=ws, cfg=force, **kw)
        graph.launch()
        torch.cuda.synchronize()
        e0, e1 = hip.Event(), hip.Event()
        e0.record()
        graph.launch()
        e1.record()
        ms = e0.elapsed_ms(e1) / reps
        fl = 2.0 * M * N * K
        tot_ms += ms
        tot_fl += fl
        name = f"{name} c{hip.LAST_CFG}"
        print(f"{name:20s} M={M:6d} N={N:6d} K={K:6d}  {ms*1e3:9.1f} us  {fl/ms/1e9:8.1f} TF/s (x{prec} MFMA: {prec*fl/ms/1e9:7.1f})")
    print(f"TOTAL {tot_ms:.3f} ms  {tot_fl/tot_ms/1e9:.1f} TF/s algorithmic")


if __name__ == "__main__":
    main()
