"""Round-4 probe: a split-K 3x3 convolution + the GroupNorm / SiLU of its output, as (a) GEMM + split-K reduce with statistics + apply
kernel and (b) GEMM + ONE reduce-and-apply kernel (mvd_gemm_desc.gna_out_sp); graph of 10 repetitions between HIP events, cold caches.
    python tools/probes/gn_fuse_probe.py"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mvdfusion_amd import hip


def main():
    ws = torch.empty(64 * 1024 * 1024, device="cuda")
    flush = torch.empty(512 * 1024 * 1024 // 4, device="cuda")
    g = torch.Generator().manual_seed(0)
    hip.AUTOTUNE = True
    for name, B, H, Cin, Cout in [("32^2 320", 8, 32, 320, 320), ("16^2 640", 8, 16, 640, 640), ("16^2 320->640", 8, 16, 320, 640),
                                  ("8^2 1280", 8, 8, 1280, 1280), ("4^2 1280", 8, 4, 1280, 1280), ("16^2 640 B2", 2, 16, 640, 640),
                                  ("32^2 320 B2", 2, 32, 320, 320)]:
        M, HW = B * H * H, H * H
        A = hip.split_planes(torch.randn(M, Cin, generator=g).cuda())
        W = hip.pack_conv3x3((torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).cuda(), None)
        bb = torch.randn(Cout, generator=g).cuda()
        gm, bt = torch.randn(Cout, generator=g).cuda(), torch.randn(Cout, generator=g).cuda()
        out = torch.empty(M, Cout, device="cuda")
        y = hip.planes_like(M, Cout, "cuda")
        st = torch.zeros(B, 32, 2, dtype=torch.int64, device="cuda")
        conv = dict(B=B, Hin=H, Win=H, Cin=Cin, Hout=H, Wout=H, stride=1, upsample=0)
        res = {}
        for mode in ("two kernels", "fused"):
            def run():
                if mode == "fused":
                    hip.gemm(A, W, out, prec=3, workspace=ws, conv=conv, bias=False, bias_b=bb, rows_per_batch=M, gn_stats=st, gn_hw=HW,
                             gn_apply=(gm, bt, 1e-5, hip.GNA_SILU | hip.GNA_OUT_UNUSED, y))
                else:
                    hip.gemm(A, W, out, prec=3, workspace=ws, conv=conv, bias=False, bias_b=bb, rows_per_batch=M, gn_stats=st, gn_hw=HW)
                    hip.groupnorm_from_stats(out, y, gm, bt, st, B, HW, Cout, 1e-5, 1)
            run()
            cfg = hip.LAST_CFG
            torch.cuda.synchronize()
            graph = hip.Graph()
            with graph:
                for _ in range(10):
                    run()
            best = 1e9
            for _ in range(4):
                flush.zero_()
                e0, e1 = hip.Event(), hip.Event()
                e0.record(); graph.launch(); e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_ms(e1) / 10 * 1e3)
            res[mode] = (best, cfg)
        print(f"{name:16s} M={M:5d} N={Cout:4d} K={9 * Cin:5d}: two kernels {res['two kernels'][0]:6.1f} us (cfg {res['two kernels'][1]})   "
              f"fused {res['fused'][0]:6.1f} us (cfg {res['fused'][1]})", flush=True)


if __name__ == "__main__":
    main()
