"""Round-4 probe: the GEGLU / QKV GEMMs of one step on every kernel configuration that serves their epilogues (since round 4 also the
wave-specialised 128x128 kernel with 64x64 consumer tiles), replayed from a graph between HIP events; LayerNorm fold on (as in the step).
    python tools/probes/epi_ws_probe.py [prec]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn as nn
from mvdfusion_amd import hip


def main():
    prec = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    ws = torch.empty(64 * 1024 * 1024, device="cuda")
    g = torch.Generator().manual_seed(0)
    flush = torch.empty(512 * 1024 * 1024 // 4, device="cuda")
    for name, M, C, heads in [("32^2", 8192, 320, 8), ("16^2", 2048, 640, 8), ("8^2", 512, 1280, 8)]:
        L = {8192: 1024, 2048: 256, 512: 64}[M]
        x = torch.randn(M, C, generator=g).cuda()
        xp = hip.split_planes(x)
        rs = hip.RowStats(M, C, "cuda")
        # row statistics of x the way a producer would have written them: run an identity-free producer (any GEMM with row_stats)
        Win = (torch.randn(C, C, generator=g) / math.sqrt(C)).cuda()
        tp = hip.planes_like(M, C, "cuda")
        tt = torch.empty(M, C, device="cuda")
        hip.gemm(xp, hip.pack_linear(Win, None), tt, out_planes=tp, row_stats=rs, workspace=ws)
        norm = nn.LayerNorm(C).cuda()
        for kind in ("geglu", "qkv"):
            if kind == "geglu":
                W = (torch.randn(8 * C, C, generator=g) / math.sqrt(C)).cuda()
                fold = hip.LnFold(W, torch.zeros(8 * C, device="cuda"), norm, geglu=True)
                outp = hip.planes_like(M, 4 * C, "cuda")
                kw = dict(epi=hip.EPI_GEGLU, out_planes=outp)
                N = 8 * C
            else:
                W = (torch.randn(3 * C, C, generator=g) / math.sqrt(C)).cuda()
                fold = hip.LnFold(W, None, norm)
                planes = hip.alloc_attn_planes(M // L, heads, L, C // heads, "cuda")
                kw = dict(epi=hip.EPI_QKV, qkv=dict(planes=planes, heads=heads, dhead=C // heads, L=L))
                N = 3 * C
            res = []
            only = [int(t) for t in os.environ.get("MVD_PROBE_CFGS", "").split(",") if t]
            for cfg in (only or hip.gemm_configs(kw["epi"])):
                def run():
                    hip.gemm(tp, fold.w, None, prec=prec, workspace=ws, cfg=cfg, splitk=1, ln=(rs, fold), **kw)
                try:
                    run()
                except Exception as e:      # a configuration that does not serve the problem
                    continue
                torch.cuda.synchronize()
                graph = hip.Graph()
                with graph:
                    for _ in range(10):
                        run()
                best = 1e9
                for _ in range(3):
                    flush.zero_()
                    e0, e1 = hip.Event(), hip.Event()
                    e0.record(); graph.launch(); e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_ms(e1) / 10 * 1e3)
                res.append((best, cfg))
            res.sort()
            fl = 2.0 * M * N * C
            print(f"{kind:5s} {name:5s} M={M} N={N} K={C}: " + "  ".join(
                f"{hip.kernel_symbol(c, prec, False).replace('gemm_', '').replace('kernel', 'k')}/{c & 1 ^ 1}:{t:.1f}us" for t, c in res[:6]), flush=True)
            print(f"      best {res[0][0]:.1f} us = {fl / res[0][0] / 1e6:.0f} TFLOP/s algorithmic; ws<128,128>: " +
                  ", ".join(f"{t:.1f}" for t, c in res if hip._cfg_parts(c)[1] == hip.WS_LOOP), flush=True)


if __name__ == "__main__":
    main()
