"""Where the three roles of the persistent GEMM (csrc/gemm_pt.hip) spend their cycles: workgroup 0's consumer 0 / loader 0 / epilogue wave 0
on the GEGLU / QKV projections of the step (LayerNorm fold on) and a plain store GEMM.  Profiling build: tools/probes/pt_stamp.sh.
    python tools/probes/pt_stamp.py"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.nn as nn
from mvdfusion_amd import hip

hip.LIB_PATHS["f16"] = os.path.join(ROOT, "tools", "probes", "libmvd_hip_ptstamp" + os.environ.get("PT_VARIANT", "") + ".so")
ONLY = os.environ.get("PT_ONLY", "").split(",") if os.environ.get("PT_ONLY") else None
PT = hip.make_cfg(1, hip.PT_LOOP)


def show(tag, ws, us):
    d = ws.view(torch.int64)[:48].cpu().tolist()
    c, l, e = d[0:16], d[16:32], d[32:48]
    hw = ws.view(torch.int64)[48:64].cpu().tolist()
    print(f"{tag}: {us:.1f} us/launch   wave -> SIMD (HW_ID[5:4]): " + " ".join(str((h >> 4) & 3) for h in hw) +
          "   wave slot: " + " ".join(str(h & 15) for h in hw), flush=True)
    print(f"   consumer0: total {c[0]} cyc, first k-tile after {c[1]}, waiting for k-tiles {c[2]} ({c[3]} waits), dump wait {c[4]}, dump {c[5]}, "
          f"k-tiles {c[6]} -> {(c[0] - c[1]) / max(c[6], 1):.0f} cyc / k-tile")
    print(f"   loader0  : total {l[0]} cyc, iterations {l[1]} (idle {l[2]}), issue cycles {l[3]} ({l[3] / max(l[6], 1):.0f} / k-tile), "
          f"slot-not-free polls {l[4]}, in-flight-limit polls {l[5]}, k-tiles {l[6]}")
    print(f"   epilogue0: total {e[0]} cyc, LN rows {e[1]}, waiting for tiles {e[2]}, epilogue {e[3]} ({e[3] / max(e[4], 1):.0f} / unit), units {e[4]}", flush=True)


def timed(run):
    run()
    torch.cuda.synchronize()
    graph = hip.Graph()
    with graph:
        for _ in range(10):
            run()
    best = 1e9
    for _ in range(3):
        e0, e1 = hip.Event(), hip.Event()
        e0.record(); graph.launch(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_ms(e1) / 10 * 1e3)
    return best


def main():
    prec = 3
    ws = torch.zeros(64 * 1024 * 1024, device="cuda")
    g = torch.Generator().manual_seed(0)
    for name, M, C, heads in [("32^2", 8192, 320, 8), ("16^2", 2048, 640, 8), ("8^2", 512, 1280, 8)]:
        L = {8192: 1024, 2048: 256, 512: 64}[M]
        x = torch.randn(M, C, generator=g).cuda()
        xp = hip.split_planes(x)
        rs = hip.RowStats(M, C, "cuda")
        Win = (torch.randn(C, C, generator=g) / math.sqrt(C)).cuda()
        tp = hip.planes_like(M, C, "cuda")
        tt = torch.empty(M, C, device="cuda")
        Wl = hip.pack_linear(Win, None)
        hip.gemm(xp, Wl, tt, out_planes=tp, row_stats=rs, workspace=ws)
        norm = nn.LayerNorm(C).cuda()
        for kind in ("geglu", "geglu-noln", "qkv", "store"):
            if ONLY and kind not in ONLY:
                continue
            if kind.startswith("geglu"):
                W = (torch.randn(8 * C, C, generator=g) / math.sqrt(C)).cuda()
                fold = hip.LnFold(W, torch.zeros(8 * C, device="cuda"), norm, geglu=True)
                outp = hip.planes_like(M, 4 * C, "cuda")
                kw = dict(epi=hip.EPI_GEGLU, out_planes=outp)
                if kind == "geglu":
                    kw["ln"] = (rs, fold)
                Wuse = fold.w
            elif kind == "qkv":
                W = (torch.randn(3 * C, C, generator=g) / math.sqrt(C)).cuda()
                fold = hip.LnFold(W, None, norm)
                planes = hip.alloc_attn_planes(M // L, heads, L, C // heads, "cuda")
                kw = dict(epi=hip.EPI_QKV, qkv=dict(planes=planes, heads=heads, dhead=C // heads, L=L), ln=(rs, fold))
                Wuse = fold.w
            else:
                Wuse = Wl
                kw = dict(out_planes=hip.planes_like(M, C, "cuda"), res=tt)
            out = torch.empty(M, C, device="cuda") if kind == "store" else None

            def run():
                hip.gemm(tp, Wuse, out, prec=prec, workspace=ws, cfg=PT, splitk=1, **kw)
            us = timed(run)
            show(f"{kind:10s} {name} M={M} N={Wuse.N} K={C}", ws, us)


if __name__ == "__main__":
    main()
