"""Run every GEMM configuration twice on the step's shapes and compare bitwise; then eager vs graph steps."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from mvdfusion_amd import hip
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import SHAPES


def gemm_runs():
    ws = torch.empty(64 * 1024 * 1024, device="cuda")
    bad = 0
    for name, M, N, K, conv in SHAPES:
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
        R = torch.randn(M, N, generator=g).cuda()
        ref = None
        for cfg in range(5, 13):   # both tile sizes x both loops x both forced tile orders
            outs = []
            for rep in range(4):
                out = torch.full((M, N), float("nan"), device="cuda")
                hip._TUNED.clear()
                hip.gemm(A, W, out, prec=4, res=R, workspace=ws, cfg=cfg, **kw)
                outs.append(out)
            torch.cuda.synchronize()
            same = all(torch.equal(outs[0], o) for o in outs[1:])
            if not same:
                print("   per-rep max|diff vs rep 3|:", [f"{float((o - outs[3]).abs().max()):.2e}" for o in outs],
                      "mismatching elements rep0:", int((outs[0] != outs[3]).sum()))
            fin = bool(torch.isfinite(outs[0]).all())
            if ref is None:
                ref = outs[0]
            dev = float((outs[0] - ref).abs().max())
            if not same or not fin or dev > 1e-3:
                bad += 1
                print(f"{name:18s} cfg {cfg}: repeatable={same} finite={fin} max|diff vs cfg5|={dev:.3e}")
    print("gemm determinism: bad =", bad)


def steps():
    from conftest import build_model, load_golden
    import test_gpu_model as T
    gd = load_golden("step_mc32_v4_d1")
    m = build_model(32)
    res = {}
    for tag, ug in (("e1", False), ("g1", True), ("g2", True), ("e2", False), ("g3", True)):
        res[tag] = T._run_step(m, gd, 4, 1, 49, 7, use_graph=ug)
    for a in res:
        for b in res:
            if a < b:
                print(a, b, "x maxdiff", float((res[a][0] - res[b][0]).abs().max()), "x0 maxdiff", float((res[a][1] - res[b][1]).abs().max()))


if __name__ == "__main__":
    gemm_runs()
    steps()
