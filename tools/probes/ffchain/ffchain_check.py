"""mvd_ffchain (csrc/ffchain.hip) against the two-launch path it replaces, on the 32x32-level shapes of the step (M = 8192, C = 320):
values (relative error of the fp32 output and of the GroupNorm statistics), then graph-replayed timing of both.
    (apply tools/probes/ffchain/integration.patch, copy ffchain.hip into mvdfusion_amd/csrc/, rebuild)
    python tools/probes/ffchain/ffchain_check.py [M]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import torch.nn as nn
import torch.nn.functional as F

from mvdfusion_amd import hip


def g(seed):
    return torch.Generator().manual_seed(seed)


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    C = 320
    dev = "cuda"
    x = torch.randn(M, C, generator=g(1))
    Win = torch.randn(C, C, generator=g(2)) / math.sqrt(C)
    bin_ = torch.randn(C, generator=g(3)) + 0.5
    res0 = torch.randn(M, C, generator=g(4))
    norm = nn.LayerNorm(C)
    with torch.no_grad():
        norm.weight.copy_(1.0 + 0.3 * torch.randn(C, generator=g(5)))
        norm.bias.copy_(0.2 * torch.randn(C, generator=g(6)))
    ws = torch.empty(32 * 1024 * 1024, device=dev)
    tp = hip.planes_like(M, C, dev)
    tt = torch.empty(M, C, device=dev)
    rs = hip.RowStats(M, C, dev)
    hip.gemm(hip.split_planes(x.to(dev)), hip.pack_linear(Win.to(dev), bin_.to(dev)), tt, res=res0.to(dev), out_planes=tp, row_stats=rs, workspace=ws,
             prec=hip.PREC_X3)
    t2 = tt.cpu()
    Wg = torch.randn(8 * C, C, generator=g(7)) / math.sqrt(C)
    bg = torch.randn(8 * C, generator=g(8))
    Wm = torch.randn(C, 5 * C, generator=g(9)) / math.sqrt(5 * C)
    bm = torch.randn(C, generator=g(10))
    xres = torch.randn(M, C, generator=g(11))
    ln = F.layer_norm(t2.double(), (C,), norm.weight.double(), norm.bias.double(), norm.eps)
    a, gt = F.linear(ln, Wg.double(), bg.double()).chunk(2, dim=-1)
    ref = F.linear(torch.cat([a * F.gelu(gt), t2.double()], 1), Wm.double(), bm.double()) + xres.double()
    fold = hip.LnFold(Wg.to(dev), bg.to(dev), norm.to(dev), geglu=True)
    wm = hip.pack_linear(Wm.to(dev), bm.to(dev))
    cat5 = hip.planes_like(M, 5 * C, dev)
    cat5[:, 2 * 4 * C:] = tp
    xr = xres.to(dev)
    B, HW = M // 1024 if M % 1024 == 0 else 1, 1024 if M % 1024 == 0 else M
    outs, stats = {}, {}

    def run(fused, out, st):
        st.zero_()
        kw1 = dict(M=M, lda=5 * C, epi=hip.EPI_GEGLU, out_planes=cat5, workspace=ws, ln=(rs, fold), prec=hip.PREC_X3)
        kw2 = dict(res=xr, workspace=ws, prec=hip.PREC_X3, gn_stats=st, gn_hw=HW, gn_groups=32)
        if fused:
            d1 = hip.gemm(cat5[:, 2 * 4 * C:], fold.w, None, desc_only=True, **kw1)
            d2 = hip.gemm(cat5, wm, out, desc_only=True, **kw2)
            assert hip.ffchain_supported(d1, d2), "descriptor pair not supported"
            hip.ffchain(d1, d2)
        else:
            hip.gemm(cat5[:, 2 * 4 * C:], fold.w, None, **kw1)
            hip.gemm(cat5, wm, out, **kw2)

    for fused in (False, True):
        out = torch.zeros(M, C, device=dev)
        st = torch.zeros(B, 32, 2, dtype=torch.int64, device=dev)
        if fused:
            cat5[:, :2 * 4 * C].zero_()
        run(fused, out, st)
        torch.cuda.synchronize()
        outs[fused], stats[fused] = out.cpu().double(), st.cpu().double()
        err = float((outs[fused] - ref).abs().max() / ref.abs().max())
        print(f"{'fused  ' if fused else 'two-gemm'}: max rel err vs float64 reference {err:.3e}")
    print(f"fused vs two-gemm: {float((outs[True] - outs[False]).abs().max() / outs[False].abs().max()):.3e}; GroupNorm statistics rel diff "
          f"{float((stats[True] - stats[False]).abs().max() / stats[False].abs().max()):.3e}")
    # timing: 20 calls per graph replay, cold-ish (other weights between calls are absent here: upper bound on what the step sees)
    for fused in (False, True):
        out = torch.zeros(M, C, device=dev)
        st = torch.zeros(B, 32, 2, dtype=torch.int64, device=dev)
        run(fused, out, st)
        graph = hip.Graph()
        with graph:
            for _ in range(20):
                run(fused, out, st)
        graph.launch()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = hip.Event(), hip.Event()
            e0.record()
            graph.launch()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_ms(e1) / 20 * 1e3)
        print(f"{'fused  ' if fused else 'two-gemm'}: {best:.1f} us per chain (M = {M})")


if __name__ == "__main__":
    main()
