"""Quick screen of the persistent role-split GEMM (csrc/gemm_pt.hip, cfg loop 10) against the plain 128x128 kernel (bit-equality) and an
fp32 reference, a few shapes per epilogue kind -- run before the pytest suite when the kernel changes (fails fast, seconds).
    python tools/probes/pt_check.py"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from mvdfusion_amd import hip

PT = hip.make_cfg(1, hip.PT_LOOP)
BASE = hip.make_cfg(1, 0)


def g(seed):
    return torch.Generator().manual_seed(seed)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def main():
    ws = torch.empty(32 * 1024 * 1024, device="cuda")
    bad = 0
    for M, N, K in [(2048, 320, 320), (8192, 320, 320), (100, 48, 96), (512, 640, 2592), (4096, 256, 736), (1024, 1024, 512)]:
        a = torch.randn(M, K, generator=g(1))
        w = torch.randn(N, K, generator=g(2)) / math.sqrt(K)
        b = torch.randn(N, generator=g(3))
        r = torch.randn(M, N, generator=g(4))
        ref = F.linear(a, w, b) + r
        Wp = hip.pack_linear(w.cuda(), b.cuda())
        ap, rc = hip.split_planes(a.cuda()), r.cuda()
        for sk in (1, 3):
            o1 = torch.full((M, N), float("nan"), device="cuda")
            o0 = torch.full((M, N), float("nan"), device="cuda")
            hip.gemm(ap, Wp, o1, prec=3, res=rc, workspace=ws, cfg=PT, splitk=sk)
            hip.gemm(ap, Wp, o0, prec=3, res=rc, workspace=ws, cfg=BASE, splitk=sk)
            torch.cuda.synchronize()
            e, eq = rel(o1, ref), torch.equal(o1, o0)
            print(f"dense M={M} N={N} K={K} splitk={sk}: rel {e:.2e} bit-equal {eq}", flush=True)
            # (K = 96 with three splits: the persistent kernel keeps >= 2 k-tiles per split and runs unsplit -- another summation order)
            bad += (e > 3e-6) + (not eq and not (K == 96 and sk == 3))
    for B, H, Cin, Cout in [(2, 32, 64, 64), (2, 16, 96, 320), (1, 8, 320, 48), (11, 4, 64, 80), (8, 32, 320, 320)]:
        x = torch.randn(B, Cin, H, H, generator=g(40))
        w = torch.randn(Cout, Cin, 3, 3, generator=g(41)) / math.sqrt(9 * Cin)
        ref = F.conv2d(x, w, None, padding=1).permute(0, 2, 3, 1).reshape(B * H * H, Cout)
        Wp = hip.pack_conv3x3(w.cuda(), None)
        xp = hip.split_planes(x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().cuda())
        kw = dict(conv=dict(B=B, Hin=H, Win=H, Cin=Cin, Hout=H, Wout=H, stride=1, upsample=0))
        o1 = torch.full((B * H * H, Cout), float("nan"), device="cuda")
        o0 = torch.full((B * H * H, Cout), float("nan"), device="cuda")
        hip.gemm(xp, Wp, o1, prec=3, workspace=ws, cfg=PT, splitk=1, **kw)
        hip.gemm(xp, Wp, o0, prec=3, workspace=ws, cfg=BASE, splitk=1, **kw)
        torch.cuda.synchronize()
        e, eq = rel(o1, ref), torch.equal(o1, o0)
        print(f"conv B={B} H={H} Cin={Cin} Cout={Cout}: rel {e:.2e} bit-equal {eq}", flush=True)
        bad += (e > 3e-6) + (not eq)
    for M, C in [(256, 64), (2048, 320)]:
        a = torch.randn(M, C, generator=g(11))
        w = torch.randn(8 * C, C, generator=g(12)) / math.sqrt(C)
        b = torch.randn(8 * C, generator=g(13))
        h = F.linear(a, w, b)
        v, gt = h.chunk(2, dim=-1)
        ref = v * F.gelu(gt)
        Wp = hip.pack_linear(w.cuda(), b.cuda(), geglu=True)
        ap = hip.split_planes(a.cuda())
        o1 = torch.full((M, 4 * C), float("nan"), device="cuda")
        o0 = torch.full((M, 4 * C), float("nan"), device="cuda")
        p1, p0 = hip.planes_like(M, 4 * C, "cuda"), hip.planes_like(M, 4 * C, "cuda")
        hip.gemm(ap, Wp, o1, prec=3, epi=hip.EPI_GEGLU, workspace=ws, splitk=1, cfg=PT, out_planes=p1)
        hip.gemm(ap, Wp, o0, prec=3, epi=hip.EPI_GEGLU, workspace=ws, splitk=1, cfg=BASE, out_planes=p0)
        torch.cuda.synchronize()
        e, eq = rel(o1, ref), torch.equal(o1, o0) and torch.equal(p1, p0)
        print(f"geglu M={M} C={C}: rel {e:.2e} bit-equal {eq}", flush=True)
        bad += (e > 3e-6) + (not eq)
    for Bv, L, H, d in [(2, 256, 8, 40), (2, 64, 8, 160), (8, 1024, 8, 40)]:
        C = H * d
        M = Bv * L
        x = torch.randn(M, C, generator=g(21))
        w = torch.randn(3 * C, C, generator=g(22)) / math.sqrt(C)
        Wp = hip.pack_linear(w.cuda(), None)
        xp = hip.split_planes(x.cuda())
        pl1 = hip.alloc_attn_planes(Bv, H, L, d, "cuda")
        pl0 = hip.alloc_attn_planes(Bv, H, L, d, "cuda")
        hip.gemm(xp, Wp, None, prec=3, epi=hip.EPI_QKV, qkv=dict(planes=pl1, heads=H, dhead=d, L=L), workspace=ws, cfg=PT, splitk=1)
        hip.gemm(xp, Wp, None, prec=3, epi=hip.EPI_QKV, qkv=dict(planes=pl0, heads=H, dhead=d, L=L), workspace=ws, cfg=BASE, splitk=1)
        torch.cuda.synchronize()
        eq = all(torch.equal(u, v) for u, v in zip(pl1, pl0))
        print(f"qkv B={Bv} L={L} heads={H} d={d}: planes bit-equal {eq}", flush=True)
        bad += (not eq)
    print("PT CHECK", "FAILED" if bad else "OK", bad, flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
