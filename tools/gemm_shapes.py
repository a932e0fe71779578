"""Per-shape GEMM time of one DDIM step (tuned configs, eager launches with HIP events): where the GEMM milliseconds go."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mvdfusion_amd import hip


def main():
    V, S, D, cfg_scale = int(os.environ.get("V", 4)), int(os.environ.get("S", 32)), 1, 2.5
    m, sd = bench.build(V, S, D, os.environ.get("PREC", "f16x3"))
    eng, inp, dn, sn = bench.prepare(m, V, S, D, cfg_scale)
    bench.run_steps(eng, 2, cfg_scale, None, True)      # tunes + captures
    torch.cuda.synchronize()
    recs = []
    real = hip.gemm

    def timed(A, W, out=None, **kw):
        e0, e1 = hip.Event(), hip.Event()
        e0.record()
        r = real(A, W, out, **kw)
        e1.record()
        conv = kw.get("conv")
        M = conv["B"] * conv["Hout"] * conv["Wout"] if conv else int(kw.get("M") or A.numel() // A.shape[-1])
        kind = "conv" if conv else {hip.EPI_STORE: "lin", hip.EPI_GEGLU: "geglu", hip.EPI_QKV: "qkv"}[kw.get("epi", hip.EPI_STORE)]
        if conv and (conv["stride"] != 1 or conv["upsample"]):
            kind = "conv-s2" if conv["stride"] != 1 else "conv-up"
        recs.append(((kind, M, W.N, W.K, hip.LAST_CFG), e0, e1))
        return r

    hip.gemm = timed
    try:
        for _ in range(3):
            eng.step(cfg_scale, do_update=True, use_graph=False)
        torch.cuda.synchronize()
    finally:
        hip.gemm = real
    n = len(recs) // 3
    by = {}
    for key, e0, e1 in recs[2 * n:]:
        b = by.setdefault(key, [0, 0.0])
        b[0] += 1
        b[1] += e0.elapsed_ms(e1)
    tot = sum(b[1] for b in by.values())
    print(f"{n} GEMM launches / step, {tot:.3f} ms (eager, event-timed)")
    for (kind, M, N, K, cfg), (cnt, ms) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        fl = 2.0 * M * N * K * cnt
        print(f"{kind:8s} M={M:6d} N={N:6d} K={K:6d} cfg={cfg:2d} x{cnt:3d}  {ms:7.3f} ms  {ms / cnt * 1e3:7.1f} us/launch  "
              f"{fl / ms / 1e9:6.1f} TF/s  {100 * ms / tot:4.1f}%")


if __name__ == "__main__":
    main()
