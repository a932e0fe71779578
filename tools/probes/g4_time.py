"""Time of the fused GridAttn kernel alone (csrc/gridattn_fused.hip, g4_fused_kernel<3>) at V = 4 / 8 / 15, 32x32 latents: graph of 5
launches between HIP events; FLOPs = T (3 516 416 + 3 072 V); pipe fraction = 3 products per MAC / 2.5 PFLOP/s.
    python tools/probes/g4_time.py [V ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from mvdfusion_amd import hip, synthetic as syn
from mvdfusion_amd.cameras import pack_cameras
from mvdfusion_amd.engine import Ctx
from mvdfusion_amd.scheduler import make_tables
from conftest import build_model

STAMP = "--stamp" in sys.argv
if STAMP:
    hip.LIB_PATHS["f16"] = os.path.join(ROOT, "tools", "probes", "libmvd_hip_g4stamp.so")


def main():
    Vs = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [4, 8, 15]
    S, D = 32, 1
    m = build_model(32, D=D)
    ga = m.view_attn
    tab = make_tables()
    tval = 381
    sac = tab["sqrt_alphas_cumprod"][tval]
    dstd = tab["sqrt_one_minus_alphas_cumprod"][tval] / sac / 10.0
    steps = torch.tensor([[float(tval), float(sac), float(dstd), 1, 1, 0, 0, 0]], dtype=torch.float32).cuda()
    it = torch.zeros(1, dtype=torch.int32, device="cuda")
    for V in Vs:
        ctx = Ctx("cuda", hip.PREC_X3)
        inp = syn.make_inputs(V, S, 4)
        g = torch.Generator().manual_seed(17)
        x = torch.randn(V, 5, S, S, generator=g).cuda()
        dn = torch.randn(1, V, D, S, S, generator=g).cuda()
        c = (torch.randn(1, 256, generator=g) * 0.5).cuda()
        cams, icam, il = pack_cameras(inp["batch_cameras"]).cuda(), pack_cameras(inp["input_cameras"]).cuda(), inp["input_latents"].cuda()
        vol = torch.zeros(V * S * S * D, 768, device="cuda")
        lib = hip.lib()
        real = lib.mvd_gridattn_fused
        evs = []

        def timed(*a):
            e0, e1 = hip.Event(), hip.Event()
            e0.record()
            for _ in range(5):
                r = real(*a)
            e1.record()
            evs.append((e0, e1))
            return r

        ga.run(ctx, x, dn, steps, it, cams, icam, il, c, vol, V, S, D, fused=True)      # warm (packs the weight stream)
        torch.cuda.synchronize()
        lib.mvd_gridattn_fused = timed
        try:
            for _ in range(3):
                ga.run(ctx, x, dn, steps, it, cams, icam, il, c, vol, V, S, D, fused=True)
            torch.cuda.synchronize()
        finally:
            lib.mvd_gridattn_fused = real
        us = min(e0.elapsed_ms(e1) for e0, e1 in evs) / 5 * 1e3
        T = V * S * S * D * V
        fl = T * (3516416.0 + 3072.0 * V)
        if STAMP:
            import ctypes as C
            dbg = torch.zeros(8, dtype=torch.int64, device="cuda")
            lib.mvd_gridattn_fused_debug.argtypes = [C.c_void_p]
            lib.mvd_gridattn_fused_debug.restype = None
            lib.mvd_gridattn_fused_debug(dbg.data_ptr())
            ga.run(ctx, x, dn, steps, it, cams, icam, il, c, vol, V, S, D, fused=True)
            torch.cuda.synchronize()
            t, w, sl = dbg[:3].tolist()
            print(f"      stamps (workgroup 0, wave 0): total {t} cycles = waiting for weight slots {w} ({100 * w / t:.0f} %) + inside the slots "
                  f"(fragment reads + MFMAs) {sl} ({100 * sl / t:.0f} %; 215 slots -> {sl / 215:.0f} per slot, MFMA time 768) + other (VALU "
                  f"phases, token generation) {t - w - sl} ({100 * (t - w - sl) / t:.0f} %)", flush=True)
            lib.mvd_gridattn_fused_debug(None)
        print(f"V={V:2d}: g4_fused_kernel<3> {us:8.1f} us   {fl / us / 1e6:6.1f} TFLOP/s algorithmic = {fl / us / 1e6 / 2500:.3f} of the dense peak, "
              f"{3 * fl / us / 1e6 / 2500:.3f} of the MFMA pipe (3 products)", flush=True)


if __name__ == "__main__":
    main()
