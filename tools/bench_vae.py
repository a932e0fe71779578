"""VAE decode on one MI355X: V latents (4, 32, 32) -> V images (3, 256, 256).

Prints one JSON line in the shape of bench.py's: decoded images / s, HIP-event latency, roofline of the GEMM family
(algorithmic FLOPs of every GEMM launch / their HIP-event time) and the CPU oracle on 16 host threads.
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvdfusion_amd import hip, synthetic as syn
from mvdfusion_amd.autoencoder import AutoencoderKL

DD = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
          num_res_blocks=2, attn_resolutions=[], dropout=0.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--latent", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--encode", action="store_true", help="time AutoencoderKL.encode of V images (3, 8S, 8S) instead")
    a = ap.parse_args()
    V, S = a.views, a.latent
    vae = AutoencoderKL(ddconfig=DD, lossconfig=dict(target="torch.nn.Identity"), embed_dim=4)
    syn.fill_module_(vae, "vae.")
    vae = vae.cuda().eval()
    z = (torch.randn(V, 4, S, S, generator=torch.Generator().manual_seed(0)) * 4.0).cuda()
    if a.encode:
        z = (torch.rand(V, 3, 8 * S, 8 * S, generator=torch.Generator().manual_seed(0)) * 2 - 1).cuda()
        vae.decode = lambda t: vae.encode(t).mode()        # same timing harness below
    for _ in range(a.warmup):
        vae.decode(z)
    torch.cuda.synchronize()
    e0, e1 = hip.Event(), hip.Event()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(a.steps):
        vae.decode(z)
    e1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gpu_ms = e0.elapsed_ms(e1) / a.steps
    # per-GEMM HIP events (eager launches anyway)
    recs, real = [], hip.gemm

    def timed(A, W, out=None, **kw):
        ev0, ev1 = hip.Event(), hip.Event()
        ev0.record()
        r = real(A, W, out, **kw)
        ev1.record()
        conv = kw.get("conv")
        M = conv["B"] * conv["Hout"] * conv["Wout"] if conv else A.shape[0]
        recs.append((2.0 * M * W.n_real * W.K, ev0, ev1))
        return r

    hip.gemm = timed
    try:
        vae.decode(z)
        torch.cuda.synchronize()
    finally:
        hip.gemm = real
    g_ms = sum(r[1].elapsed_ms(r[2]) for r in recs)
    g_fl = sum(r[0] for r in recs)
    out = {"metric": "vae-encoded-images/sec" if a.encode else "vae-decoded-images/sec", "value": V * a.steps / dt, "unit": "images/s", "n_gpus": 1, "steps": a.steps,
           "warmup": a.warmup, "ms_per_step": dt * 1e3 / a.steps, "gpu_ms_per_call_hip_events": gpu_ms,
           "higher_is_better": True, "dtype": "f16x4", "data": "synthetic",
           "config": {"workload": (f"AutoencoderKL.encode of {V} images (3,{8 * S},{8 * S}) -> latent means (4,{S},{S}), SD1 VAE encoder "
                                   "ch=128 mult 1-2-4-4 (ViewFusion.encode, viewfusion_zero_depth_rgb.py:158-159)") if a.encode else
                      (f"AutoencoderKL.decode of {V} latents (4,{S},{S}) -> {V} images (3,{8 * S},{8 * S}), SD1 VAE decoder ch=128 "
                       "mult 1-2-4-4, deterministic-fill weights (ViewFusion.decode, viewfusion_zero_depth_rgb.py:161-163)")},
           "algorithmic_tflop_per_call": g_fl / 1e12,
           "roofline": {"bound": "mfma", "kernel": "gemm_kernel (all instantiations)", "launches_per_step": len(recs),
                        "avg_launch_us": g_ms / len(recs) * 1e3, "achieved": g_fl / (g_ms * 1e-3) / 1e12, "peak": 2500.0,
                        "unit": "TFLOP/s", "frac": g_fl / (g_ms * 1e-3) / 2.5e15, "mfma_products_per_mac": 4,
                        "gemm_share_ms": g_ms, "traffic": None}}
    if not a.no_cpu_baseline:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle import ref_torch as O
        torch.set_num_threads(min(16, os.cpu_count()))
        sd = {"vae." + k: v.detach().cpu() for k, v in vae.state_dict().items()}
        zc = z.cpu()
        fn = (lambda t: O.vae_encode_moments(sd, "vae.", t)) if a.encode else (lambda t: O.viewfusion_decode(sd, t))
        with torch.no_grad():
            fn(zc[:1])
            t0 = time.perf_counter()
            fn(zc)
            ct = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": V / ct, "unit": "images/s", "cores": min(16, os.cpu_count()), "kind": "port",
                               "sample": f"1 timed pass over the same {V} inputs, fp32 PyTorch eager, {ct:.2f} s"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
