#!/usr/bin/env python
"""bench.py -- denoising-steps/sec of the MI355X-native MVD-Fusion hot path (BASELINE.json metric).

A "step" is one DDIM iteration for all V views: GridAttn (depth unproject / reproject / cross-view aggregation) +
the classifier-free-guidance pair of view-conditioned UNet passes (batched as 2V) + CFG combine + DDIM update, i.e.
one replay of the captured hipGraph (DDIMSampler.sample's loop body, mvdfusion/sampler.py:119-142).

  N=1  : BASELINE.json configs[1] -- V=4 views x 256^2 images (32x32 latents), D=1, cfg 2.5, full-width SD1.x UNet.
         (--views 8 = configs[2]'s workload on one GPU; --views 8 --latent 64 = configs[3].)
  N>1  : BASELINE.json configs[2] -- V=8 views sharded by view over the N ranks (one process per GPU), one RCCL
         all-gather of the updated latent rows per step (mvdfusion_amd/parallel.py).  Total work is fixed => "strong".
  --shard-emulate r/N : ONE GPU runs the shard rank r of an N-way view-parallel job would run (its Vq query views against
         all V references) and reports the projected view-parallel speed-up bound t(unsharded) / t(shard).

Weights: deterministic non-zero synthetic fill of the real architecture (no checkpoints offline); data: synthetic GSO rig.
Prints ONE JSON line on rank 0.  Extra objects: `roofline` (dominant kernel family, HIP-event timed, algorithmic FLOPs
against the dense 16-bit MFMA peak), `roofline_groups` (attention, GroupNorm, LayerNorm, GridAttn aggregation) and
`cpu_baseline` (the CPU oracle timed on this host's cores on a bounded sample).
Environment: MVD_PREFETCH=ws|0 (weight prefetch of the captured step; `config.weight_prefetch` records it), MVD_HIP_LIB=<.so>
(another build of the same ABI, tools/probes/ab_build.sh: same-box A/B runs), MVD_BENCH_DUMP_GEMMS=<file> (one line per mvd_gemm launch of
the eager profile pass: kernel, shape, microseconds -- the tool that found round 5's occupancy regression).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_16BIT_DENSE_PEAK = 2.5e15  # FLOP/s, /opt/skills/guides/MI355X_MICROARCH.md (dense; not the 2:1-sparse figure)
HBM_PEAK = 8.0e12               # B/s (spec; ~6.3 TB/s achievable), same guide
ROUND = "r06"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def tree_fingerprint():
    """sha256 over the sources that decide which kernels a step launches (csrc/*.hip, *.hpp, the header, the host mirror): files under
    profiles/ written by tools/pmc_traffic.py carry it, and their numbers enter the bench line only when it matches the running tree."""
    import glob
    import hashlib
    h = hashlib.sha256()
    pats = ("mvdfusion_amd/csrc/*.hip", "mvdfusion_amd/csrc/*.hpp", "include/*.h", "mvdfusion_amd/*.py")
    for f in sorted(f for pat in pats for f in glob.glob(os.path.join(ROOT, pat))):
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def weight_prefetch_label():
    """How the captured step prefetches weights (mvdfusion_amd/viewfusion_zero_depth_rgb.py: MVD_PREFETCH)."""
    from mvdfusion_amd import viewfusion_zero_depth_rgb as vf
    mode = vf.PREFETCH_WEIGHTS
    if mode != "ws":
        return "off"
    opts = ",".join(f"{k}={v}" for k, v in sorted(vf.PREFETCH_OPTIONS.items()))
    return "in-kernel (role-split consumer wavefronts request the following launches' weights)" + (f" [{opts}]" if opts else "")


def build(V, S, D, precision, sd=None):
    from mvdfusion_amd import synthetic as syn
    from mvdfusion_amd.configs import model_config
    from mvdfusion_amd.viewfusion_zero_depth_rgb import ViewFusion
    t0 = time.time()
    with syn.skip_default_init():      # (the fill / load_state_dict below overwrites every parameter)
        m = ViewFusion(**model_config(320, D=D, S=S, precision=precision))
    if sd is None:      # deterministic non-zero fill keyed by the parameter names (the reference zero-inits every residual branch: SURVEY T1)
        syn.fill_module_(m)
        sd = {k: v.detach().clone() for k, v in m.state_dict().items() if not k.startswith("scheduler.")}
    else:
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not unexpected and all(k.startswith("scheduler.") for k in missing), (missing[:5], unexpected[:5])
    m = m.cuda().eval()
    log(f"[bench] model built in {time.time() - t0:.1f}s")
    return m, sd


def prepare(m, V, S, D, cfg_scale, q0=0, Vq=None, seed=0):
    from mvdfusion_amd import synthetic as syn
    from mvdfusion_amd.engine import ddim_step_table
    inp = syn.make_inputs(V, S, seed=seed)
    dn, sn = syn.step_noise(V, S, D, 50, seed=seed)
    eng = m.engine(V, S, D, cfg_scale != 1.0, q0=q0, Vq=Vq)
    eng.set_conditioning(inp["batch_cameras"], inp["input_latents"].cuda(), inp["input_cameras"], inp["clip_v_embed"].cuda())
    st, dd = m.ddim.tables()
    eng.set_schedule(ddim_step_table(st, dd, [49 - i for i in range(50)]), dn, sn)
    eng.x.copy_(inp["x_T"])
    return eng, inp, dn, sn


def run_steps(eng, n, cfg_scale, exchange=None, use_graph=True):
    """n DDIM iterations (wrapping to a fresh sample every 50)."""
    for _ in range(n):
        if eng.done == eng.n_rows:
            eng.rewind()
        eng.step(cfg_scale, do_update=True, use_graph=use_graph)
        if exchange is not None:
            exchange.gather(eng.x)


def profile_kernel_groups(m, eng, cfg_scale):
    """One eager (non-graph) step with HIP events on the launch stream around every mvd_gemm launch and around the other
    kernel groups (attention, GroupNorm, LayerNorm, the whole GridAttn block)."""
    from mvdfusion_amd import hip
    from mvdfusion_amd.view_attn_efficient2 import GridAttn
    recs, groups = [], {"attention": [], "groupnorm": [], "layernorm": [], "gridattn": [], "gridattn_fused_kernel": []}
    real = dict(gemm=hip.gemm, attention=hip.attention, groupnorm=hip.groupnorm, layernorm=hip.layernorm, ga=GridAttn.run)

    def ev_pair():
        return hip.Event(), hip.Event()

    def timed_gemm(A, W, out=None, **kw):
        e0, e1 = ev_pair()
        e0.record()
        r = real["gemm"](A, W, out, **kw)
        e1.record()
        conv = kw.get("conv")
        M = conv["B"] * conv["Hout"] * conv["Wout"] if conv else int(kw.get("M") or A.numel() // A.shape[-1])
        N, Kp = W.n_real, W.K
        rows_in = conv["B"] * conv["Hin"] * conv["Win"] if conv else M
        k_in = conv["Cin"] if conv else Kp
        n_out = 4.0 * M * N * ((out is not None) + (kw.get("out_planes") is not None) + (kw.get("res") is not None)) \
            + (4.0 * M * N if kw.get("qkv") else 0.0)
        abytes = 4.0 * rows_in * k_in + 4.0 * W.N * Kp + n_out      # A planes + packed W (hi+lo = 4 B/elem) + outputs / residual
        recs.append(dict(sym=hip.kernel_symbol(hip.LAST_CFG, kw.get("prec", 4), bool(conv)), M=M, N=N, K=Kp,
                         flops=2.0 * M * N * Kp, bytes=abytes, prec=kw.get("prec", 4), ev=(e0, e1)))
        return r

    def timed_attention(planes, out, B, heads, L, dhead, prec=hip.PREC_X4, Lkeys=0):
        e0, e1 = ev_pair()
        e0.record()
        r = real["attention"](planes, out, B, heads, L, dhead, prec=prec, Lkeys=Lkeys)
        e1.record()
        groups["attention"].append(dict(flops=4.0 * B * heads * L * L * dhead, bytes=0.0, ev=(e0, e1)))
        return r

    def timed_groupnorm(x, y, gamma, beta, B, HW, Cc, eps, silu, ws):
        e0, e1 = ev_pair()
        e0.record()
        r = real["groupnorm"](x, y, gamma, beta, B, HW, Cc, eps, silu, ws)
        e1.record()
        groups["groupnorm"].append(dict(flops=0.0, bytes=12.0 * B * HW * Cc, ev=(e0, e1)))   # read x twice, write planes
        return r

    def timed_layernorm(x, y, w, b, rows, Cc, eps=1e-5, w_plus_one=False, y_f32=None):
        e0, e1 = ev_pair()
        e0.record()
        r = real["layernorm"](x, y, w, b, rows, Cc, eps, w_plus_one, y_f32)
        e1.record()
        groups["layernorm"].append(dict(flops=0.0, bytes=8.0 * rows * Cc, ev=(e0, e1)))
        return r

    def timed_ga(self, ctx, x, *a, **kw):
        e0, e1 = ev_pair()
        e0.record()
        r = real["ga"](self, ctx, x, *a, **kw)
        e1.record()
        V, S, D = a[8], a[9], a[10]
        Vq = kw.get("Vq") or V
        T = Vq * S * S * D * V
        # the aggregation tail (723->256, 3 DiT blocks over V, pool, 256->768) per token
        groups["gridattn"].append(dict(flops=T * (3516416.0 + 3072.0 * V), bytes=0.0, ev=(e0, e1)))
        return r

    lib = hip.lib()
    real_fused = lib.mvd_gridattn_fused
    # every mvd_gemm descriptor of the step, by value: replayed back to back from ONE graph below (graph-replay time of the family, in-run)
    import ctypes as C
    real_mvd_gemm, descs = lib.mvd_gemm, []

    def recording_mvd_gemm(dref, stream):
        descs.append(hip.GemmDesc.from_buffer_copy(bytes(dref._obj)))
        return real_mvd_gemm(dref, stream)

    def timed_fused(*a):
        e0, e1 = ev_pair()
        e0.record()
        r = real_fused(*a)
        e1.record()
        V, Vq, S, D = a[12], a[14], a[15], a[16]
        groups["gridattn_fused_kernel"].append(dict(flops=Vq * S * S * D * V * (3516416.0 + 3072.0 * V), bytes=0.0, ev=(e0, e1)))
        return r

    lib.mvd_gridattn_fused = timed_fused
    lib.mvd_gemm = recording_mvd_gemm
    hip.gemm, hip.attention, hip.groupnorm, hip.layernorm, GridAttn.run = timed_gemm, timed_attention, timed_groupnorm, \
        timed_layernorm, timed_ga
    try:
        it, x = eng.done, eng.x.clone()
        if it == eng.n_rows:
            it = 0
        eng.rewind(it)
        eng.step(cfg_scale, do_update=True, use_graph=False)
        torch.cuda.synchronize()
        eng.rewind(it)
        eng.x.copy_(x)
    finally:
        hip.gemm, hip.attention, hip.groupnorm, hip.layernorm, GridAttn.run = real["gemm"], real["attention"], \
            real["groupnorm"], real["layernorm"], real["ga"]
        lib.mvd_gridattn_fused = real_fused
        lib.mvd_gemm = real_mvd_gemm
    # the GEMM family under graph replay, measured HERE: one graph holding every mvd_gemm launch of the step (their split-K reduce /
    # GroupNorm-apply kernels ride along), the same buffers, one event pair per replay -- no eager launch gaps inside the timed region
    family_replay_ms = None
    try:
        graph = hip.Graph()
        with graph:
            for dd in descs:
                hip.check(real_mvd_gemm(C.byref(dd), hip.stream()))
        graph.launch()
        torch.cuda.synchronize()
        best = float("inf")
        for _ in range(3):
            e0, e1 = ev_pair()
            e0.record()
            graph.launch()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_ms(e1))
        family_replay_ms = best
        eng.rewind(it)           # (the replays re-added producer statistics: the next step zeroes its arena anyway; restore the state)
        eng.x.copy_(x)
    except Exception as e:       # the figure is optional: never fail the bench line for it
        log(f"[bench] family graph replay skipped: {e}")
    by = {}
    if os.environ.get("MVD_BENCH_DUMP_GEMMS"):          # diagnosis: one line per mvd_gemm launch of the eager profile pass
        with open(os.environ["MVD_BENCH_DUMP_GEMMS"], "w") as f:
            for i, r in enumerate(recs):
                f.write(f"{i} {r['sym']} M={r['M']} N={r['N']} K={r['K']} {r['ev'][0].elapsed_ms(r['ev'][1]) * 1e3:.1f}\n")
    for r in recs:
        ms = r["ev"][0].elapsed_ms(r["ev"][1])
        b = by.setdefault(r["sym"], dict(n=0, ms=0.0, flops=0.0, bytes=0.0, mfma_flops=0.0))
        b["n"] += 1
        b["ms"] += ms
        b["flops"] += r["flops"]
        b["mfma_flops"] += r["flops"] * r["prec"]
        b["bytes"] += r["bytes"]
    gsum = {}
    for k, lst in groups.items():
        ms = sum(r["ev"][0].elapsed_ms(r["ev"][1]) for r in lst)
        gsum[k] = dict(n=len(lst), ms=ms, flops=sum(r["flops"] for r in lst), bytes=sum(r["bytes"] for r in lst))
    return by, gsum, family_replay_ms


def cpu_baseline(sd, V, S, D, cfg_scale, n_timed=3, threads=16):
    """The CPU oracle (port of the reference path, pinned to it by tests/golden) on this host's cores."""
    from mvdfusion_amd import synthetic as syn
    from oracle import ref_torch as O
    # PyTorch eager on "all host cores" collapses on a 256-core box (415 s/step measured: thread oversubscription on
    # ~3000 small ops); 16 threads is near the sweet spot and is what `cores` reports (the all-core figure is recorded by
    # `--cpu-threads 0` runs, profiles/r02_cpu_allcores.json).
    torch.set_num_threads(min(threads, os.cpu_count()) if threads > 0 else os.cpu_count())
    inp = syn.make_inputs(V, S, seed=0)
    dn, sn = syn.step_noise(V, S, D, 50, seed=0)
    tab = O.ddpm_tables()
    dd = O.ddim_schedule(tab)
    cams = lambda c: {"R": c.R, "T": c.T, "f": c.focal_length, "p": c.principal_point}
    x = inp["x_T"]
    times = []
    with torch.no_grad():
        for i in range(1 + n_timed):
            t0 = time.time()
            x, _ = O.denoise_step(sd, x, cams(inp["batch_cameras"]), inp["input_latents"], cams(inp["input_cameras"]),
                                  inp["clip_v_embed"], tab, dd, 49 - i, dn[i], sn[i], cfg_scale=cfg_scale, n_pts_per_ray=D,
                                  unet_kw=dict(image_size=S))
            times.append(time.time() - t0)
    dt = sum(times[1:]) / n_timed
    out = {"value": 1.0 / dt, "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{n_timed} timed DDIM steps (+1 warm-up) of the same V={V} S={S} workload, fp32 PyTorch eager, "
                     f"{dt:.2f} s/step (min {min(times[1:]):.2f}, max {max(times[1:]):.2f}); extrapolated 50-step sample "
                     f"{50 * dt:.0f} s"}
    # north_star's "all host cores" figure: one step of the SAME oracle with torch.set_num_threads(os.cpu_count()) takes minutes on a
    # 200+-core host (thread oversubscription on ~3000 small ops), so it is measured by tools/cpu_allcores.sh, not in every run; the
    # newest record under profiles/ rides along with the round that measured it.  `cores`-thread value above = the FASTER, fairer baseline.
    import glob
    recs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cpu_allcores.json")))
    if recs and (V, S, D) == (4, 32, 1):
        out["all_host_cores"] = dict(json.load(open(recs[-1])), measured_in=os.path.basename(recs[-1]).split("_")[0],
                                     note="same oracle, every host core: slower than the 16-thread figure above (oversubscription)")
    return out, x


def workload_label(V, S, D, N, cfg_scale):
    idx = {(4, 32): 1, (8, 32): 2, (8, 64): 3}.get((V, S))
    if N > 1:
        idx = 2 if (V, S) == (8, 32) else idx
    head = f"BASELINE.json configs[{idx}]" if idx is not None and D == 1 else "off-BASELINE workload"
    return (f"{head}: V={V} views x {8 * S}^2 images ({S}x{S} latents), D={D}, 50-step DDIM (eta 1), cfg {cfg_scale}, "
            "SD1.x UNet 320ch + 10 view-aligned transformers + GridAttn, random-init (deterministic fill) weights")


def spawn_ranks(n):
    """Re-launch this command as n ranks under torch.distributed.run (one process per GPU, RCCL over xGMI)."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    if "--dry-run" in sys.argv:
        ndev = n
    if ndev < n and not os.environ.get("MVD_DIST_SHARE_GPU"):      # (MVD_DIST_SHARE_GPU=1: ranks share devices -- functional tests only, never a measurement)
        print(f"bench.py: --gpus {n} needs {n} GPUs on this node, found {ndev}; refusing to measure fewer devices than requested",
              file=sys.stderr, flush=True)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_run(a):
    """`bench.py --gpus N --dry-run`: the PLUMBING of the N-rank job without a GPU -- rank environment, process group
    (gloo), view partition (uniform or ragged), the per-step exchange, the barrier-bracketed timed region, max over ranks, rank 0's JSON line
    with the contract's keys.  The per-rank compute is a toy update of the rank's own latent rows; the printed value is NOT a measurement
    (`data` says so).  tests/test_cpu_distributed.py runs it at N = 8 so that the first real 8-GPU launch cannot fail on plumbing."""
    from mvdfusion_amd import synthetic as syn
    from mvdfusion_amd.parallel import ViewExchange
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if a.gpus > 1 and a.gpus != world:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.set_num_threads(1)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    N = max(a.gpus, world) if world > 1 else 1
    V = a.views or (4 if N == 1 else 8)
    S, D = a.latent, a.depth_samples
    ex = ViewExchange(V) if world > 1 else None
    q0, Vq = (ex.q0, ex.Vq) if ex else (0, V)
    inp = syn.make_inputs(V, S, seed=0)
    dn, sn = syn.step_noise(V, S, D, 50, seed=0)
    x = inp["x_T"].clone()

    def steps(n, i0):
        for i in range(i0, i0 + n):
            new = 0.5 * x + 0.25 * torch.tanh(x.mean(dim=0, keepdim=True)) + 0.1 * sn[i % 50]
            x[q0:q0 + Vq] = new[q0:q0 + Vq]
            if ex is not None:
                ex.gather(x)

    def sync():
        if world > 1:
            dist.barrier()
    steps(a.warmup, 0)
    sync()
    t0 = time.perf_counter()
    steps(a.steps, a.warmup)
    sync()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    chk = x.double().sum().reshape(1)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert float(lo) == float(hi), "ranks disagree on the replicated latents"
    dt = float(t)
    if rank == 0:
        print(json.dumps({
            "metric": "denoising-steps/sec", "value": a.steps / dt, "unit": "steps/s", "n_gpus": N, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt * 1e3 / a.steps, "higher_is_better": True, "rccl_ranks": 0, "scaling": "strong", "vs_baseline": None,
            "dtype": "none (dry run)", "data": "DRY RUN: no GPU work, plumbing rehearsal on gloo -- not a measurement",
            "dry_run": True, "checksum": float(chk),
            "config": {"workload": workload_label(V, S, D, N, 2.5), "views": V, "latent": S, "depth_samples": D,
                       "parallelism": f"view-parallel: {V} views over {N} ranks " + str([r[1] for r in ex.ranges] if ex else [V]) +
                                      ", 1 all-gather of latent rows per step over torch.distributed backend gloo"}}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--dry-run", action="store_true", help="plumbing rehearsal of the N-rank job on CPU / gloo (no GPU, no measurement)")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--views", type=int, default=None)
    ap.add_argument("--latent", type=int, default=32)
    ap.add_argument("--depth-samples", type=int, default=1)
    ap.add_argument("--precision", default=None, help="f16x4 | f16x3 | bf16x3 | f16 | bf16, or a per-layer-class policy "
                    "'f16x4:conv=3,geglu=3' (mvdfusion_amd.hip.parse_precision); default: mvdfusion_amd.configs.DEFAULT_PRECISION")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16, help="threads of the CPU baseline (0 = all host cores)")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the V=8 x 256^2 single-GPU workload (north-star 1-GPU target) "
                    "that the default N=1 run times in the same process")
    ap.add_argument("--gn-two-pass", action="store_true", help="A/B: GroupNorm statistics by their own kernels instead of the producers")
    ap.add_argument("--ln-kernels", action="store_true", help="A/B: LayerNorm kernels instead of the fold into the QKV / GEGLU GEMMs")
    ap.add_argument("--gn-apply-kernels", action="store_true", help="A/B: GroupNorm apply always as its own launch (not inside the split-K reduce)")
    ap.add_argument("--train-step", action="store_true",
                    help="instead of the denoising metric: time the TRAINING step (BASELINE configs[4] per GPU: fwd + bwd + AdamW on one "
                         "scene of --views views, --depth-samples samples; tools/bench_train.py) and print its JSON line")
    ap.add_argument("--tune-cache", default=None, help="JSON file with the GEMM autotuner's choices: loaded if it exists (no "
                    "re-tuning: identical kernels across the bench run and the rocprofv3 passes), written after warm-up otherwise")
    ap.add_argument("--shard-emulate", default=None, metavar="r/N",
                    help="single GPU: also time the work of rank r of an N-way view-parallel job (Vq = V/N query views); "
                         "several ranks: '0/8,7/8'")
    a = ap.parse_args()
    if a.train_step:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_train
        sys.argv = [sys.argv[0], "--views", str(a.views or 8), "--depth-samples", str(a.depth_samples if a.depth_samples > 1 else 3),
                    "--steps", str(min(a.steps, 5))]
        return bench_train.main()

    if a.dry_run and (a.gpus == 1 or "WORLD_SIZE" in os.environ):
        return dry_run(a)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: spawn the N ranks ourselves (one process per GPU, RCCL) -- never measure one GPU and call it N
        return spawn_ranks(a.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    if world > 1 and a.gpus > 1 and a.gpus != world:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    if world > ndev and not os.environ.get("MVD_DIST_SHARE_GPU"):
        raise SystemExit(f"bench.py: {world} ranks need {world} GPUs, this node has {ndev} "
                         "(MVD_DIST_SHARE_GPU=1 lets ranks share devices for functional tests only)")
    local = local % ndev
    torch.cuda.set_device(local)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MVD_DIST_BACKEND", "nccl")      # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    N = max(a.gpus, world) if world > 1 else 1
    V = a.views or (4 if N == 1 else 8)
    S, D, cfg_scale = a.latent, a.depth_samples, 2.5

    from mvdfusion_amd import hip
    from mvdfusion_amd.configs import DEFAULT_PRECISION
    from mvdfusion_amd.parallel import ViewExchange
    a.precision = a.precision or DEFAULT_PRECISION
    p_fmt, p_default, p_policy = hip.parse_precision(a.precision)
    if a.tune_cache and os.path.exists(a.tune_cache):
        log(f"[bench] {hip.load_tuned(a.tune_cache)} tuned GEMM configurations loaded from {a.tune_cache}")
    m, sd = build(V, S, D, a.precision)
    ex = ViewExchange(V) if world > 1 else None
    q0, Vq = (ex.q0, ex.Vq) if ex else (0, None)
    eng, inp, dn, sn = prepare(m, V, S, D, cfg_scale, q0=q0, Vq=Vq)
    graph = not a.no_graph
    if a.gn_two_pass:
        from mvdfusion_amd import engine as _engine
        _engine.Ctx.gn_from_producer = False
    if a.ln_kernels:
        from mvdfusion_amd import engine as _engine
        _engine.Ctx.ln_fold = False
    if a.gn_apply_kernels:
        from mvdfusion_amd import engine as _engine
        _engine.Ctx.gn_fuse = False

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_run(e, exch):
        run_steps(e, a.warmup, cfg_scale, exch, graph)
        sync()
        ev0, ev1 = hip.Event(), hip.Event()
        t0 = time.perf_counter()
        ev0.record()
        run_steps(e, a.steps, cfg_scale, exch, graph)
        ev1.record()
        sync()
        return time.perf_counter() - t0, ev0.elapsed_ms(ev1) / a.steps

    dt, gpu_ms = timed_run(eng, ex)
    if a.tune_cache and not os.path.exists(a.tune_cache) and rank == 0:
        hip.save_tuned(a.tune_cache)
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_step = dt * 1e3 / a.steps

    out = None
    if rank == 0:
        f_unet = {32: 225.09e9, 64: 1047.09e9}.get(S, 225.09e9 * (S / 32) ** 2) if D == 1 else 235.81e9
        T = V * V * S * S * D
        f_grid = T * (3516416 + 3072 * V) + V * S * S * D * 393216 + (V + 1) * S * S * 2560
        step_flops = 2 * V * f_unet + f_grid
        out = {
            "metric": "denoising-steps/sec", "value": a.steps / dt, "unit": "steps/s", "n_gpus": N, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "rccl_ranks": world if backend == "nccl" else 0,
            # N = 1: one GPU runs the whole fixed workload -- neither weak nor strong scaling applies; the contract's field holds "strong"
            # (total work fixed) for every N so that the driver's per-N values are comparable
            "scaling": "strong", "vs_baseline": None,
            "dtype": f"{a.precision} ({p_fmt} MFMA operands" + (f" split hi+lo, {p_default} partial products" if p_default > 1 else "") +
                     (f"; layer classes at other counts: {p_policy}" if p_policy else "") + ", fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": workload_label(V, S, D, N, cfg_scale),
                       "views": V, "latent": S, "depth_samples": D, "cfg_scale": cfg_scale,
                       "parallelism": "single GPU, CFG pair batched as 2V" if N == 1 else
                       f"view-parallel: {V} views over {N} GPUs, 1 all-gather of latent rows per step over "
                       f"{'RCCL (torch.distributed backend nccl)' if backend == 'nccl' else 'torch.distributed backend ' + str(backend)}",
                       "hipgraph": graph, "weight_prefetch": weight_prefetch_label(),
                       "gemm_tuning": f"{hip.TUNED_SOURCE}; {hip.TUNED_IN_RUN} problems tuned in this run"},
            "gpu_ms_per_step_hip_events": gpu_ms,
            "algorithmic_tflop_per_step": step_flops / 1e12,
            "algorithmic_tflops": step_flops / (dt / a.steps) / 1e12,
        }
    # ---- roofline of the dominant kernel family + per-group rooflines (N == 1 only; HIP events on the launch stream)
    if world == 1:
        by, gsum, family_replay_ms = profile_kernel_groups(m, eng, cfg_scale)
        tot = sum(b["ms"] for b in by.values())
        for k, b in sorted(by.items(), key=lambda kv: -kv[1]["ms"]):
            log(f"[bench] {k:40s} launches {b['n']:4d}  total {b['ms']:8.3f} ms  avg {b['ms'] / b['n'] * 1e3:8.1f} us  "
                f"{b['flops'] / b['ms'] / 1e9:8.1f} TFLOP/s algorithmic")
        # The dominant kernel is mvd_gemm's gemm_kernel (one template, csrc/gemm_plain.hpp; the template arguments are the
        # tile / loop variants the autotuner picks per shape).  Headline = the whole family (every GEMM launch of the
        # step); `variants` lists each instantiation under the symbol rocprofv3 reports, for cross-checking
        # profiles/<round>_bench_n1_kernel_stats.csv.
        n_all = sum(b["n"] for b in by.values())
        fl_all = sum(b["flops"] for b in by.values())
        nprod = sum(b["mfma_flops"] for b in by.values()) / fl_all      # FLOP-weighted mean of the partial products per MAC (precision policy)
        by_all = sum(b["bytes"] for b in by.values())
        ach = fl_all / (tot * 1e-3)
        # HBM traffic: only from PMC passes of THIS workload AND THIS TREE (tools/pmc_traffic.sh <tag> --views V --latent S ...: the file
        # records the fingerprint of the sources it was measured on), else null
        pmc, tsrc = {}, None
        tfile = os.path.join(ROOT, "profiles", f"{ROUND}_pmc_traffic_v{V}_s{S}_d{D}.json")
        tree = tree_fingerprint()
        if os.path.exists(tfile):
            doc = json.load(open(tfile))
            if doc.get("tree") == tree:
                pmc = doc["kernels"]
                tsrc = (f"profiles/{os.path.basename(tfile)} (tree {tree}): separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of this "
                        "command and workload, (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch over the graph-replayed steps only; averaged over "
                        "every GEMM-family launch of those passes")
            else:
                tsrc = (f"profiles/{os.path.basename(tfile)} was measured on tree {doc.get('tree')}, this run is tree {tree}: counter traffic "
                        "omitted (re-run tools/pmc_traffic.sh)")
        # family figure: every GEMM-family kernel of the PMC passes (same workload; the tuner's per-shape picks differ a little from run to
        # run, so the per-variant match below may be partial -- `traffic_launch_coverage` -- while the family average stays comparable)
        fam = [v for kk, v in pmc.items() if kk.startswith(("gemm_kernel", "gemm_ws_kernel", "conv_patch_kernel"))]
        fam_n = sum(v["launches_profiled"] for v in fam)
        fam_traffic = sum(v["hbm_bytes_per_launch"] * v["launches_profiled"] for v in fam) / fam_n if fam_n else None
        variants, tr_sum, tr_n = [], 0.0, 0
        for k, b in sorted(by.items(), key=lambda kv: -kv[1]["ms"]):
            ent = pmc.get(k)
            if ent is None:      # conv_patch_kernel<BM, BN, WM, WN, NS, PI>: the host mirror does not know PI (chosen by the library)
                cand = [v for kk, v in pmc.items() if kk.startswith(k.rstrip(">"))]
                ent = cand[0] if len(cand) == 1 else {}
            t = ent.get("hbm_bytes_per_launch")
            if t is not None:
                tr_sum += t * b["n"]
                tr_n += b["n"]
            variants.append({"kernel": k, "launches_per_step": b["n"], "avg_launch_us": b["ms"] / b["n"] * 1e3,
                             "achieved": b["flops"] / (b["ms"] * 1e-3) / 1e12,
                             "frac": b["flops"] / (b["ms"] * 1e-3) / MFMA_16BIT_DENSE_PEAK,
                             "algorithmic_bytes_per_launch": b["bytes"] / b["n"], "traffic": t})
        out["roofline"] = {"bound": "mfma", "kernel": "gemm_kernel<BM, BN, WM, WN, NS, AMODE, LOOP> (all instantiations)",
                           "launches_per_step": n_all, "avg_launch_us": tot / n_all * 1e3, "achieved": ach / 1e12,
                           "peak": MFMA_16BIT_DENSE_PEAK / 1e12, "unit": "TFLOP/s", "frac": ach / MFMA_16BIT_DENSE_PEAK,
                           "traffic": fam_traffic,
                           "traffic_launch_coverage": tr_n / n_all if n_all else 0.0,
                           "traffic_unit": "bytes per launch (memory-side requests, Infinity-Cache hits included)",
                           "traffic_source": tsrc, "algorithmic_bytes_per_launch": by_all / n_all,
                           "mfma_products_per_mac": nprod, "mfma_pipe_frac": nprod * ach / MFMA_16BIT_DENSE_PEAK,
                           "note": "achieved = algorithmic FLOPs (2*M*N*K of the fp32 problem) / HIP-event time of the eager "
                                   "launches; the split-operand kernels issue `mfma_products_per_mac` MFMA products per "
                                   "algorithmic MAC, so the MFMA pipe runs at mfma_pipe_frac of the dense 16-bit peak",
                           "gemm_share_of_step_ms": tot, "variants": variants}
        # the same family under GRAPH REPLAY (no eager launch gaps inside the event pairs), measured in THIS run: every mvd_gemm launch of
        # the step replayed from one graph (profile_kernel_groups); the reduce / GroupNorm-apply kernels mvd_gemm launches ride along
        out["roofline"]["tree"] = tree
        if family_replay_ms:
            out["roofline"].update(frac_graph_replay=fl_all / (family_replay_ms * 1e-3) / MFMA_16BIT_DENSE_PEAK,
                                   graph_replay_family_ms=family_replay_ms,
                                   graph_replay_source="this run: one hipGraph of the step's mvd_gemm launches (incl. the split-K reduce / "
                                                       "GroupNorm-apply kernels they launch), min of 3 replays, one HIP-event pair each")
        rg = {}
        for k, g in gsum.items():
            if not g["n"] or g["ms"] <= 0:
                continue
            e = {"launches_per_step": g["n"], "ms_per_step": g["ms"]}
            if g["flops"]:
                ga_prod = 3 if p_policy.get("ga", p_default) == 3 else 4        # (the fused kernel runs 3 or 4 products; one-product modes: 4)
                np_g = ga_prod if k.startswith("gridattn") else (p_policy.get("attn", p_default) if k == "attention" else nprod)
                e.update(bound="mfma", achieved=g["flops"] / (g["ms"] * 1e-3) / 1e12, peak=MFMA_16BIT_DENSE_PEAK / 1e12,
                         unit="TFLOP/s", frac=g["flops"] / (g["ms"] * 1e-3) / MFMA_16BIT_DENSE_PEAK,
                         mfma_products_per_mac=np_g, mfma_pipe_frac=np_g * g["flops"] / (g["ms"] * 1e-3) / MFMA_16BIT_DENSE_PEAK)
            else:
                e.update(bound="hbm", achieved=g["bytes"] / (g["ms"] * 1e-3) / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s",
                         frac=g["bytes"] / (g["ms"] * 1e-3) / HBM_PEAK)
            rg[k] = e
        if "gridattn" in rg:
            rg["gridattn"]["note"] = ("whole GridAttn block (z-embed, token generation, aggregation transformer, pooling, 256->768); "
                                      "FLOPs = T*(3516416 + 3072*V), the aggregation tail SURVEY.md section 8(d) puts the >=40% "
                                      "MFMA target on")
        if "gridattn_fused_kernel" in rg:
            rg["gridattn_fused_kernel"]["note"] = ("g4_fused_kernel alone (csrc/gridattn_fused.hip): token generation + pre layer + 3 DiT "
                                                   "blocks + pooling in one launch; the north-star 'cross-view attention kernel'. "
                                                   "mfma_pipe_frac = fraction of the dense 16-bit MFMA peak actually issued (mfma_products_per_mac products per MAC)")
        out["roofline_groups"] = rg
        if a.shard_emulate:
            from mvdfusion_amd.parallel import view_range
            shards = []
            for item in a.shard_emulate.split(","):          # "0/8" or several ranks "0/8,7/8" (the slowest rank bounds the job)
                r_, n_ = (int(t) for t in item.split("/"))
                sq0, sVq = view_range(V, r_, n_)
                eng_s, *_ = prepare(m, V, S, D, cfg_scale, q0=sq0, Vq=sVq)
                dts, gms = timed_run(eng_s, None)
                shards.append({"rank": r_, "world": n_, "q0": sq0, "Vq": sVq, "ms_per_step": dts * 1e3 / a.steps,
                               "gpu_ms_per_step_hip_events": gms, "projected_speedup": dt / dts})
                del eng_s
            worst = max(shards, key=lambda e: e["ms_per_step"])
            out["shard_emulate"] = dict(worst, ranks=shards,
                                        note="time of ONE rank's share (its Vq query views against all V references, CFG batch "
                                             "2*Vq) on one GPU; projected_speedup = t(V views unsharded) / t(slowest emulated shard) bounds the "
                                             "N-GPU strong-scaling result (the all-gather of 20 KB latent rows is not included)")
        if not a.no_cpu_baseline:
            cb, _ = cpu_baseline(sd, V, S, D, cfg_scale, n_timed=a.cpu_steps, threads=a.cpu_threads)
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_baseline"] = out["value"] / cb["value"]
        if not a.no_secondary and (V, S, D) == (4, 32, 1):
            # BASELINE.json north_star's 1-GPU target workload (8 views x 256^2 x 50 DDIM steps), timed in this same process
            V2 = 8
            eng2, *_ = prepare(m, V2, S, D, cfg_scale)
            dt2, gms2 = timed_run(eng2, None)
            T2 = V2 * V2 * S * S * D
            fl2 = 2 * V2 * f_unet + T2 * (3516416 + 3072 * V2) + V2 * S * S * D * 393216 + (V2 + 1) * S * S * 2560
            sec = {"workload": workload_label(V2, S, D, 1, cfg_scale), "value": a.steps / dt2, "unit": "steps/s", "steps": a.steps,
                   "warmup": a.warmup, "ms_per_step": dt2 * 1e3 / a.steps, "gpu_ms_per_step_hip_events": gms2,
                   "algorithmic_tflops": fl2 / (dt2 / a.steps) / 1e12, "frac_of_dense_16bit_peak": fl2 / (dt2 / a.steps) / MFMA_16BIT_DENSE_PEAK}
            del eng2
            if not a.no_cpu_baseline:
                cb2, _ = cpu_baseline(sd, V2, S, D, cfg_scale, n_timed=2, threads=a.cpu_threads)
                sec["cpu_baseline"] = cb2
                sec["speedup_vs_cpu_baseline"] = sec["value"] / cb2["value"]
            out["secondary"] = sec
    else:
        # view-parallel speed-up reference: the same V-view workload on ONE GPU (rank 0), unsharded
        if rank == 0:
            eng1, *_ = prepare(m, V, S, D, cfg_scale)
            run_steps(eng1, a.warmup, cfg_scale, None, graph)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_steps(eng1, a.steps, cfg_scale, None, graph)
            torch.cuda.synchronize()
            v1 = a.steps / (time.perf_counter() - t1)
            out["view_parallel"] = {"views": V, "single_gpu_same_workload_steps_per_s": v1,
                                    "speedup_over_single_gpu": out["value"] / v1,
                                    "note": "strong scaling over BASELINE configs[2] (V=8 views shared by the N GPUs); the N=1 "
                                            "default run is configs[1] (V=4), a different workload -- divide by "
                                            "single_gpu_same_workload_steps_per_s (= `bench.py --views 8` on one GPU), not by "
                                            "the N=1 headline value, for the scaling efficiency"}
        dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
