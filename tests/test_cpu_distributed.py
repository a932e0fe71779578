"""gloo tests (CPU) of the view-parallel path at world sizes 2 AND 8 (the target machine: 8 ranks; V = 8 one view per rank, V = 15 -- the
as-shipped view count -- as the ragged 2,2,2,2,2,2,2,1 split): partitioning, the per-step all-gather exchange, host-noise slicing.  The
per-rank compute is the CPU oracle here (test infrastructure); on the GPU the same loop drives StepEngine."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_spec, rel_err


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy_step(i, x, noise):
    """A cheap stand-in for the denoiser with the SAME data dependencies: the new row of view v needs every view's current row (GridAttn
    reads all V latents) and that view's slice of the full-V noise."""
    return 0.5 * x + 0.25 * torch.tanh(x.mean(dim=0, keepdim=True) + x.roll(1, 0)) + 0.1 * noise[i]


def _toy_worker(rank, world, port, V, steps, q):
    """world-8 runs: the exchange, the ragged partition and the noise slicing are what is under test, so the per-rank compute is the toy
    step (eight oracle processes of a V = 15 step are minutes of CPU time and test nothing more)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mvdfusion_amd import synthetic as syn
        from mvdfusion_amd.parallel import ViewExchange, run_view_parallel
        inp = syn.make_inputs(V, 32, seed=4)
        dn, sn = syn.step_noise(V, 32, 1, 50, seed=4)          # every rank draws the FULL noise, same seed
        ex = ViewExchange(V)

        def local_step(i, x):
            xn = _toy_step(i, x, sn)
            poison = torch.full_like(x, float("nan"))           # rows this rank does not own must come from the peers
            poison[ex.q0:ex.q0 + ex.Vq] = xn[ex.q0:ex.q0 + ex.Vq]
            x.copy_(poison)

        x = run_view_parallel(inp["x_T"].clone(), steps, local_step, ex)
        q.put((rank, ex.q0, ex.Vq, x))
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, V, steps, q, threads=2):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(threads)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mvdfusion_amd import synthetic as syn
        from mvdfusion_amd.parallel import ViewExchange, run_view_parallel
        from oracle import ref_torch as O
        sd = syn.det_fill_state_dict(load_spec(32))
        inp = syn.make_inputs(V, 32, seed=4)
        dn, sn = syn.step_noise(V, 32, 1, 50, seed=4)          # every rank draws the FULL noise, same seed
        tab = O.ddpm_tables()
        dd = O.ddim_schedule(tab)
        cams = lambda c: {"R": c.R, "T": c.T, "f": c.focal_length, "p": c.principal_point}
        ex = ViewExchange(V)

        def local_step(i, x):
            with torch.no_grad():
                xn, _ = O.denoise_step(sd, x.clone(), cams(inp["batch_cameras"]), inp["input_latents"],
                                       cams(inp["input_cameras"]), inp["clip_v_embed"], tab, dd, 49 - i, dn[i], sn[i],
                                       cfg_scale=2.5, unet_kw=dict(model_channels=32))
            poison = torch.full_like(x, float("nan"))           # rows this rank does not own must come from the peers
            poison[ex.q0:ex.q0 + ex.Vq] = xn[ex.q0:ex.q0 + ex.Vq]
            x.copy_(poison)

        x = run_view_parallel(inp["x_T"].clone(), steps, local_step, ex)
        q.put((rank, ex.q0, ex.Vq, x))
    finally:
        dist.destroy_process_group()


def _spawn(target, world, args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _check_partition(res, V, world):
    owned = sorted((q0, n) for _, q0, n, _ in res)
    assert len(owned) == world and owned[0][0] == 0 and owned[-1][0] + owned[-1][1] == V
    assert all(owned[i][0] + owned[i][1] == owned[i + 1][0] for i in range(world - 1)) and all(n >= 1 for _, n in owned)
    return [n for _, n in owned]


@pytest.mark.parametrize("V,world,shards", [(8, 8, [1] * 8), (15, 8, [2, 2, 2, 2, 2, 2, 2, 1]), (16, 8, [2] * 8), (9, 4, [3, 2, 2, 2])])
def test_view_parallel_world8_exchange(V, world, shards):
    """The 8-rank job of BASELINE configs[2] (one view per GPU) and the as-shipped V = 15 (ragged: ONE padded all-gather per step,
    parallel.ViewExchange.gather) on gloo: every rank ends with the single-process trajectory, bit for bit (the exchange only copies)."""
    from mvdfusion_amd import synthetic as syn
    steps = 4
    res = _spawn(_toy_worker, world, (V, steps))
    assert _check_partition(res, V, world) == shards
    inp = syn.make_inputs(V, 32, seed=4)
    dn, sn = syn.step_noise(V, 32, 1, 50, seed=4)
    x = inp["x_T"].clone()
    torch.set_num_threads(1)
    for i in range(steps):
        x = _toy_step(i, x, sn)
    for _, _, _, xr in res:
        assert torch.equal(xr, x)


@pytest.mark.parametrize("V,world", [(4, 2), (3, 2)])
def test_view_parallel_matches_single_process(V, world):
    from mvdfusion_amd import synthetic as syn
    from oracle import ref_torch as O
    steps = 2
    threads = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, V, steps, q, threads)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sd = syn.det_fill_state_dict(load_spec(32))
    inp = syn.make_inputs(V, 32, seed=4)
    dn, sn = syn.step_noise(V, 32, 1, 50, seed=4)
    tab = O.ddpm_tables()
    dd = O.ddim_schedule(tab)
    cams = lambda c: {"R": c.R, "T": c.T, "f": c.focal_length, "p": c.principal_point}
    x = inp["x_T"]
    nthreads = torch.get_num_threads()
    torch.set_num_threads(threads)    # same intra-op partitioning (= summation order) as the workers
    try:
        with torch.no_grad():
            for i in range(steps):
                x, _ = O.denoise_step(sd, x, cams(inp["batch_cameras"]), inp["input_latents"], cams(inp["input_cameras"]),
                                      inp["clip_v_embed"], tab, dd, 49 - i, dn[i], sn[i], cfg_scale=2.5,
                                      unet_kw=dict(model_channels=32))
    finally:
        torch.set_num_threads(nthreads)
    _check_partition(res, V, world)
    for _, _, _, xr in res:
        assert not torch.isnan(xr).any()
        assert rel_err(xr, x) < 5e-5


@pytest.mark.parametrize("V,shards", [(8, [1] * 8), (15, [2, 2, 2, 2, 2, 2, 2, 1])])
def test_bench_gpus8_dry_run_contract(V, shards):
    """`python bench.py --gpus 8 --dry-run`: the driver's 8-rank launch rehearsed on gloo -- self-spawned ranks under torch.distributed.run,
    the partition, one exchange per step, max-over-ranks timing, ONE JSON line from rank 0 carrying the contract's keys."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry-run", "--steps", "6", "--warmup", "2",
                        "--views", str(V)], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    doc = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config"):
        assert k in doc, k
    assert doc["n_gpus"] == 8 and doc["steps"] == 6 and doc["warmup"] == 2 and doc["dry_run"] is True and "DRY RUN" in doc["data"]
    assert doc["config"]["views"] == V and str(shards) in doc["config"]["parallelism"]
    assert abs(doc["value"] - 6 / (doc["ms_per_step"] * 6e-3)) < 1e-6 * doc["value"]
    # and the driver's own launch form (ranks handed in by torch.distributed.run) at world 2
    env = dict(os.environ)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3",
                        "--warmup", "1", "--views", str(V)], capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    doc2 = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert doc2["n_gpus"] == 2
