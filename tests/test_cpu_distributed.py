"""world_size-2 gloo tests (CPU) of the view-parallel path: partitioning, the per-step all-gather exchange, host-noise
slicing.  The per-rank compute is the CPU oracle here (test infrastructure); on the GPU the same loop drives StepEngine."""
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


def _worker(rank, world, port, V, steps, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
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


@pytest.mark.parametrize("V", [4, 3])
def test_view_parallel_matches_single_process(V):
    from mvdfusion_amd import synthetic as syn
    from oracle import ref_torch as O
    steps, world = 2, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, V, steps, q)) for r in range(world)]
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
    torch.set_num_threads(2)          # same intra-op partitioning (= summation order) as the workers
    try:
        with torch.no_grad():
            for i in range(steps):
                x, _ = O.denoise_step(sd, x, cams(inp["batch_cameras"]), inp["input_latents"], cams(inp["input_cameras"]),
                                      inp["clip_v_embed"], tab, dd, 49 - i, dn[i], sn[i], cfg_scale=2.5,
                                      unet_kw=dict(model_channels=32))
    finally:
        torch.set_num_threads(nthreads)
    owned = sorted((q0, n) for _, q0, n, _ in res)
    assert owned[0][0] == 0 and owned[0][0] + owned[0][1] == owned[1][0] and owned[1][0] + owned[1][1] == V
    for _, _, _, xr in res:
        assert not torch.isnan(xr).any()
        assert rel_err(xr, x) < 5e-5
