"""Worker of tests/test_gpu_distributed.py: one rank of a 2-rank view-parallel sampling job, every rank on the ONE GPU of the
test box (LOCAL_RANK % device_count).  Runs mvdfusion_amd.parallel.sample_view_parallel (StepEngine + hipGraph + ViewExchange
all-gather) for a few DDIM steps; rank 0 also runs the unsharded engine and prints the comparison as one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    backend, V, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if backend == "nccl":       # RCCL
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    from conftest import build_model
    from mvdfusion_amd import synthetic as syn
    from mvdfusion_amd.parallel import sample_view_parallel
    m = build_model(32)
    m.ddim.feed_prev_depth = len(sys.argv) > 4 and sys.argv[4] == "feed_prev_depth"      # sampler.py:83-84,135-140 under sharding
    S, D = 32, 1
    inp = syn.make_inputs(V, S, seed=4)
    dn, sn = syn.step_noise(V, S, D, 50, seed=4)              # every rank draws the FULL noise from the same seed
    x = sample_view_parallel(m, inp["batch_cameras"], inp["input_latents"], inp["input_cameras"], inp["clip_v_embed"], 2.5,
                             inp["x_T"].cuda(), dn, sn, num_steps=steps, use_graph=True,
                             force_collective=world == 1)      # one-rank RCCL job: still run the all-gather
    torch.cuda.synchronize()
    # every rank must hold the same full latent tensor after the last exchange
    ref = x.clone()
    dist.broadcast(ref, src=0)
    same = bool(torch.equal(ref, x))
    out = {"rank": rank, "backend": backend, "replicas_identical": same}
    if rank == 0:
        m.ddim.noise_source = lambda *a: (dn, sn)
        x1 = m.ddim.sample(inp["batch_cameras"], inp["input_latents"], inp["input_cameras"], inp["clip_v_embed"],
                           unconditional_scale=2.5, depth=True, verbose=False, x_T=inp["x_T"].cuda(), num_steps=steps)
        d = (x - x1).double()
        out.update(max_abs_diff=float(d.abs().max()), rmse=float((d ** 2).mean().sqrt()), finite=bool(torch.isfinite(x).all()))
    m.ddim.feed_prev_depth = False
    out["feed_prev_depth"] = len(sys.argv) > 4 and sys.argv[4] == "feed_prev_depth"
    dist.barrier()
    print("DISTJSON " + json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
