"""Worker of tests/test_gpu_distributed.py::test_two_rank_ddp_training_step: one rank of a 2-rank data-parallel training step, both
ranks on the ONE GPU of the test box.  The reference's recipe (train.py:38,86-95): the model wrapped in torch's
DistributedDataParallel(find_unused_parameters=True), ``loss = model(batch, cfg); loss.backward()`` -- the HIP backward feeds DDP's
gradient hooks, which all-reduce (mean) over the process group.  Every rank checks its averaged `.grad` against the mean of the two
ranks' LOCAL gradients (computed without DDP and exchanged explicitly)."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    backend = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if backend == "nccl":       # RCCL
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    from conftest import load_golden
    from test_gpu_vae import _training_setup
    gd = dict(load_golden("train_grads_mc32_v4_d3"))
    m, batch, tc, draws = _training_setup(gd)
    g = torch.Generator().manual_seed(int(gd["batch_seed"]) + 17 * rank)          # a different scene per rank (data parallel)
    batch["images"] = torch.rand(16, 3, 256, 256, generator=g).cuda()
    batch["depths"] = torch.rand(16, 1, 256, 256, generator=g).cuda()
    for p in list(m.vae.parameters()) + list(m.clip_image_encoder.parameters()):
        p.requires_grad_(False)
    m._noise_source = draws
    # local gradients (no DDP), then their mean over the ranks
    loss_l, grads_l = m.gradients(batch, tc, noise_source=draws)
    probe = ["view_attn.final_layer_b.weight", "cc_projection.4.weight", "time_embed.2.bias",
             "unet_model.unet_model.output_blocks.11.2.aligned_attn_transformer_blocks.0.attn2.to_v.weight",
             "unet_model.unet_model.middle_block.2.aligned_attn_proj_in.weight"]
    want = {}
    for n in probe:
        t = grads_l[n].detach().clone().contiguous()
        dist.all_reduce(t)
        want[n] = t / world
    ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[local] if backend == "nccl" else None, find_unused_parameters=True)
    loss = ddp(batch, tc)
    loss.backward()
    pd = dict(m.named_parameters())
    errs = {}
    for n in probe:
        gq = pd[n].grad
        errs[n] = float((gq.reshape(want[n].shape) - want[n]).abs().max() / (want[n].abs().max() + 1e-30))
    lt = torch.tensor([float(loss.detach())], device="cuda")
    lts = [torch.zeros_like(lt) for _ in range(world)]
    dist.all_gather(lts, lt)
    out = {"rank": rank, "backend": backend, "loss_local": float(loss_l), "loss_ddp": float(loss.detach()), "losses": [float(x) for x in lts],
           "max_rel_err_vs_mean_of_local_grads": max(errs.values()), "n_grads": sum(p.grad is not None for p in m.parameters())}
    dist.barrier()
    print("DISTJSON " + json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
