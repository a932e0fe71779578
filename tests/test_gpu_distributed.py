"""View-parallel path on the GPU: TWO ranks (one process each, both on this box's single GPU) run
sample_view_parallel -- StepEngine shards + captured hipGraphs + the ViewExchange all-gather of the updated latent rows -- over
RCCL (backend "nccl").  Some RCCL builds refuse two ranks on one device; then the same job runs over gloo (CUDA tensors staged
through the host) and the test records which backend carried it.  The 8-GPU run itself belongs to the driver."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(backend, V, steps, worker="_dist_gpu_worker.py", nproc=2, extra=()):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", worker), backend, str(V), str(steps), *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    recs, dec, pos = [], json.JSONDecoder(), 0          # (the two ranks' lines may interleave on one line)
    while True:
        pos = r.stdout.find("DISTJSON ", pos)
        if pos < 0:
            break
        pos += len("DISTJSON ")
        try:
            obj, _ = dec.raw_decode(r.stdout, pos)
            recs.append(obj)
        except json.JSONDecodeError:
            pass
    return r, recs


@pytest.mark.parametrize("V", [4])
def test_two_rank_view_parallel_on_one_gpu(V):
    steps = 3
    r, recs = _run("nccl", V, steps)
    used = "nccl"
    if r.returncode != 0 or len(recs) != 2:
        print("RCCL with two ranks on one device did not run here:\n" + (r.stderr[-1500:] or r.stdout[-1500:]))
        r, recs = _run("gloo", V, steps)
        used = "gloo"
    assert r.returncode == 0 and len(recs) == 2, r.stderr[-3000:]
    print(f"view-parallel 2-rank job carried by backend: {used}")
    rec0 = [x for x in recs if x["rank"] == 0][0]
    assert all(x["replicas_identical"] for x in recs)
    assert rec0["finite"] and rec0["rmse"] < 1e-5 and rec0["max_abs_diff"] < 1e-4, rec0
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        json.dump({"backend": used, "records": recs}, open(os.path.join(out, "dist_2rank_one_gpu.json"), "w"))


def test_two_rank_view_parallel_feed_prev_depth():
    """feed_prev_depth (mvdfusion/sampler.py:83-84,135-140) under view sharding: two ranks, from the second iteration on each
    rank's GridAttn samples depth around ITS OWN previous x0 rows; the sharded trajectory must equal the unsharded sampler's (which is
    pinned to the reference's own loop by test_sampler_feed_prev_depth_vs_reference_golden), and differ from the plain algorithm's."""
    steps, V = 3, 4
    r, recs = _run("nccl", V, steps, extra=("feed_prev_depth",))
    if r.returncode != 0 or len(recs) != 2:
        r, recs = _run("gloo", V, steps, extra=("feed_prev_depth",))
    assert r.returncode == 0 and len(recs) == 2, r.stderr[-3000:]
    rec0 = [x for x in recs if x["rank"] == 0][0]
    assert all(x["replicas_identical"] and x["feed_prev_depth"] for x in recs)
    assert rec0["finite"] and rec0["rmse"] < 1e-5 and rec0["max_abs_diff"] < 1e-4, rec0


def test_view_parallel_feed_prev_depth_single_process():
    """The same property without a process group (world 1): sample_view_parallel honours feed_prev_depth, resets the engine's depth
    mode afterwards, and the flag changes the trajectory."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import build_model, rmse
    from mvdfusion_amd import synthetic as syn
    from mvdfusion_amd.parallel import sample_view_parallel
    m = build_model(32)
    V, S, D, steps = 4, 32, 1, 3
    inp = syn.make_inputs(V, S, seed=4)
    dn, sn = syn.step_noise(V, S, D, 50, seed=4)
    args = (inp["batch_cameras"], inp["input_latents"], inp["input_cameras"], inp["clip_v_embed"])
    outs = {}
    for flag in (False, True):
        m.ddim.feed_prev_depth = flag
        try:
            xs = sample_view_parallel(m, *args, 2.5, inp["x_T"].cuda(), dn, sn, num_steps=steps)
            assert m.engine(V, S, D, True, q0=0, Vq=V).depth_mode == 0
            m.ddim.noise_source = lambda *a: (dn, sn)
            x1 = m.ddim.sample(*args, unconditional_scale=2.5, depth=True, verbose=False, x_T=inp["x_T"].cuda(), num_steps=steps)
        finally:
            m.ddim.feed_prev_depth, m.ddim.noise_source = False, None
        assert rmse(xs, x1) < 1e-6, (flag, rmse(xs, x1))
        outs[flag] = xs.cpu()
    assert rmse(outs[True], outs[False]) > 1e-4


def test_two_rank_ddp_training_step():
    """BASELINE configs[4] in miniature: the reference's DDP recipe (train.py:38, 86-95) with the HIP forward + backward: two ranks,
    a different scene each, torch's DistributedDataParallel around the model; after loss.backward() every rank's `.grad` equals the
    mean of the two ranks' local gradients (RCCL when it accepts two ranks on one device, else gloo)."""
    r, recs = _run("nccl", 0, 0, worker="_dist_ddp_worker.py")
    used = "nccl"
    if r.returncode != 0 or len(recs) != 2:
        r, recs = _run("gloo", 0, 0, worker="_dist_ddp_worker.py")
        used = "gloo"
    assert r.returncode == 0 and len(recs) == 2, (r.stderr[-3000:], r.stdout[-1500:])
    print(f"DDP training step carried by backend: {used}: {recs}")
    for x in recs:
        assert x["max_rel_err_vs_mean_of_local_grads"] < 1e-5, x
        assert abs(x["loss_local"] - x["loss_ddp"]) < 1e-6 and x["n_grads"] >= 300
    assert abs(recs[0]["losses"][0] - recs[0]["losses"][1]) > 1e-6          # the ranks really trained on different scenes
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        json.dump({"backend": used, "records": recs}, open(os.path.join(out, "dist_ddp_2rank_one_gpu.json"), "w"))


def test_rccl_one_rank_view_parallel():
    """RCCL itself (backend "nccl", device_id bound) carries the view-parallel job: a ONE-rank process group on this box's GPU runs
    sample_view_parallel with the in-place all-gather of ViewExchange forced on after every captured step -- no gloo fallback here."""
    r, recs = _run("nccl", 4, 3, nproc=1)
    assert r.returncode == 0 and len(recs) == 1, (r.stderr[-3000:], r.stdout[-1500:])
    rec = recs[0]
    assert rec["backend"] == "nccl" and rec["replicas_identical"] and rec["finite"]
    assert rec["rmse"] < 1e-5 and rec["max_abs_diff"] < 1e-4, rec
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        json.dump({"backend": "nccl", "world": 1, "records": recs}, open(os.path.join(out, "dist_rccl_1rank.json"), "w"))


def test_rccl_one_rank_ddp_training_step():
    """The DDP-wrapped training step (train.py:38, 86-95) over RCCL with a one-rank group: DDP's bucketed all-reduce runs through RCCL
    and leaves the gradients equal to the local ones."""
    r, recs = _run("nccl", 0, 0, worker="_dist_ddp_worker.py", nproc=1)
    assert r.returncode == 0 and len(recs) == 1, (r.stderr[-3000:], r.stdout[-1500:])
    x = recs[0]
    assert x["backend"] == "nccl" and x["max_rel_err_vs_mean_of_local_grads"] < 1e-5 and x["n_grads"] >= 300, x
    assert abs(x["loss_local"] - x["loss_ddp"]) < 1e-6
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        json.dump({"backend": "nccl", "world": 1, "records": recs}, open(os.path.join(out, "dist_ddp_rccl_1rank.json"), "w"))


@pytest.mark.skipif(os.environ.get("MVD_TEST_FULL") != "1", reason="50 s: MVD_TEST_FULL=1 (the N > 1 JSON contract is covered on CPU by "
                    "tests/test_cpu_distributed.py::test_bench_gpus8_dry_run_contract, the 2-rank data path by the tests above)")
def test_bench_two_ranks_json_contract():
    """`python bench.py --gpus 2` end to end on this box's single GPU (MVD_DIST_SHARE_GPU=1: functional only): bench.py spawns the two
    ranks itself under torch.distributed.run, the ranks shard BASELINE configs[2]'s 8 views, exchange latent rows once per step, and
    rank 0 prints ONE JSON line with the multi-GPU contract the driver reads at N > 1 (n_gpus, scaling, rccl_ranks / backend,
    config.parallelism, view_parallel.speedup_over_single_gpu).  RCCL when it accepts two ranks on one device, else gloo."""
    bench = os.path.join(ROOT, "bench.py")
    used, line, r = None, None, None
    # (RCCL refuses two ranks on one device -- known on the 1-GPU test boxes, 30 s per attempt -- so it is only tried where two exist)
    for backend in (("nccl", "gloo") if torch.cuda.device_count() >= 2 else ("gloo",)):
        env = dict(os.environ, MVD_DIST_SHARE_GPU="1", MVD_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
        r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                           capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode == 0 and len(lines) == 1:
            used, line = backend, lines[0]
            break
        print(f"bench.py --gpus 2 over {backend} did not run here:\n" + r.stderr[-1200:])
    assert line is not None, r.stderr[-3000:]
    j = json.loads(line)
    assert j["metric"] == "denoising-steps/sec" and j["n_gpus"] == 2 and j["steps"] == 3 and j["scaling"] == "strong"
    assert j["rccl_ranks"] == (2 if used == "nccl" else 0)
    assert j["config"]["views"] == 8 and "view-parallel: 8 views over 2 GPUs" in j["config"]["parallelism"]
    vp = j["view_parallel"]
    assert vp["views"] == 8 and vp["single_gpu_same_workload_steps_per_s"] > 0 and vp["speedup_over_single_gpu"] > 0
    assert j["value"] > 1.0 and j["vs_baseline"] is None and j["data"] == "synthetic"
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        json.dump({"backend": used, "line": j}, open(os.path.join(out, "bench_2rank_one_gpu.json"), "w"))
