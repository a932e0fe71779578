"""Per-layer-class precision sensitivity: each class of hip.PREC_KINDS at three partial products with the rest at
four, 50-step full-width stochastic trajectory (V = 4) against the float64 evaluation and the fp32 oracle trajectory of the reference
algorithm (tests/golden/traj_mc320_v4_d1_50steps*.npz), plus the time of the 50 replayed steps.  Every policy is run `--reps` times with
a fresh GEMM autotuning (other tile / split-K picks = other fp32 summation orders): the spread between repetitions is the noise floor of
the chaotic random-weight trajectory, which the per-class differences have to be read against.

    python tools/prec_sweep.py [--reps 2] [--policies f16x4 f16x3 f16x4:conv=3 ...] > gpurun_out/prec_sweep.json
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from conftest import build_model, load_golden, rmse
    import test_gpu_model as T
    from mvdfusion_amd import hip
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--policies", nargs="*", default=None)
    a = ap.parse_args()
    kinds = [k for k in hip.PREC_KINDS if k != "xattn"]          # (xattn: the D > 1 pixel cross-attention, absent at D = 1)
    pols = a.policies or (["f16x4", "f16x3"] + [f"f16x4:{k}=3" for k in kinds] +
                          ["f16x4:conv=3,geglu=3", "f16x4:conv=3,geglu=3,ffproj=3", "f16x3:attn=4,out=4,proj=4,qkv=4"])
    gd, g32 = load_golden("traj_mc320_v4_d1_50steps_f64"), load_golden("traj_mc320_v4_d1_50steps")
    fmts = {hip.parse_precision(pol)[0] for pol in pols}
    assert len(fmts) == 1, f"one operand format per process (library flavour): {fmts}"
    m = build_model(320, precision="bf16x3" if fmts == {"bf16"} else "f16x4")
    rows = []
    for pol in pols:
        _, m.precision, m.precision_policy = hip.parse_precision(pol)
        for rep in range(a.reps):
            m._engines.clear()
            hip._TUNED.clear()
            _, _, _, x, inter = T._sample(m, 4, 1, 11, 50)          # tunes + captures
            torch.cuda.synchronize()
            t0 = time.time()
            _, _, _, x, inter = T._sample(m, 4, 1, 11, 50)
            torch.cuda.synchronize()
            dt = time.time() - t0
            e64 = [rmse(inter[int(k)]["xt"], gd["xs"][j]) for j, k in enumerate(gd["kept"])]
            e32 = [rmse(inter[int(k)]["xt"], g32["xs"][j]) for j, k in enumerate(gd["kept"])]
            row = dict(policy=pol, rep=rep, s_per_50_steps=dt, max_rmse_vs_f64=max(e64), max_rmse_vs_fp32=max(e32),
                       final_rmse_vs_f64=e64[-1])
            rows.append(row)
            print(json.dumps(row), file=sys.stderr, flush=True)
    print(json.dumps(dict(fp32_oracle_vs_f64=[float(e) for e in gd["fp32_oracle_rmse"]], rows=rows)))


if __name__ == "__main__":
    main()
