"""Tune the GEMM configurations of the BASELINE workloads ONCE on a GPU box and write the committed cache mvdfusion_amd/tuned/gemm_<fmt>.json
(hip.default_tuned_path): bench.py, the step traces and the counter passes then launch one kernel mix.

    MVD_TUNE_CACHE=0 MVD_TUNE_SPLITS=1 python tools/tune_all.py [--quick]

Workloads: configs[1] V=4; configs[2] V=8 unsharded and its 8- / 4- / 2-way shards (one rank each: every rank of a uniform split meets the
same problem shapes); the as-shipped V=15 and its ragged 8-way shards (Vq = 2 and 1); configs[3] V=8 at 64x64 latents; D=3 (configs[4]'s
forward shapes at V=8).  Each problem is timed with cold weights (hip._autotune) including explicit split-K counts."""
import os
import sys
import time

os.environ.setdefault("MVD_TUNE_CACHE", "0")
os.environ.setdefault("MVD_TUNE_SPLITS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from mvdfusion_amd import hip


def main():
    quick = "--quick" in sys.argv
    jobs = [(4, 32, 1, None), (8, 32, 1, None), (8, 32, 1, (0, 1)), (8, 32, 1, (0, 2)), (8, 32, 1, (0, 4))]
    if not quick:
        jobs += [(15, 32, 1, None), (15, 32, 1, (0, 2)), (15, 32, 1, (14, 1)), (8, 64, 1, None), (8, 32, 3, None)]
    models = {}
    t00 = time.time()
    for V, S, D, shard in jobs:
        key = (S, D)
        if key not in models:
            models.clear()
            torch.cuda.empty_cache()
            models[key] = bench.build(V, S, D, os.environ.get("PREC", "f16x3"))[0]
        m = models[key]
        t0 = time.time()
        n0 = len(hip._TUNED)
        q0, Vq = shard if shard else (0, None)
        eng, *_ = bench.prepare(m, V, S, D, 2.5, q0=q0, Vq=Vq)
        bench.run_steps(eng, 2, 2.5, None, True)
        torch.cuda.synchronize()
        print(f"[tune_all] V={V} S={S} D={D} shard={shard}: {len(hip._TUNED) - n0} new problems, {time.time() - t0:.0f} s", flush=True)
        m._engines.clear()
        del eng
        hip.save_tuned(hip.default_tuned_path(), merge=False)
    print(f"[tune_all] {len(hip._TUNED)} problems -> {hip.default_tuned_path()} in {time.time() - t00:.0f} s (GEMM sources {hip.gemm_fingerprint()})")


if __name__ == "__main__":
    main()
