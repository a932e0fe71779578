"""Build libmvd_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

    python -m mvdfusion_amd.csrc.build [--force]
"""
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["api.hip", "gemm.hip", "gemm_plain_t0.hip", "gemm_plain_t1.hip", "gemm_plain_t2.hip", "gemm_plain_t3.hip", "gemm_plain_t4.hip",
           "gemm_ws.hip", "gemm_patch.hip", "norm.hip", "attention.hip", "elementwise.hip", "gridattn.hip", "gridattn_fused.hip", "backward.hip"]
LIB = os.path.join(HERE, "libmvd_hip.so")            # fp16 MFMA operands (default)
LIB_BF16 = os.path.join(HERE, "libmvd_hip_bf16.so")  # bf16 MFMA operands (-DMVD_OPERAND_BF16)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    common = [os.path.join(HERE, "common.hpp"), os.path.join(HERE, "gridattn_common.hpp"), os.path.join(HERE, "..", "..", "include", "mvd_hip.h")]
    gemm_hdrs = [os.path.join(HERE, h) for h in ("gemm_common.hpp", "gemm_device.hpp", "gemm_plain.hpp")]      # (the GEMM units only)
    flavours = [("", [], LIB), ("_bf16", ["-DMVD_OPERAND_BF16"], LIB_BF16)]
    jobs, links = [], []
    for suffix, defs, lib in flavours:
        objs = []
        for src in SOURCES:
            s_ = os.path.join(HERE, src)
            o = os.path.join(HERE, src.replace(".hip", suffix + ".o"))
            objs.append(o)
            if force or _stale(o, [s_] + common + (gemm_hdrs if src.startswith("gemm") else [])):
                jobs.append([hipcc] + FLAGS + defs + ["-c", s_, "-o", o])
        links.append((lib, objs))

    def run(cmd):
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[{time.time() - t0:5.1f} s] " + " ".join(cmd), flush=True)
            if r.stderr.strip():
                print(r.stderr)

    jobs.sort(key=lambda c: not os.path.basename(c[-3]).startswith("gemm_"))      # the long compiles (GEMM kernel families, ~20 s each) first
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(run, jobs))
    for lib, objs in links:
        if jobs or force or _stale(lib, objs):
            run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
