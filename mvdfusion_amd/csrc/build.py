"""Build libmvd_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

    python -m mvdfusion_amd.csrc.build [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["api.hip", "gemm.hip", "gemm_pt.hip", "norm.hip", "attention.hip", "elementwise.hip", "gridattn.hip", "gridattn_fused.hip", "backward.hip"]
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
    common = [os.path.join(HERE, "common.hpp"), os.path.join(HERE, "gridattn_common.hpp"), os.path.join(HERE, "gemm_common.hpp"),
              os.path.join(HERE, "..", "..", "include", "mvd_hip.h")]
    flavours = [("", [], LIB), ("_bf16", ["-DMVD_OPERAND_BF16"], LIB_BF16)]
    jobs, links = [], []
    for suffix, defs, lib in flavours:
        objs = []
        for src in SOURCES:
            s_ = os.path.join(HERE, src)
            o = os.path.join(HERE, src.replace(".hip", suffix + ".o"))
            objs.append(o)
            if force or _stale(o, [s_] + common):
                jobs.append([hipcc] + FLAGS + defs + ["-c", s_, "-o", o])
        links.append((lib, objs))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(run, jobs))
    for lib, objs in links:
        if jobs or force or _stale(lib, objs):
            run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
