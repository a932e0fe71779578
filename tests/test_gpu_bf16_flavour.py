"""The bf16 flavour of the library (libmvd_hip_bf16.so: the same sources with -DMVD_OPERAND_BF16, bf16 MFMA operands hi + lo, three
partial products) -- BASELINE.json configs[3] names bf16.  The operand type is fixed per process (MVD_OPERAND_FORMAT is read at
import), so the op-level tests and one denoising-step golden are re-run in a SUBPROCESS with MVD_OPERAND_FORMAT=bf16: every kernel of
that .so executes and is checked against the same references with the bf16x3 tolerances the tests carry."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=1500):
    env = dict(os.environ, MVD_OPERAND_FORMAT="bf16")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    tail = (r.stdout[-2500:] + "\n" + r.stderr[-1500:])
    assert r.returncode == 0, tail
    return r.stdout


def test_bf16_flavour_op_tests():
    """Every kernel of libmvd_hip_bf16.so against the op references (the op file without its two cfg matrices; MVD_TEST_FULL=1: all of it)."""
    from conftest import FULL
    # (default: every op test except the two cfg matrices -- a property of the templates, exercised by the default flavour, and every
    #  tuner-selected cfg of the bf16 library still runs inside the step goldens below; MVD_TEST_FULL=1: the whole file)
    sel = [] if FULL else ["-k", "not (test_gemm_configurations_agree or test_gemm_groupnorm_statistics)"]
    out = _run([os.path.join("tests", "test_gpu_ops.py")] + sel)
    assert " passed" in out and "failed" not in out, out[-800:]


def test_bf16_flavour_denoise_step_and_gridattn_goldens():
    """One full-width and one reduced-width denoising step, the fused GridAttn at V = 4 and V = 15 and the graph replay, against the
    reference's goldens in the bf16 flavour (MVD_TEST_FULL=1: every step / GridAttn golden, 95 s)."""
    from conftest import FULL
    k = "test_denoise_step_vs_reference_golden or test_gridattn_vs_reference_golden or test_graph_replay_equals_eager" if FULL else \
        "step_mc320_v4_d1-320 or step_mc32_v4_d1-32 or gridattn_v4_d1-4 or gridattn_v15_d1 or test_graph_replay_equals_eager"
    out = _run([os.path.join("tests", "test_gpu_model.py"), "-k", k])
    assert " passed" in out and "failed" not in out, out[-800:]
    import ctypes
    from mvdfusion_amd import hip
    assert ctypes.CDLL(hip.LIB_PATHS["bf16"]).mvd_operand_format() == 0xbf16
