import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


# Order of the GPU suite: the reference-golden tests of every SURVEY section-8 row first (whole steps / UNet /
# GridAttn, then VAE / prepare_batch / training gradients), the op-level backward and distributed tests next, the op matrices last, and
# the bf16 flavour's subprocess at the very end -- a failure or a slow box costs the least evidence that way.
_FILE_ORDER = ("test_gpu_model.py", "test_gpu_vae.py", "test_gpu_backward.py", "test_gpu_distributed.py", "test_gpu_ops.py",
               "test_gpu_bf16_flavour.py")
# MVD_TEST_FULL=1: the exhaustive cfg x split-K matrices of tests/test_gpu_ops.py (every configuration under every split mode) and the
# whole op file in the bf16 flavour; default: every configuration once, split modes in rotation (suite budget: <= 600 s on the GPU box).
FULL = os.environ.get("MVD_TEST_FULL") == "1"


def cfg_splitk_matrix(cfgs, splitks=(1, 0, 1, 3)):
    """(cfg, splitk) pairs for the GEMM configuration matrices: all of them under MVD_TEST_FULL=1, else every cfg once with the split
    modes in rotation (1 = no split: the bit-equality leg; 0 = the library's model; 3 = forced)."""
    cfgs = list(cfgs)
    if FULL:
        return [(c, s) for c in cfgs for s in sorted(set(splitks))]
    return [(c, splitks[i % len(splitks)]) for i, c in enumerate(cfgs)]


def pytest_collection_modifyitems(config, items):
    rank = {name: i for i, name in enumerate(_FILE_ORDER)}
    items.sort(key=lambda it: rank.get(os.path.basename(str(it.fspath)), -1))       # stable: the order inside a file is kept; CPU files first
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiu" and z[k].ndim > 0 else z[k]) for k in z.files}


def load_spec(mc):
    return [(k, tuple(s)) for k, s in json.load(open(os.path.join(GOLD, f"state_dict_spec_mc{mc}.json")))]


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def planes_to_float(p):
    """split planes (rows, 2*K) int16, per row K/32 blocks of [32 hi | 32 lo] (fp16 or bf16) -> fp32 (rows, K)."""
    from mvdfusion_amd import hip
    p = p.detach().cpu()
    rows, k2 = p.shape
    q = p.view(rows, k2 // 64, 2, 32)
    if hip.OPERAND_FORMAT == "bf16":
        f = lambda t: (t.to(torch.int32) << 16).view(torch.float32)
    else:
        f = lambda t: t.view(torch.float16).float()
    return (f(q[:, :, 0, :].contiguous()) + f(q[:, :, 1, :].contiguous())).reshape(rows, k2 // 2)


def rmse(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float(((a - b) ** 2).mean().sqrt())


from mvdfusion_amd.configs import UNET_PARAMS  # noqa: E402,F401  (the yaml's model block lives in the package: mvdfusion_amd/configs.py)
from mvdfusion_amd import configs as _configs  # noqa: E402

DEFAULT_PRECISION = "bf16x3" if os.environ.get("MVD_OPERAND_FORMAT") == "bf16" else _configs.DEFAULT_PRECISION


def model_config(mc=320, D=1, S=32, precision=None):
    return _configs.model_config(mc, D, S, precision or DEFAULT_PRECISION)


_MODELS = {}


def build_model(mc=320, D=1, S=32, precision=None):
    """ViewFusion on cuda:0 with the deterministic non-zero fill (cached per configuration)."""
    from mvdfusion_amd import synthetic as syn
    from mvdfusion_amd.viewfusion_zero_depth_rgb import ViewFusion
    precision = precision or DEFAULT_PRECISION
    key = (mc, D, S, precision)
    if key not in _MODELS:
        with syn.skip_default_init():            # (every parameter is overwritten by the fill: the default initialisers are 9 s per full-width build)
            m = ViewFusion(**model_config(mc, D, S, precision))
        syn.fill_module_(m)
        _MODELS[key] = m.cuda().eval()
    return _MODELS[key]


def clip_kat_case():
    """A vision transformer whose output can be computed BY HAND (plain numpy loops below, float64): 128 wide, 2 heads, 14x14 patches
    of a 224^2 image, 2 blocks with  q = k = 0 (uniform attention over the 257 tokens), v = out_proj = identity, c_proj = 0 (the MLP adds 0),
    patch embedding = (1 + c / 128) x the mean of colour channel c % 3 of the patch, class / positional embeddings that differ per
    position.  Pins -- independently of oracle/shims.py -- the class token sitting at position 0, the raster order of the patches and
    their pairing with the positional embedding, the head concatenation order, the number of keys (257, not the padded row count) and
    ln_pre / ln_1 / ln_post.  Returns (state_dict with keys 'visual.*', image (1,3,224,224) already preprocessed, expected (1,64))."""
    W, P, g, L, heads, layers = 128, 14, 16, 257, 2, 2
    gen = torch.Generator().manual_seed(123)
    img = torch.rand(1, 3, 224, 224, generator=gen) * 2.0 - 1.0
    sd = {}
    conv = torch.zeros(W, 3, P, P)
    for c in range(W):
        conv[c, c % 3] = (1.0 + c / 128.0) / (P * P)
    sd["visual.conv1.weight"] = conv
    sd["visual.class_embedding"] = torch.linspace(-0.5, 0.5, W)
    sd["visual.positional_embedding"] = 0.02 * torch.arange(L, dtype=torch.float32)[:, None] * torch.cos(torch.arange(W, dtype=torch.float32))[None, :] \
        + 0.3 * torch.sin(0.37 * torch.arange(W, dtype=torch.float32))[None, :]
    for n in ("ln_pre", "ln_post"):
        sd[f"visual.{n}.weight"], sd[f"visual.{n}.bias"] = torch.ones(W), torch.zeros(W)
    for i in range(layers):
        b = f"visual.transformer.resblocks.{i}."
        wqkv = torch.zeros(3 * W, W)
        wqkv[2 * W:] = torch.eye(W)
        sd[b + "attn.in_proj_weight"], sd[b + "attn.in_proj_bias"] = wqkv, torch.zeros(3 * W)
        sd[b + "attn.out_proj.weight"], sd[b + "attn.out_proj.bias"] = torch.eye(W), torch.zeros(W)
        sd[b + "ln_1.weight"], sd[b + "ln_1.bias"] = torch.ones(W) * (1.0 + 0.1 * i), torch.full((W,), 0.05 * i)
        sd[b + "ln_2.weight"], sd[b + "ln_2.bias"] = torch.ones(W), torch.zeros(W)
        sd[b + "mlp.c_fc.weight"] = torch.randn(4 * W, W, generator=gen) * 0.05
        sd[b + "mlp.c_fc.bias"] = torch.zeros(4 * W)
        sd[b + "mlp.c_proj.weight"], sd[b + "mlp.c_proj.bias"] = torch.zeros(W, 4 * W), torch.zeros(W)
    proj = torch.zeros(W, 64)
    for j in range(64):
        proj[2 * j, j] = 1.0
        proj[2 * j + 1, j] = -0.5
    sd["visual.proj"] = proj
    # ---- by hand
    x = img[0].double().numpy()
    tok = np.zeros((L, W))
    tok[0] = sd["visual.class_embedding"].double().numpy()
    for py in range(g):
        for px in range(g):
            means = [x[ch, py * P:(py + 1) * P, px * P:(px + 1) * P].sum() / (P * P) for ch in range(3)]
            for c in range(W):
                tok[1 + py * g + px, c] = (1.0 + c / 128.0) * means[c % 3]
    tok += sd["visual.positional_embedding"].double().numpy()

    def ln(v, w=1.0, b=0.0):
        mu = v.mean(axis=-1, keepdims=True)
        var = ((v - mu) ** 2).mean(axis=-1, keepdims=True)
        return (v - mu) / np.sqrt(var + 1e-5) * w + b

    h = ln(tok)
    for i in range(layers):
        y = ln(h, 1.0 + 0.1 * i, 0.05 * i)
        h = h + y.mean(axis=0, keepdims=True)          # uniform attention over all 257 tokens, v = out_proj = identity; the MLP adds 0
    cls = ln(h[0])
    expected = cls @ proj.double().numpy()
    return sd, img, torch.from_numpy(expected).float()[None]
