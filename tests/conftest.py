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


def pytest_collection_modifyitems(config, items):
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
    """split-bf16 plane pair (2, ...) int16 -> fp32 value hi + lo."""
    p = p.detach().cpu()
    f = lambda t: (t.to(torch.int32) << 16).view(torch.float32)
    return f(p[0]) + f(p[1])


def rmse(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float(((a - b) ** 2).mean().sqrt())


UNET_PARAMS = dict(image_size=32, in_channels=10, out_channels=5, model_channels=320, attention_resolutions=[4, 2, 1],
                   num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True,
                   use_view_aligned_transformer=True, transformer_depth=1, context_dim=768, use_checkpoint=True,
                   legacy=False)


def model_config(mc=320, D=1, S=32, precision="bf16x3"):
    """The `params:` block of configs/mvd_gso.yaml (model part) as a dict."""
    up = dict(UNET_PARAMS)
    up["model_channels"] = mc
    up["image_size"] = S
    return dict(
        view_attn_config=dict(target="mvdfusion.view_attn_efficient2.GridAttn",
                              params=dict(in_channels=5, input_size=S, output_dim=768, num_layers=3,
                                          z_near_far_scale=0.8, n_pts_per_ray=D)),
        unet_config=dict(target="mvdfusion.unet.UNetModel", params=up),
        ddpm_config=dict(target="mvdfusion.scheduler.DDPMScheduler", params=dict(timesteps=1000)),
        vae_path=None, unet_path=None, z_scale_factor=0.18215, objective="noise", loss_type="l2",
        embed_camera_pose=True, finetune_projection=True, finetune_unet=False, finetune_cross_attn=True,
        finteune_view_attn=True, drop_conditions=True, precision=precision)


_MODELS = {}


def build_model(mc=320, D=1, S=32, precision="bf16x3"):
    """ViewFusion on cuda:0 with the deterministic non-zero fill (cached per configuration)."""
    from mvdfusion_amd import synthetic as syn
    from mvdfusion_amd.viewfusion_zero_depth_rgb import ViewFusion
    key = (mc, D, S, precision)
    if key not in _MODELS:
        m = ViewFusion(**model_config(mc, D, S, precision))
        syn.fill_module_(m)
        _MODELS[key] = m.cuda().eval()
    return _MODELS[key]
