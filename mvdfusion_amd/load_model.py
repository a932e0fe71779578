"""Config-driven factory -- mirror of ``utils/load_model.py`` (utils/load_model.py:10-110).

The reference's yaml files name classes by dotted path (``mvdfusion.unet.UNetModel`` ...).  ``instantiate_from_config``
resolves those same strings to the MI355X-native mirrors, so ``configs/*.yaml`` work unchanged.
"""
import importlib
from collections import OrderedDict

import torch

# reference dotted path -> native module
TARGET_MAP = {
    "mvdfusion.viewfusion_zero_depth_rgb.ViewFusion": "mvdfusion_amd.viewfusion_zero_depth_rgb.ViewFusion",
    "mvdfusion.view_attn_efficient2.GridAttn": "mvdfusion_amd.view_attn_efficient2.GridAttn",
    "mvdfusion.unet.UNetModel": "mvdfusion_amd.unet.UNetModel",
    "mvdfusion.scheduler.DDPMScheduler": "mvdfusion_amd.scheduler.DDPMScheduler",
    "external.sd1.ldm.models.autoencoder.AutoencoderKL": "mvdfusion_amd.autoencoder.AutoencoderKL",   # decode side on HIP
}
IGNORED = {"__is_first_stage__", "__is_unconditional__", "dataset", "trainer", "saver"}

# zero123 checkpoint -> this UNet: block indices shift where a ViewAlignedFeatureTransformer was inserted
# (mvdfusion/unet.py:70-86)
_UNET_PARAM_MAP = {"output_blocks.5.2.conv.": "output_blocks.5.3.conv.", "output_blocks.8.2.conv.": "output_blocks.8.3.conv.",
                   "middle_block.2.": "middle_block.3."}


def get_obj_from_str(string):
    string = TARGET_MAP.get(string, string)
    module, cls = string.rsplit(".", 1)
    return getattr(importlib.import_module(module), cls)


def instantiate_from_config(config):
    if "target" not in config:
        if config in IGNORED:
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**dict(config.get("params", dict())))


def load_unet_checkpoint(model, ckpt, remove_keys=(), prefix="model.diffusion_model."):
    """Load an SD / zero123 ``state_dict`` into the view-conditioned UNet (strict=False; aligned_attn_* stay as
    initialised; input/output convs whose channel counts changed are dropped)."""
    sd = torch.load(ckpt, map_location="cpu")["state_dict"]
    out = OrderedDict()
    for k, v in sd.items():
        if not k.startswith(prefix):
            continue
        name = k[len(prefix):]
        for src, dst in _UNET_PARAM_MAP.items():
            if name.startswith(src):
                name = dst + name[len(src):]
                break
        if name in remove_keys:
            continue
        out[name] = v
    return model.load_state_dict(out, strict=False)


def load_model_from_config(config, ckpt=None, **kwargs):
    model = instantiate_from_config(config)
    if ckpt is not None:
        model.load_state_dict(torch.load(ckpt, map_location="cpu")["state_dict"], strict=False)
    model.eval()
    return model
