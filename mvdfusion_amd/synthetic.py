"""Synthetic inputs and deterministic weights for parity tests and bench.py (no datasets / checkpoints).

* :func:`det_fill` -- name-keyed, machine-independent, NON-ZERO parameter fill.  The reference zero-inits every
  ResBlock output conv, every transformer ``proj_out``, the UNet head and the GridAttn adaLN layers
  (openaimodel.py:229, attention.py:259, mvdfusion/attention.py:114, unet.py:499, view_attn_efficient2.py:174-176),
  so plain random init would make the UNet output exactly 0 and parity vacuous (SURVEY.md trap T1).
* :func:`gso_rig` -- the fixed 16-camera GSO evaluation rig (dataset/gso_test.py:48-56,134-149).
* :func:`make_inputs` -- everything ``DDIMSampler.sample`` consumes, shaped as ``ViewFusion.prepare_batch`` would
  return it (viewfusion_zero_depth_rgb.py:165-273), with a seeded stand-in for the VAE latents / CLIP vector.
"""
import math
import zlib

import numpy as np
import torch

from .cameras import Cameras, get_camera_slice, get_relative_camera, look_at_view_transform

_NORM_TOKENS = (".norm.", ".norm1.", ".norm2.", ".norm3.", "in_layers.0.", "out_layers.0.", "out.0.",
                "aligned_attn_norm.", ".norm_out.", ".ln_1.", ".ln_2.", ".ln_pre.", ".ln_post.", ".ln_final.")
_RESIDUAL_OUT_TOKENS = ("out_layers.3.", ".proj_out.", "aligned_attn_proj_out.", "to_out.0.", "ff.net.2.",
                        "attn.proj.", "mlp.fc2.")


def det_fill(name, shape, dtype=torch.float32):
    """Deterministic fill keyed by the parameter NAME (crc32 -> numpy RandomState, frozen stream)."""
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    n = int(np.prod(shape)) if len(shape) else 1
    v = rs.standard_normal(n).astype(np.float32).reshape(shape)
    is_norm = any(tok in name for tok in _NORM_TOKENS)
    if name.endswith("weight") and len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        gain = 1.0
        if any(tok in name for tok in _RESIDUAL_OUT_TOKENS):
            gain = 0.5
        if "adaLN_modulation" in name:
            gain = 0.5
        if name.endswith("out.2.weight"):
            gain = 1.0
        v *= gain / math.sqrt(fan_in)
    elif name.endswith("weight") and is_norm:
        v = 1.0 + 0.1 * v
    elif name.endswith("bias") and is_norm:
        v = 0.05 * v
    elif name.endswith("bias"):
        v = 0.02 * v
    else:
        v = 0.02 * v
    return torch.from_numpy(np.ascontiguousarray(v)).to(dtype).reshape(shape)


def det_fill_state_dict(spec):
    """spec: iterable of (name, shape) -> {name: tensor}."""
    return {k: det_fill(k, tuple(s)) for k, s in spec}


def fill_module_(module, prefix=""):
    """Overwrite every parameter of an nn.Module with det_fill(prefix + name)."""
    with torch.no_grad():
        for k, p in module.named_parameters():
            p.copy_(det_fill(prefix + k, tuple(p.shape)).to(p.device, p.dtype))
    return module


def synthetic_zero123_ckpt(unet_sd, in_ch=8, out_ch=4):
    """A zero123 / SD-shaped checkpoint for the view-conditioned UNet whose (reference-named) state_dict is `unet_sd`: no
    aligned_attn_* keys, the blocks that follow an inserted ViewAlignedFeatureTransformer at their ORIGINAL indices
    (middle_block.3 -> .2, output_blocks.{5,8}.3.conv -> .2.conv), 8-channel stem / 4-channel head, `model.diffusion_model.`
    prefix, plus first-stage / cond-stage keys that must be ignored.  Values: det_fill keyed by the checkpoint key."""
    ck = {}
    for k, v in unet_sd.items():
        if "aligned_attn_" in k:
            continue
        name = k
        for dst, src in (("middle_block.3.", "middle_block.2."), ("output_blocks.5.3.conv.", "output_blocks.5.2.conv."),
                         ("output_blocks.8.3.conv.", "output_blocks.8.2.conv.")):
            if name.startswith(dst):
                name = src + name[len(dst):]
        shape = list(v.shape)
        if name == "input_blocks.0.0.weight":
            shape[1] = in_ch
        if name in ("out.2.weight", "out.2.bias"):
            shape[0] = out_ch
        ck["model.diffusion_model." + name] = det_fill("zero123." + name, tuple(shape))
    ck["first_stage_model.encoder.conv_in.weight"] = torch.zeros(4, 3, 3, 3)
    ck["cond_stage_model.model.visual.proj"] = torch.zeros(8, 8)
    return {"state_dict": ck, "global_step": 1}



# ---------------------------------------------------------------------------------------------
GSO_AZIMUTHS = [0.0, 0.39269909262657166, 0.7853981852531433, 1.1780972480773926, 1.5707963705062866,
                1.9634953737258911, 2.356194496154785, 2.7488934993743896, 3.1415927410125732, 3.5342917442321777,
                3.9269907474517822, 4.319689750671387, 4.71238899230957, 5.105088233947754, 5.497786998748779,
                5.890486240386963]
GSO_ELEVATION = 0.5235987901687622


def gso_rig():
    """16 views, azimuth k*22.5deg (+90), elevation 30deg, distance 1.5, focal 2.1875, principal (0,0)."""
    az = torch.tensor(GSO_AZIMUTHS, dtype=torch.float32)
    el = torch.full((16,), GSO_ELEVATION, dtype=torch.float32)
    R, T = look_at_view_transform(1.5, el * 180 / torch.pi, az * 180 / torch.pi + 90)
    f = torch.full((16, 2), 2.1875)
    p = torch.zeros(16, 2)
    return Cameras(R, T, f, p)


def select_views(num_total, V):
    """random_views:false => linspace(0, B-1, 1+V).long(); first is the input view (viewfusion...:198-200)."""
    idx = torch.linspace(0, num_total - 1, 1 + V).long()
    return idx[:1], idx[1:]


def cam_embed(input_cam, batch_cams):
    """The 28 camera scalars appended to the CLIP vector (viewfusion_zero_depth_rgb.py:247-258)."""
    V = len(batch_cams)
    i = torch.cat([input_cam.R.reshape(1, 9), input_cam.T, input_cam.focal_length], dim=-1).expand(V, -1)
    b = torch.cat([batch_cams.R.reshape(V, 9), batch_cams.T, batch_cams.focal_length], dim=-1)
    return torch.cat([i, b], dim=-1)[:, None, :]      # (V,1,28)


def make_inputs(V, S=32, seed=0):
    """Synthetic stand-in for ``prepare_batch`` output + the initial noise x_T.

    Returns dict(batch_cameras, input_cameras, input_latents (1,5,S,S), clip_v_embed (V,1,796), x_T (V,5,S,S)).
    """
    g = torch.Generator().manual_seed(1234 + seed)
    rig = get_relative_camera(gso_rig(), [0])
    in_idx, b_idx = select_views(16, V)
    input_cam = get_camera_slice(rig, in_idx)
    batch_cams = get_camera_slice(rig, b_idx)
    lat = torch.randn(1, 4, S, S, generator=g) * 0.18215 * 4.0     # VAE latents*0.18215 have std ~0.7
    input_latents = torch.cat([lat, torch.zeros(1, 1, S, S)], dim=1)  # depth forced to 0 (:215)
    clip = torch.randn(1, 1, 768, generator=g).expand(V, -1, -1)
    clip_v_embed = torch.cat([clip, cam_embed(input_cam, batch_cams)], dim=-1).contiguous()
    x_T = torch.randn(V, 5, S, S, generator=torch.Generator().manual_seed(seed))
    return {"batch_cameras": batch_cams, "input_cameras": input_cam, "input_latents": input_latents,
            "clip_v_embed": clip_v_embed, "x_T": x_T}


def step_noise(V, S, D, num_steps, seed=0):
    """Host noise in the reference's draw order (SURVEY.md trap T2): per step, first the depth-sample
    noise (V,D,S,S) (view_attn_efficient2.py:431), then the DDIM noise (V,5,S,S) (sampler.py:64, not drawn
    at the last step)."""
    g = torch.Generator().manual_seed(99991 + seed)
    depth, ddim = [], []
    for i in range(num_steps):
        depth.append(torch.randn(V, D, S, S, generator=g))
        ddim.append(torch.randn(V, 5, S, S, generator=g) if i < num_steps - 1 else torch.zeros(V, 5, S, S))
    return torch.stack(depth), torch.stack(ddim)


class StubClipImageEncoder(torch.nn.Module):
    """Deterministic stand-in for FrozenCLIPImageEmbedder (encoders/modules.py:402-441) in prepare_batch parity tests: the
    per-channel image mean through a fixed name-keyed 3x768 map -> (n, 1, 768).  CLIP weights are not available offline and
    CLIP is not on the HIP path; the stub only has to be the SAME function on the reference side and on this side."""

    def __init__(self):
        super().__init__()
        self.register_buffer("proj", det_fill("stub_clip.proj.weight", (3, 768)) * math.sqrt(3.0), persistent=False)

    def encode(self, x):
        return (x.float().mean((2, 3)) @ self.proj.to(x.device))[:, None, :]
