"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference through
oracle/shims.py) -- TEST INFRASTRUCTURE, container-only (the reference does not exist on the GPU box).

For every fixture the same inputs are also pushed through the restatement in oracle/ref_torch.py and the
two are required to agree (this is what pins the oracle).  Only inputs that cannot be re-derived from seeds
and the outputs are stored; weights come from mvdfusion_amd.synthetic.det_fill (name-keyed, no weight files).

Run:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden [--only NAME]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import shims  # noqa: E402

shims.install()

from oracle import ref_torch as O  # noqa: E402
from mvdfusion_amd import synthetic as syn  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
UNET_PARAMS = dict(image_size=32, in_channels=10, out_channels=5, model_channels=320,
                   attention_resolutions=[4, 2, 1], num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8,
                   use_spatial_transformer=True, use_view_aligned_transformer=True, transformer_depth=1,
                   context_dim=768, use_checkpoint=True, legacy=False)


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def fill_ref(module, prefix=""):
    return syn.fill_module_(module, prefix)


def sd_of(module, prefix=""):
    return {prefix + k: v.detach().clone() for k, v in module.state_dict().items()}


def ref_cams(c):
    from pytorch3d.renderer import PerspectiveCameras
    return PerspectiveCameras(R=c.R, T=c.T, focal_length=c.focal_length, principal_point=c.principal_point)


def cam_dict(c):
    return {"R": c.R, "T": c.T, "f": c.focal_length, "p": c.principal_point}


def save(name, **arrs):
    os.makedirs(GOLD, exist_ok=True)
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    kb = os.path.getsize(os.path.join(GOLD, name + ".npz")) / 1024
    print(f"  wrote tests/golden/{name}.npz ({kb:.0f} KB)")


# ---------------------------------------------------------------------------------------------
def gold_schedule():
    from mvdfusion.scheduler import DDPMScheduler
    from mvdfusion.sampler import DDIMSampler

    class M:
        pass
    m = M()
    m.scheduler = DDPMScheduler(1000)
    s = DDIMSampler(m, ddim_num_steps=50, ddim_discretize="uniform", ddim_eta=1.0, latent_size=32, z_dim=4)
    tab = O.ddpm_tables()
    dd = O.ddim_schedule(tab)
    assert torch.equal(tab["alphas_cumprod"], m.scheduler.alphas_cumprod)
    assert torch.equal(tab["sqrt_alphas_cumprod"], m.scheduler.sqrt_alphas_cumprod)
    assert torch.equal(tab["sqrt_one_minus_alphas_cumprod"], m.scheduler.sqrt_one_minus_alphas_cumprod)
    assert np.array_equal(dd["timesteps"].numpy(), s.ddim_timesteps)
    for a, b in (("alphas", s.ddim_alphas), ("alphas_prev", s.ddim_alphas_prev), ("sigmas", s.ddim_sigmas),
                 ("sqrt_one_minus_alphas", s.ddim_sqrt_one_minus_alphas)):
        assert torch.equal(dd[a], b), a
    # DDIM update with injected noise
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 5, 8, 8, generator=g)
    eps = torch.randn(2, 5, 8, 8, generator=g)
    outs = {}
    for index in (49, 25, 1, 0):
        torch.manual_seed(77)
        xp, x0 = s.denoise_apply_impl(x, index, eps, is_step0=index == 0)
        torch.manual_seed(77)
        noise = torch.randn_like(x) if index > 0 else None
        oxp, ox0 = O.ddim_update(x, eps, dd, index, noise)
        assert torch.equal(oxp, xp) and torch.equal(ox0, x0), index
        outs[f"x_prev_{index}"] = xp
        outs[f"x0_{index}"] = x0
    save("schedule", alphas_cumprod=tab["alphas_cumprod"], ddim_timesteps=dd["timesteps"], ddim_alphas=dd["alphas"],
         ddim_alphas_prev=dd["alphas_prev"], ddim_sigmas=dd["sigmas"],
         ddim_sqrt_one_minus_alphas=dd["sqrt_one_minus_alphas"], x=x, eps=eps,
         noise=torch.randn(2, 5, 8, 8, generator=torch.Generator().manual_seed(77)), **outs)
    print("  schedule: oracle == reference (bit-exact)")


def gold_cameras():
    """Known-answer fixtures for the camera algebra (rig, re-basing, projection round trip)."""
    from pytorch3d.renderer import look_at_view_transform, PerspectiveCameras
    from utils.camera_utils import _get_relative_camera, _get_camera_slice
    az = torch.tensor(syn.GSO_AZIMUTHS)
    el = torch.full((16,), syn.GSO_ELEVATION)
    R, T = look_at_view_transform(dist=1.5, azim=az * 180 / torch.pi + 90, elev=el * 180 / torch.pi,
                                  up=((0, 1, 0),))
    cams = PerspectiveCameras(R=R, T=T, focal_length=((2.1875, 2.1875),), principal_point=((0, 0),))
    rel = _get_relative_camera(cams, query_idx=torch.tensor([0]))
    mine = syn.gso_rig()
    from mvdfusion_amd.cameras import get_relative_camera
    mrel = get_relative_camera(mine, [0])
    assert rel_err(mine.R, R) < 1e-6 and rel_err(mine.T, T) < 1e-6
    assert rel_err(mrel.R, rel.R) < 1e-6 and rel_err(mrel.T, rel.T) < 1e-6
    g = torch.Generator().manual_seed(3)
    pts = torch.randn(1, 64, 3, generator=g) * 0.3
    ndc = rel.transform_points_ndc(pts)
    o_ndc = O.project_ndc(rel.R, rel.T, rel.focal_length, rel.principal_point, pts[0])
    assert rel_err(o_ndc, ndc) < 1e-5
    xy_d = torch.cat([torch.rand(16, 64, 2, generator=g) * 2 - 1, torch.rand(16, 64, 1, generator=g) * 2 + 0.5], -1)
    unp = rel.unproject_points(xy_d, from_ndc=True)
    o_unp = O.unproject_ndc(rel.R, rel.T, rel.focal_length, rel.principal_point, xy_d[..., :2], xy_d[..., 2])
    assert rel_err(o_unp, unp) < 1e-5
    assert rel_err(O.camera_center(rel.R, rel.T), rel.get_camera_center()) < 1e-6
    save("cameras", R=R, T=T, rel_R=rel.R, rel_T=rel.T, pts=pts[0], ndc=ndc, xy_d=xy_d, unproj=unp,
         centers=rel.get_camera_center())
    print("  cameras: rig / re-basing / project / unproject agree with the reference call sites")


def _gridattn_case(V, D, S, seed, t_val, tokens=True, tol=2e-5, prev_depth=False):
    from mvdfusion.view_attn_efficient2 import GridAttn
    from mvdfusion.scheduler import DDPMScheduler
    ga = GridAttn(in_channels=5, input_size=S, output_dim=768, num_layers=3, z_near_far_scale=0.8, n_pts_per_ray=D)
    fill_ref(ga, "view_attn.")
    ga.eval()
    sd = sd_of(ga, "view_attn.")
    inp = syn.make_inputs(V, S, seed)
    g = torch.Generator().manual_seed(100 + seed)
    x = torch.randn(V, 5, S, S, generator=g)
    t_embed = torch.randn(V, 256, generator=g) * 0.5
    t = torch.full((V,), t_val, dtype=torch.long)
    sched = DDPMScheduler(1000)
    prev = (torch.randn(V, 1, S, S, generator=g) * 0.4) if prev_depth else None       # overwrite_attn_depth (:418-426)
    torch.manual_seed(4242 + seed)
    with torch.no_grad():
        ref = ga(x, ref_cams(inp["batch_cameras"]), torch.ones(V), t_embed, t, sched, overwrite_attn_depth=prev,
                 input_latents=inp["input_latents"], input_cameras=ref_cams(inp["input_cameras"]))
    torch.manual_seed(4242 + seed)
    depth_noise = torch.randn(V, D, S, S)
    tab = O.ddpm_tables()
    with torch.no_grad():
        mine = O.gridattn_forward(sd, "view_attn.", x, cam_dict(inp["batch_cameras"]), t_embed, t, tab, depth_noise,
                                  inp["input_latents"], cam_dict(inp["input_cameras"]), n_pts_per_ray=D, overwrite_attn_depth=prev)
        tokens = O.gridattn_forward(sd, "view_attn.", x, cam_dict(inp["batch_cameras"]), t_embed, t, tab, depth_noise,
                                    inp["input_latents"], cam_dict(inp["input_cameras"]), n_pts_per_ray=D,
                                    return_tokens=True) if tokens else torch.zeros(97, 1, 1)
    e = rel_err(mine, ref)
    print(f"  gridattn V={V} D={D} S={S} t={t_val}: oracle vs reference rel-max err {e:.2e}")
    assert e < tol, e
    extra = dict(prev_depth=prev) if prev_depth else {}
    return dict(x=x, t_embed=t_embed, t=t, depth_noise=depth_noise, out=ref, seed=np.int64(seed), **extra,
                tokens_sample=tokens[::97][:, :, :].contiguous(), tokens_stride=np.int64(97))


def gold_gridattn():
    c = _gridattn_case(4, 1, 32, 0, 981)
    # keep the fixture small: fp16-free, but subsample the (V,S,S,D,768) output on a strided lattice + full stats
    out = c.pop("out")
    save("gridattn_v4_d1", out_full_view0=out[0, ::2, ::2], out_strided=out[:, ::5, ::7, :, ::3],
         out_mean=out.mean(), out_std=out.std(), out_l2=out.norm(), **c)
    c = _gridattn_case(3, 3, 32, 1, 501)
    out = c.pop("out")
    save("gridattn_v3_d3", out_strided=out[:, ::5, ::7, :, ::3], out_mean=out.mean(), out_std=out.std(),
         out_l2=out.norm(), **c)
    c = _gridattn_case(8, 1, 32, 2, 21)
    out = c.pop("out")
    c.pop("tokens_sample")
    save("gridattn_v8_d1", out_strided=out[:, ::5, ::7, :, ::3], out_mean=out.mean(), out_std=out.std(),
         out_l2=out.norm(), **c)


def _unet(model_channels, S=32):
    from mvdfusion.unet import UNetModel
    p = dict(UNET_PARAMS)
    p["model_channels"] = model_channels
    p["image_size"] = S            # keys ViewAlignedFeatureTransformer.level_mapper (mvdfusion/attention.py:117)
    net = UNetModel(**p)
    fill_ref(net, "unet_model.unet_model.")
    net.eval()
    return net


def _unet_inputs(V, D, S, seed):
    g = torch.Generator().manual_seed(200 + seed)
    x = torch.randn(V, 10, S, S, generator=g)
    ctx = torch.randn(V, 1, 768, generator=g)
    vol = torch.randn(V, S, S, D, 768, generator=g) * 0.5
    return x, ctx, vol


def gold_unet(model_channels, V, D, tag, t_val=981, S=32, full=True):
    from mvdfusion.unet import UNetWrapper
    net = _unet(model_channels, S)
    sd = sd_of(net, "unet_model.unet_model.")
    x, ctx, vol = _unet_inputs(V, D, S, model_channels + V + D)
    t1 = torch.tensor([t_val], dtype=torch.long)
    levels = UNetWrapper.get_volume_feats_pyramid(type("W", (), {"unet_model": net})(), vol)
    t0 = time.time()
    with torch.no_grad():
        ref = net(x, t1, ctx, volume_feats=levels)
    dt = time.time() - t0
    with torch.no_grad():
        mine = O.unet_forward(sd, "unet_model.unet_model.", x, t1, ctx, O.volume_pyramid(vol),
                              model_channels=model_channels, image_size=S)
    e = rel_err(mine, ref)
    print(f"  unet mc={model_channels} V={V} D={D}: ref {dt:.1f}s, oracle vs reference rel-max err {e:.2e}, "
          f"out std {float(ref.std()):.3f}")
    assert e < 2e-5, e
    if full:
        save(tag, x=x, ctx=ctx, vol_seed=np.int64(model_channels + V + D), t=t1, out=ref,
             spec=json.dumps([[k, list(v.shape)] for k, v in net.state_dict().items()]))
    else:      # large latents: inputs re-derive from the seed, keep view 0 + a strided lattice + summaries of the output
        save(tag, vol_seed=np.int64(model_channels + V + D), t=t1, out_view0=ref[0], out_strided=ref[:, :, ::3, ::5].contiguous(),
             out_std=ref.std(), out_l2=ref.norm())
    return net


def gold_step(model_channels, V, D, tag, indices=(49, 1, 0), S=32, lean=False):
    """One DDIMSampler.denoise_apply through a ViewFusion-shaped stand-in (no VAE / CLIP: not on the path)."""
    import torch.nn as nn
    from mvdfusion.view_attn_efficient2 import GridAttn
    from mvdfusion.scheduler import DDPMScheduler
    from mvdfusion.sampler import DDIMSampler
    from mvdfusion.unet import UNetWrapper
    from mvdfusion.viewfusion_zero_depth_rgb import ViewFusion

    class Facade(nn.Module):
        """The exact ViewFusion members that apply_model touches (viewfusion_zero_depth_rgb.py:282-345)."""
        embed_time = ViewFusion.embed_time
        apply_model = ViewFusion.apply_model

        def __init__(self):
            super().__init__()
            self.view_attn = GridAttn(in_channels=5, input_size=S, output_dim=768, num_layers=3,
                                      z_near_far_scale=0.8, n_pts_per_ray=D)
            w = UNetWrapper.__new__(UNetWrapper)
            nn.Module.__init__(w)
            w.unet_model = _unet(model_channels, S)
            w.drop_conditions, w.use_zero_123 = False, True
            self.unet_model = w
            self.scheduler = DDPMScheduler(1000)
            self.cc_projection = nn.Sequential(nn.Linear(796, 768), nn.SiLU(True), nn.Linear(768, 768),
                                               nn.SiLU(True), nn.Linear(768, 768))
            self.time_embed_dim = 256
            self.time_embed = nn.Sequential(nn.Linear(256, 256), nn.SiLU(True), nn.Linear(256, 256))
            self.register_buffer("_device", torch.tensor([0.0]), persistent=False)

    m = Facade()
    for name in ("view_attn", "cc_projection", "time_embed"):
        fill_ref(getattr(m, name), name + ".")
    m.eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    sampler = DDIMSampler(m, ddim_num_steps=50, ddim_discretize="uniform", ddim_eta=1.0, latent_size=S, z_dim=4)
    inp = syn.make_inputs(V, S, seed=7)
    tab, dd = O.ddpm_tables(), O.ddim_schedule(O.ddpm_tables())
    bc, ic = ref_cams(inp["batch_cameras"]), ref_cams(inp["input_cameras"])
    res = {}
    x = inp["x_T"]
    for index in indices:
        tval = int(dd["timesteps"][index])
        ts = torch.full((V,), tval, dtype=torch.long)
        torch.manual_seed(900 + index)
        t0 = time.time()
        with torch.no_grad():
            xp, x0 = sampler.denoise_apply(x, bc, inp["input_latents"], ic, inp["clip_v_embed"], ts, index,
                                           is_step0=index == 0, cfg_scale=2.5)
        dt = time.time() - t0
        torch.manual_seed(900 + index)
        dn = torch.randn(V, D, S, S)
        sn = torch.randn(V, 5, S, S) if index > 0 else None
        with torch.no_grad():
            oxp, ox0 = O.denoise_step(sd, x, cam_dict(inp["batch_cameras"]), inp["input_latents"],
                                      cam_dict(inp["input_cameras"]), inp["clip_v_embed"], tab, dd, index, dn, sn,
                                      cfg_scale=2.5, n_pts_per_ray=D,
                                      unet_kw=dict(model_channels=model_channels, image_size=S))
        e = max(rel_err(oxp, xp), rel_err(ox0, x0))
        print(f"  step mc={model_channels} V={V} D={D} index={index}: ref {dt:.1f}s, oracle vs reference {e:.2e}")
        assert e < 5e-5, e
        if not lean:      # lean fixtures: the test re-draws the noise from torch.manual_seed(900 + index) like this script
            res[f"depth_noise_{index}"] = dn
            res[f"step_noise_{index}"] = sn if sn is not None else torch.zeros(V, 5, S, S)
        res[f"x_prev_{index}"] = xp
        res[f"x0_{index}"] = x0
    if lean:
        save(tag, indices=np.asarray(indices), noise_seed_base=np.int64(900), **res)      # x = make_inputs(V, S, seed=7)["x_T"]
    else:
        save(tag, x=x, indices=np.asarray(indices), **res)


def gold_gridattn_v15():
    """The largest view count the reference allows on the GSO rig (1 input + 15 targets): exercises the generic
    (non-templated) cross-view attention path."""
    c = _gridattn_case(15, 1, 32, 3, 741)
    out = c.pop("out")
    c.pop("tokens_sample")
    save("gridattn_v15_d1", out_strided=out[:, ::5, ::7, :, ::3], out_mean=out.mean(), out_std=out.std(),
         out_l2=out.norm(), **c)


def gold_gridattn_small(V, D, seed, t_val, tag):
    """View counts the reference's configs ship that are not powers of two (mvd_train.yaml: 5 and 7 views): the fused kernel pads them."""
    c = _gridattn_case(V, D, 32, seed, t_val, tokens=False)
    out = c.pop("out")
    c.pop("tokens_sample")
    save(tag, out_strided=out[:, ::5, ::7, :, ::3], out_mean=out.mean(), out_std=out.std(), out_l2=out.norm(), **c)


def gold_gridattn_prev_depth():
    """overwrite_attn_depth (view_attn_efficient2.py:418-426): the depth channel comes from the caller instead of the x0-estimate."""
    c = _gridattn_case(4, 1, 32, 6, 301, tokens=False, prev_depth=True)
    out = c.pop("out")
    c.pop("tokens_sample")
    save("gridattn_v4_d1_prevdepth", out_strided=out[:, ::5, ::7, :, ::3], out_mean=out.mean(), out_std=out.std(), out_l2=out.norm(), **c)


def gold_sample_feed_prev_depth(model_channels=32, V=2, S=32, steps=3):
    """The reference's DDIMSampler.sample loop with feed_prev_depth=True (sampler.py:83-84,119-142): from the second iteration on
    GridAttn samples depth around the PREVIOUS step's x0-estimate.  The loop is truncated to the first `steps` iterations by patching
    the sampler's timestep list (the per-step arithmetic is untouched); torch's global generator supplies the reference's own draws,
    replayed for the oracle in the same order."""
    import torch.nn as nn
    from mvdfusion.view_attn_efficient2 import GridAttn
    from mvdfusion.scheduler import DDPMScheduler
    from mvdfusion.sampler import DDIMSampler
    from mvdfusion.unet import UNetWrapper
    from mvdfusion.viewfusion_zero_depth_rgb import ViewFusion

    class Facade(nn.Module):
        embed_time = ViewFusion.embed_time
        apply_model = ViewFusion.apply_model

        def __init__(self):
            super().__init__()
            self.view_attn = GridAttn(in_channels=5, input_size=S, output_dim=768, num_layers=3, z_near_far_scale=0.8, n_pts_per_ray=1)
            w = UNetWrapper.__new__(UNetWrapper)
            nn.Module.__init__(w)
            w.unet_model = _unet(model_channels, S)
            w.drop_conditions, w.use_zero_123 = False, True
            self.unet_model = w
            self.scheduler = DDPMScheduler(1000)
            self.cc_projection = nn.Sequential(nn.Linear(796, 768), nn.SiLU(True), nn.Linear(768, 768), nn.SiLU(True), nn.Linear(768, 768))
            self.time_embed_dim = 256
            self.time_embed = nn.Sequential(nn.Linear(256, 256), nn.SiLU(True), nn.Linear(256, 256))
            self.register_buffer("_device", torch.tensor([0.0]), persistent=False)

    m = Facade()
    for name in ("view_attn", "cc_projection", "time_embed"):
        fill_ref(getattr(m, name), name + ".")
    m.eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    sampler = DDIMSampler(m, ddim_num_steps=50, ddim_discretize="uniform", ddim_eta=1.0, latent_size=S, z_dim=4, feed_prev_depth=True)
    inp = syn.make_inputs(V, S, seed=9)
    bc, ic = ref_cams(inp["batch_cameras"]), ref_cams(inp["input_cameras"])
    tab, dd = O.ddpm_tables(), O.ddim_schedule(O.ddpm_tables())
    # the reference loop: x_T = randn, then per step [depth noise in GridAttn, update noise] -- drive it by hand over the first `steps`
    torch.manual_seed(777)
    x = torch.randn(V, 5, S, S)
    x_T = x.clone()
    prev, xs, x0s = None, [], []
    for i in range(steps):
        index = 49 - i
        ts = torch.full((V,), int(dd["timesteps"][index]), dtype=torch.long)
        with torch.no_grad():
            x, x0 = sampler.denoise_apply(x, bc, inp["input_latents"], ic, inp["clip_v_embed"], ts, index, is_step0=False,
                                          prev_depth=prev, cfg_scale=2.5)
        prev = x0[:, 4:].clone()
        xs.append(x)
        x0s.append(x0)
    torch.manual_seed(777)
    xo = torch.randn(V, 5, S, S)
    prev, dns, sns = None, [], []
    for i in range(steps):
        dn, sn = torch.randn(V, 1, S, S), torch.randn(V, 5, S, S)
        dns.append(dn)
        sns.append(sn)
        with torch.no_grad():
            xo, x0o = O.denoise_step(sd, xo, cam_dict(inp["batch_cameras"]), inp["input_latents"], cam_dict(inp["input_cameras"]),
                                     inp["clip_v_embed"], tab, dd, 49 - i, dn, sn, cfg_scale=2.5,
                                     unet_kw=dict(model_channels=model_channels, image_size=S), prev_depth=prev)
        prev = x0o[:, 4:].clone()
        e = max(rel_err(xo, xs[i]), rel_err(x0o, x0s[i]))
        print(f"  feed_prev_depth step {i}: oracle vs reference {e:.2e}")
        assert e < 5e-5, e
    save("sample_prevdepth_mc32_v2", x_T=x_T, depth_noise=torch.stack(dns), step_noise=torch.stack(sns), xs=torch.stack(xs), x0s=torch.stack(x0s))


def gold_gridattn_v8_s64():
    """BASELINE configs[3]'s cross-view problem: V = 8 views x 64x64 latents (T = 262 144 tokens, 0.94 TFLOP): strided lattice +
    norms of the (8, 64, 64, 1, 768) feature frustum."""
    c = _gridattn_case(8, 1, 64, 5, 401, tokens=False, tol=5e-5)      # (fp32 restatement vs reference at 262 144 tokens: 3.3e-5)
    out = c.pop("out")
    c.pop("tokens_sample")
    save("gridattn_v8_d1_s64", out_strided=out[:, ::5, ::7, :, ::3], out_mean=out.mean(), out_std=out.std(), out_l2=out.norm(), **c)


def gold_trajectory(model_channels, V, D, tag, steps=5, S=32):
    """A short stochastic DDIM trajectory (first `steps` iterations of sampler.sample's loop, sampler.py:119-142)."""
    sd, _ = synth_state_dict(model_channels, D, S)
    inp = syn.make_inputs(V, S, seed=11)
    tab, dd = O.ddpm_tables(), O.ddim_schedule(O.ddpm_tables())
    dn, sn = syn.step_noise(V, S, D, 50, seed=11)
    x = inp["x_T"]
    xs = []
    for i in range(steps):
        index = 50 - i - 1
        with torch.no_grad():
            x, x0 = O.denoise_step(sd, x, cam_dict(inp["batch_cameras"]), inp["input_latents"],
                                   cam_dict(inp["input_cameras"]), inp["clip_v_embed"], tab, dd, index, dn[i], sn[i],
                                   cfg_scale=2.5, n_pts_per_ray=D, unet_kw=dict(model_channels=model_channels))
        xs.append(x)
    keep = list(range(steps)) if steps <= 8 else [i for i in range(steps) if i % 5 == 4 or i == 0]
    save(tag, xs=torch.stack([xs[i] for i in keep]), kept=np.asarray(keep))


def gold_trajectory_f64(model_channels, V, D, tag, steps=50, S=32):
    """The same trajectory evaluated in float64 (weights / inputs / noise identical, arithmetic in double): the
    roundoff-free value of the reference algorithm.  Random-weight networks amplify fp32 roundoff by ~10^3 over the
    50 stochastic steps, so this -- not one particular fp32 evaluation order -- is what a 1e-3 tolerance is measured
    against; the fp32 oracle's own deviation from it is stored alongside (`fp32_oracle_rmse`)."""
    sd, _ = synth_state_dict(model_channels, D, S)
    inp = syn.make_inputs(V, S, seed=11)
    dn, sn = syn.step_noise(V, S, D, 50, seed=11)
    ref32 = load_npz(tag.replace("_f64", ""))
    torch.set_default_dtype(torch.float64)
    try:
        sd64 = {k: v.double() for k, v in sd.items()}
        tab = {k: v.double() for k, v in O.ddpm_tables().items()}
        dd = O.ddim_schedule(O.ddpm_tables())
        dd = {k: (v.double() if v.is_floating_point() else v) for k, v in dd.items()}
        c64 = lambda c: {k: v.double() for k, v in cam_dict(c).items()}
        x = inp["x_T"].double()
        xs = []
        t0 = time.time()
        for i in range(steps):
            with torch.no_grad():
                x, _ = O.denoise_step(sd64, x, c64(inp["batch_cameras"]), inp["input_latents"].double(),
                                      c64(inp["input_cameras"]), inp["clip_v_embed"].double(), tab, dd, 50 - i - 1,
                                      dn[i].double(), sn[i].double(), cfg_scale=2.5, n_pts_per_ray=D,
                                      unet_kw=dict(model_channels=model_channels))
            xs.append(x)
            if i % 10 == 0:
                print(f"    f64 step {i} ({time.time() - t0:.0f}s)", flush=True)
    finally:
        torch.set_default_dtype(torch.float32)
    keep = [int(k) for k in ref32["kept"]]
    dev = [float(((torch.from_numpy(ref32["xs"][j]).double() - xs[k]) ** 2).mean().sqrt()) for j, k in enumerate(keep)]
    print("  fp32 oracle vs float64:", ["%.2e" % d for d in dev])
    save(tag, xs=torch.stack([xs[k] for k in keep]).float(), kept=np.asarray(keep), fp32_oracle_rmse=np.asarray(dev))


def load_npz(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def synth_state_dict(model_channels, D, S=32):
    """The flat state_dict of the hot-path parameters from shapes alone (spec fixture) + det_fill."""
    spec = json.load(open(os.path.join(GOLD, f"state_dict_spec_mc{model_channels}.json")))
    return syn.det_fill_state_dict(spec), spec


def gold_spec(model_channels):
    import torch.nn as nn
    from mvdfusion.view_attn_efficient2 import GridAttn
    net = _unet(model_channels)
    ga = GridAttn(in_channels=5, input_size=32, output_dim=768, num_layers=3, z_near_far_scale=0.8, n_pts_per_ray=1)
    spec = [["view_attn." + k, list(v.shape)] for k, v in ga.state_dict().items()]
    spec += [["unet_model.unet_model." + k, list(v.shape)] for k, v in net.state_dict().items()]
    cc = nn.Sequential(nn.Linear(796, 768), nn.SiLU(True), nn.Linear(768, 768), nn.SiLU(True), nn.Linear(768, 768))
    spec += [["cc_projection." + k, list(v.shape)] for k, v in cc.state_dict().items()]
    te = nn.Sequential(nn.Linear(256, 256), nn.SiLU(True), nn.Linear(256, 256))
    spec += [["time_embed." + k, list(v.shape)] for k, v in te.state_dict().items()]
    os.makedirs(GOLD, exist_ok=True)
    with open(os.path.join(GOLD, f"state_dict_spec_mc{model_channels}.json"), "w") as f:
        json.dump(spec, f)
    print(f"  wrote state_dict_spec_mc{model_channels}.json ({len(spec)} keys, "
          f"{sum(int(np.prod(s)) for _, s in spec) / 1e6:.1f} M params)")



VAE_DDCONFIG = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
                    num_res_blocks=2, attn_resolutions=[], dropout=0.0)          # configs/mvd_gso.yaml:53-74


def gold_vae_decode(ch, B, zs, tag, full=True):
    """The REAL AutoencoderKL.decode (+ ViewFusion.decode's unnormalize/clip) on seeded latents; decoder weights from the
    name-keyed fill.  The decoder is fully convolutional, so small latent sides exercise the identical code."""
    from external.sd1.ldm.models.autoencoder import AutoencoderKL
    from utils.common_utils import unnormalize
    dd = dict(VAE_DDCONFIG)
    dd["ch"] = ch
    vae = AutoencoderKL(ddconfig=dd, lossconfig=dict(target="torch.nn.Identity"), embed_dim=4)
    fill_ref(vae, "vae.")
    vae.eval()
    sd = {k: v for k, v in sd_of(vae, "vae.").items() if ".encoder." not in k and not k.startswith("vae.quant_conv")}
    g = torch.Generator().manual_seed(700 + ch + zs)
    z = torch.randn(B, 4, zs, zs, generator=g) * 0.18215 * 4.0        # ~ the scale of sampled latents
    t0 = time.time()
    with torch.no_grad():
        raw = vae.decode(z * 1 / 0.18215)
        ref = unnormalize(raw).clip(0.0, 1.0)                         # viewfusion_zero_depth_rgb.py:161-163
    dt = time.time() - t0
    with torch.no_grad():
        mine_raw = O.vae_decode(sd, "vae.", z * 1 / 0.18215, ch=ch)
        mine = O.viewfusion_decode(sd, z, ch=ch)
    e_raw, e = rel_err(mine_raw, raw), float((mine - ref).abs().max())
    print(f"  vae decode ch={ch} z={zs}: ref {dt:.1f}s, oracle vs reference raw rel-max {e_raw:.2e}, image max-abs {e:.2e}; "
          f"raw std {float(raw.std()):.3f}, clipped fraction {float(((raw < -1) | (raw > 1)).float().mean()):.3f}")
    assert e_raw < 2e-5 and e < 2e-5, (e_raw, e)
    spec = json.dumps([[k, list(v.shape)] for k, v in sd.items()])
    if full:
        save(tag, z=z, raw=raw, image=ref, spec=spec)
    else:
        save(tag, z=z, raw_strided=raw[:, :, ::7, ::5].contiguous(), raw_std=raw.std(), raw_l2=raw.norm(),
             image_view0_sub=ref[0, :, ::4, ::4].contiguous(), spec=spec)


def gold_vae_encode(ch, B, res, tag, full=True):
    """The REAL AutoencoderKL.encode(normalize(x)).mode() * 0.18215 (ViewFusion.encode) on seeded images in [0,1]."""
    from external.sd1.ldm.models.autoencoder import AutoencoderKL
    from utils.common_utils import normalize
    dd = dict(VAE_DDCONFIG)
    dd["ch"] = ch
    vae = AutoencoderKL(ddconfig=dd, lossconfig=dict(target="torch.nn.Identity"), embed_dim=4)
    fill_ref(vae, "vae.")
    vae.eval()
    sd = {k: v for k, v in sd_of(vae, "vae.").items() if ".decoder." not in k and not k.startswith("vae.post_quant_conv")}
    g = torch.Generator().manual_seed(800 + ch + res)
    x = torch.rand(B, 3, res, res, generator=g) * 1.2 - 0.1            # a few values outside [0,1]: normalize() clips them
    t0 = time.time()
    with torch.no_grad():
        post = vae.encode(normalize(x))
        ref = post.mode() * 0.18215                                    # viewfusion_zero_depth_rgb.py:158-159
    dt = time.time() - t0
    with torch.no_grad():
        mine = O.viewfusion_encode(sd, x, ch=ch)
    e = rel_err(mine, ref)
    print(f"  vae encode ch={ch} res={res}: ref {dt:.1f}s, oracle vs reference rel-max {e:.2e}; latent std {float(ref.std()):.4f}")
    assert e < 2e-6, e
    spec = json.dumps([[k, list(v.shape)] for k, v in sd.items()])
    if full:
        save(tag, x=x, z=ref, logvar=post.logvar, spec=spec)
    else:
        save(tag, x_seed=np.int64(800 + ch + res), z=ref, spec=spec)


def gold_prepare_batch(tag, random_views, with_depths, seed):
    """The REAL ViewFusion.prepare_batch (viewfusion_zero_depth_rgb.py:165-273) on a 16-view GSO-rig batch of seeded 64x64
    images, with the reference VAE (ch=32, name-keyed fill) and the stub CLIP encoder on both sides."""
    import torch.nn as nn
    from external.sd1.ldm.models.autoencoder import AutoencoderKL
    from mvdfusion.viewfusion_zero_depth_rgb import ViewFusion

    class Facade(nn.Module):
        prepare_batch = ViewFusion.prepare_batch
        encode = ViewFusion.encode
        encode_clip = ViewFusion.encode_clip

        def __init__(self):
            super().__init__()
            dd = dict(VAE_DDCONFIG)
            dd["ch"] = 32
            self.vae = AutoencoderKL(ddconfig=dd, lossconfig=dict(target="torch.nn.Identity"), embed_dim=4)
            fill_ref(self.vae, "vae.")
            self.clip_image_encoder = syn.StubClipImageEncoder()
            self.z_scale_factor, self.embed_camera_pose = 0.18215, True

    m = Facade().eval()
    rig = syn.gso_rig()
    g = torch.Generator().manual_seed(seed)
    batch = dict(images=torch.rand(16, 3, 64, 64, generator=g), R=rig.R, T=rig.T, f=rig.focal_length, c=rig.principal_point)
    if with_depths:
        batch["depths"] = torch.rand(16, 1, 64, 64, generator=g)
    cfg = dict(input_batch_size=1, train_batch_size=4, random_views=random_views)
    gen = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        bl, bc, il, ic, cv = m.prepare_batch(batch, cfg, generator=gen)
    print(f"  prepare_batch random_views={random_views} depths={with_depths}: batch_latents {tuple(bl.shape)} std {float(bl.std()):.3f}, "
          f"clip_v_embed {tuple(cv.shape)}")
    save(tag, seed=np.int64(seed), batch_latents=bl, input_latents=il, clip_v_embed=cv, bc_R=bc.R, bc_T=bc.T,
         bc_f=bc.focal_length, bc_p=bc.principal_point, ic_R=ic.R, ic_T=ic.T, ic_f=ic.focal_length, ic_p=ic.principal_point)

def gold_train_loss(model_channels, V, D, tag, seed, S_img=256, grads_tag=None, lean=False):
    """The REAL ViewFusion.forward / p_losses (viewfusion_zero_depth_rgb.py:362-397) -- prepare_batch (reference VAE ch=32,
    stub CLIP), shared random timestep, q_sample, GridAttn + UNetWrapper.forward(is_train=True) WITH the condition dropout
    of unet.py:109-151, MSE -- on a seeded 16-view GSO batch.  The seed is chosen so that the dropout masks are not all-keep.
    Stored: the loss, the prediction (strided), and the random draws of this call in the reference's order (timestep,
    q_sample noise [re-derivable], depth-sample noise [re-derivable], dropout uniform)."""
    import torch.nn as nn
    from external.sd1.ldm.models.autoencoder import AutoencoderKL
    from mvdfusion.view_attn_efficient2 import GridAttn
    from mvdfusion.scheduler import DDPMScheduler
    from mvdfusion.unet import UNetWrapper
    from mvdfusion.viewfusion_zero_depth_rgb import ViewFusion
    S = S_img // 8

    class Facade(nn.Module):
        prepare_batch = ViewFusion.prepare_batch
        encode = ViewFusion.encode
        encode_clip = ViewFusion.encode_clip
        embed_time = ViewFusion.embed_time
        apply_model = ViewFusion.apply_model
        p_losses = ViewFusion.p_losses
        forward = ViewFusion.forward

        def __init__(self):
            super().__init__()
            dd = dict(VAE_DDCONFIG)
            dd["ch"] = 32
            self.vae = AutoencoderKL(ddconfig=dd, lossconfig=dict(target="torch.nn.Identity"), embed_dim=4)
            fill_ref(self.vae, "vae.")
            self.clip_image_encoder = syn.StubClipImageEncoder()
            self.z_scale_factor, self.embed_camera_pose = 0.18215, True
            self.view_attn = GridAttn(in_channels=5, input_size=S, output_dim=768, num_layers=3, z_near_far_scale=0.8,
                                      n_pts_per_ray=D)
            w = UNetWrapper.__new__(UNetWrapper)
            nn.Module.__init__(w)
            w.unet_model = _unet(model_channels, S)
            w.drop_conditions, w.drop_scheme, w.use_zero_123 = True, "default", True
            self.unet_model = w
            self.scheduler = DDPMScheduler(1000)
            self.cc_projection = nn.Sequential(nn.Linear(796, 768), nn.SiLU(True), nn.Linear(768, 768), nn.SiLU(True),
                                               nn.Linear(768, 768))
            self.time_embed_dim = 256
            self.time_embed = nn.Sequential(nn.Linear(256, 256), nn.SiLU(True), nn.Linear(256, 256))
            self.feed_prev_depth, self.objective = False, "noise"
            self.loss_fn = torch.nn.functional.mse_loss
            self.register_buffer("_device", torch.tensor([0.0]), persistent=False)

    m = Facade()
    for name in ("view_attn", "cc_projection", "time_embed"):
        fill_ref(getattr(m, name), name + ".")
    m.train()                                    # is_train=True path; the model has no dropout / batch-norm layers
    rig = syn.gso_rig()
    g = torch.Generator().manual_seed(seed)
    batch = dict(images=torch.rand(16, 3, S_img, S_img, generator=g), R=rig.R, T=rig.T, f=rig.focal_length, c=rig.principal_point,
                 depths=torch.rand(16, 1, S_img, S_img, generator=g))
    cfg = dict(input_batch_size=1, train_batch_size=V, random_views=False)
    # the draws of p_losses in order: randint (V,), randn (V,5,S,S), [GridAttn] normal == mean + std * randn (V,D,S,S), rand (V,)
    draw_seed = None
    for cand in range(1000, 1100):
        torch.manual_seed(cand)
        torch.randint(0, 1000, (V,))
        torch.randn(V, 5, S, S)
        torch.randn(V, D, S, S)
        r = torch.rand(V)
        if bool((r <= 0.2).any()) and bool((r > 0.2).any()):
            draw_seed = cand
            break
    torch.manual_seed(draw_seed)
    t_draw = torch.randint(0, 1000, (V,))
    torch.randn(V, 5, S, S)
    torch.randn(V, D, S, S)
    drop_rand = torch.rand(V)
    preds = {}
    orig = Facade.apply_model

    def spy(self, *a, **k):
        out = orig(self, *a, **k)
        preds["pred"] = out.detach().clone()
        return out

    Facade.apply_model = spy
    if grads_tag is not None:
        # the same call with autograd on: loss.backward() as train.py:90-95 does.  Stored: the gradients of the UNet output head
        # (GroupNorm32 affine + conv3x3), the gradient that reaches the head's input (strided), and the L2 norm of EVERY
        # parameter gradient (later backward slices pin against those).
        head = m.unet_model.unet_model.out
        grabbed = {}
        head.register_full_backward_hook(lambda mod, gin, gout: grabbed.__setitem__("dh", gin[0].detach().clone()))
        last = m.unet_model.unet_model.output_blocks[11]       # ResBlock + SpatialTransformer + ViewAlignedFeatureTransformer
        last.register_full_backward_hook(lambda mod, gin, gout: grabbed.__setitem__("dcat", gin[0].detach().clone()))
        torch.manual_seed(draw_seed)
        loss_g = m(batch, cfg)
        loss_g.backward()
        named = [(n, p) for n, p in m.named_parameters() if p.grad is not None]
        hp = "unet_model.unet_model.out."
        g = {n: p.grad.detach().clone() for n, p in named if n.startswith(hp)}
        dh = grabbed["dh"]                                     # (V, mc, S, S)
        bp = "unet_model.unet_model.output_blocks.11."
        blk = [(n, p.grad.detach().clone()) for n, p in named if n.startswith(bp)]
        dcat = grabbed["dcat"]                                 # gradient at the input of the last output block (V, 2 mc, S, S)
        extra = {f"blk11_g{i}": gr for i, (_, gr) in enumerate(blk)}
        if lean:          # full width: only the two-number fingerprints of every gradient + strided activations' gradients (MBs otherwise)
            extra = {}
            g = {k: v.flatten()[:8] for k, v in g.items()}
        save(grads_tag, blk11_names=np.array([n for n, _ in blk]), dcat_strided=dcat[:, :, ::3, ::5].contiguous(), dcat_norm=dcat.norm(),
             **extra, loss=loss_g.detach(), out0_weight=g[hp + "0.weight"], out0_bias=g[hp + "0.bias"], out2_weight=g[hp + "2.weight"],
             out2_bias=g[hp + "2.bias"], dh_strided=dh[:, :, ::3, ::5].contiguous(), dh_norm=dh.norm(),
             grad_names=np.array([n for n, _ in named]), grad_norms=np.array([float(p.grad.double().norm() if lean else p.grad.norm()) for _, p in named], dtype=np.float64),      # (fp32 norm of a
             # 14.7 M-element gradient is itself only good to ~1e-3: the full-width fixture takes it in float64)
             # projection of every gradient onto a seeded N(0,1) direction (CPU generator, seed 1000 + index): with the norm, a
             # two-number fingerprint per parameter that a wrong layout / sign / missing term cannot match
             grad_projs=np.array([float((p.grad.double().flatten() * torch.randn(p.numel(), generator=torch.Generator().manual_seed(1000 + i))
                                         .double()).sum()) for i, (_, p) in enumerate(named)], dtype=np.float64),
             batch_seed=np.int64(seed), draw_seed=np.int64(draw_seed), t=t_draw, drop_rand=drop_rand)
        print(f"  train grads: loss {float(loss_g):.6f}, |dh| {float(dh.norm()):.4e}, {len(named)} parameter gradients")
        m.zero_grad(set_to_none=True)
    torch.manual_seed(draw_seed)
    t0 = time.time()
    with torch.no_grad():
        loss = m(batch, cfg)
    print(f"  train loss mc={model_channels} V={V} D={D}: ref {time.time() - t0:.1f}s, loss {float(loss):.6f}, draw seed {draw_seed}, "
          f"t {int(t_draw[0])}, drop uniform {[round(float(x), 3) for x in drop_rand]}")
    save(tag, loss=loss, pred_strided=preds["pred"][:, :, ::3, ::5].contiguous(), pred_std=preds["pred"].std(), batch_seed=np.int64(seed),
         draw_seed=np.int64(draw_seed), t=t_draw, drop_rand=drop_rand)


def gold_clip(name, tag, B=2, R=256, seed=41):
    """The REAL FrozenCLIPImageEmbedder (encoders/modules.py:402-441: bicubic resize -> (x+1)/2 -> CLIP mean/std ->
    model.encode_image -> .float(), .encode() adds the token axis) on seeded images in [-1, 1].  OpenAI's `clip` package is not in
    the reference tree, so `clip.load` is the structural restatement in oracle/shims.py (published CLIP model.py); weights =
    name-keyed fill.  Pins the reference's preprocessing / call order and the oracle restatement; the ViT arithmetic itself is
    only pinned to the published architecture (parity unpinned, see oracle/ref_torch.py)."""
    from external.sd1.ldm.modules.encoders.modules import FrozenCLIPImageEmbedder
    enc = FrozenCLIPImageEmbedder(model=name)
    fill_ref(enc, "clip_image_encoder.")
    enc.eval()
    sd = {k: v for k, v in sd_of(enc, "clip_image_encoder.").items() if ".visual." in k}
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, R, R, generator=g) * 2.0 - 1.0
    t0 = time.time()
    with torch.no_grad():
        ref = enc.encode(x)
    dt = time.time() - t0
    heads = {"ViT-L/14": 16, "tiny-test": 2}[name]
    with torch.no_grad():
        mine = O.clip_image_embed(sd, x, heads=heads)
    e = rel_err(mine, ref)
    print(f"  clip {name}: ref {dt:.1f}s, out {tuple(ref.shape)} std {float(ref.std()):.4f}, oracle vs reference rel-max {e:.2e}")
    assert e < 2e-5, e
    save(tag, seed=np.int64(seed), out=ref, spec=json.dumps([[k, list(v.shape)] for k, v in enc.state_dict().items()]))


def gold_ckpt_remap(model_channels=32):
    """The REAL checkpoint loader (utils/load_model.py:26-110 with UNetWrapper's replace_key / param_mapper / remove_keys,
    mvdfusion/unet.py:70-93, viewfusion_zero_depth_rgb.py:69) on a synthetic zero123-shaped checkpoint: per-key sums of the
    resulting UNet state_dict."""
    import tempfile
    from utils.load_model import load_model_from_config
    p = dict(UNET_PARAMS)
    p["model_channels"] = model_channels
    cfg = {"target": "mvdfusion.unet.UNetModel", "params": p}
    base = _unet(model_channels)
    ck = syn.synthetic_zero123_ckpt(base.state_dict())
    pm = {}
    for pre_src, pre_dst in (("output_blocks.5.2.conv.", "output_blocks.5.3.conv."), ("output_blocks.8.2.conv.", "output_blocks.8.3.conv.")):
        for leaf in ("weight", "bias"):
            pm[pre_src + leaf] = pre_dst + leaf
    for leaf in ("in_layers.0", "in_layers.2", "emb_layers.1", "out_layers.0", "out_layers.3"):
        for wb in ("weight", "bias"):
            pm[f"middle_block.2.{leaf}.{wb}"] = f"middle_block.3.{leaf}.{wb}"
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "zero123.ckpt")
        torch.save(ck, path)
        torch.manual_seed(0)
        net = load_model_from_config(cfg, path, verbose=False, replace_key=["model.diffusion_model.", ""],
                                     ignore_keys=["aligned_attn_"], param_mapper=pm,
                                     remove_keys=["input_blocks.0.0.weight", "out.2.weight", "out.2.bias"])
    sd = net.state_dict()
    loaded = {k: float(v.double().sum()) for k, v in sd.items() if "aligned_attn_" not in k and
              k not in ("input_blocks.0.0.weight", "out.2.weight", "out.2.bias")}
    # every loaded key must carry the checkpoint's values (not the initialiser's)
    for k, vsum in loaded.items():
        src = k
        for dst, s0 in (("middle_block.3.", "middle_block.2."), ("output_blocks.5.3.conv.", "output_blocks.5.2.conv."),
                        ("output_blocks.8.3.conv.", "output_blocks.8.2.conv.")):
            if src.startswith(dst):
                src = s0 + src[len(dst):]
        assert abs(vsum - float(ck["state_dict"]["model.diffusion_model." + src].double().sum())) < 1e-9, k
    save("ckpt_remap_mc32", keys=json.dumps(sorted(loaded)), sums=np.asarray([loaded[k] for k in sorted(loaded)]))
    print(f"  checkpoint remap: {len(loaded)} keys loaded by the reference loader from the synthetic zero123 checkpoint")


ALL = {
    "schedule": gold_schedule,
    "cameras": gold_cameras,
    "spec32": lambda: gold_spec(32),
    "spec320": lambda: gold_spec(320),
    "gridattn": gold_gridattn,
    "unet32": lambda: gold_unet(32, 4, 1, "unet_mc32_v4_d1"),
    "unet32_d3": lambda: gold_unet(32, 2, 3, "unet_mc32_v2_d3", t_val=501),
    "unet64": lambda: gold_unet(64, 2, 1, "unet_mc64_v2_d1", t_val=21),
    "unet320": lambda: gold_unet(320, 2, 1, "unet_mc320_v2_d1"),
    "step32": lambda: gold_step(32, 4, 1, "step_mc32_v4_d1"),
    "step32_d3": lambda: gold_step(32, 2, 3, "step_mc32_v2_d3", indices=(30,)),
    "step320": lambda: gold_step(320, 4, 1, "step_mc320_v4_d1", indices=(49, 0)),
    "step32_s64": lambda: gold_step(32, 4, 1, "step_mc32_v4_d1_s64", indices=(49, 0), S=64, lean=True),
    "step320_v8": lambda: gold_step(320, 8, 1, "step_mc320_v8_d1", indices=(49,), lean=True),
    "step320_v8_s64": lambda: gold_step(320, 8, 1, "step_mc320_v8_d1_s64", indices=(49,), S=64, lean=True),      # BASELINE configs[3]
    "gridattn_v8_s64": gold_gridattn_v8_s64,
    "gridattn_prevdepth": gold_gridattn_prev_depth,
    "sample_prevdepth": gold_sample_feed_prev_depth,
    "unet320_s64": lambda: gold_unet(320, 2, 1, "unet_mc320_v2_d1_s64", S=64, full=False),
    "unet320_d3": lambda: gold_unet(320, 2, 3, "unet_mc320_v2_d3", t_val=501),
    "gridattn_v15": gold_gridattn_v15,
    "gridattn_v7": lambda: gold_gridattn_small(7, 1, 8, 621, "gridattn_v7_d1"),          # configs/mvd_train.yaml:97 (7 views)
    "gridattn_v5_d3": lambda: gold_gridattn_small(5, 3, 9, 161, "gridattn_v5_d3"),       # configs/mvd_train.yaml:90 (5 views, D = 3)
    "step320_v15": lambda: gold_step(320, 15, 1, "step_mc320_v15_d1", indices=(49,), lean=True),     # configs/mvd_gso.yaml:97 as shipped
    "step320_v8_d3": lambda: gold_step(320, 8, 3, "step_mc320_v8_d3", indices=(30,), lean=True),     # forward of BASELINE configs[4]'s geometry
    "ckpt_remap": gold_ckpt_remap,
    "clip_tiny": lambda: gold_clip("tiny-test", "clip_tiny"),
    "clip_l14": lambda: gold_clip("ViT-L/14", "clip_vit_l14"),
    "train32_d3": lambda: gold_train_loss(32, 4, 3, "train_loss_mc32_v4_d3", seed=31, grads_tag="train_grads_mc32_v4_d3"),
    "train320_v8_d3": lambda: gold_train_loss(320, 8, 3, "train_loss_mc320_v8_d3", seed=37, grads_tag="train_grads_mc320_v8_d3", lean=True),   # BASELINE configs[4]
    "train320_d3": lambda: gold_train_loss(320, 2, 3, "train_loss_mc320_v2_d3", seed=35, grads_tag="train_grads_mc320_v2_d3", lean=True),
    "traj32": lambda: gold_trajectory(32, 4, 1, "traj_mc32_v4_d1", steps=5),
    "traj320": lambda: gold_trajectory(320, 4, 1, "traj_mc320_v4_d1_50steps", steps=50),
    "traj320_f64": lambda: gold_trajectory_f64(320, 4, 1, "traj_mc320_v4_d1_50steps_f64", steps=50),
    "vae32": lambda: gold_vae_decode(32, 2, 8, "vae_dec_ch32_z8"),
    "vae128": lambda: gold_vae_decode(128, 2, 8, "vae_dec_ch128_z8"),
    "vae128_z32": lambda: gold_vae_decode(128, 1, 32, "vae_dec_ch128_z32", full=False),
    "vaeenc32": lambda: gold_vae_encode(32, 2, 64, "vae_enc_ch32_r64"),
    "vaeenc128": lambda: gold_vae_encode(128, 2, 64, "vae_enc_ch128_r64"),
    "vaeenc128_r256": lambda: gold_vae_encode(128, 1, 256, "vae_enc_ch128_r256", full=False),
    "prep": lambda: gold_prepare_batch("prepare_batch_fixed", False, False, 21),
    "prep_rand": lambda: gold_prepare_batch("prepare_batch_random_depths", True, True, 22),
}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    for k, fn in ALL.items():
        if a.only and k not in a.only:
            continue
        print(f"[{k}]")
        fn()
