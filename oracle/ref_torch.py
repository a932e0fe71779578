"""CPU restatement of the MVD-Fusion denoising hot path -- TEST INFRASTRUCTURE (see oracle/__init__.py).

Plain fp32 PyTorch on CPU, functional style over a flat ``sd = {state_dict key: tensor}`` whose keys are
the reference's own (SURVEY.md section 8b "state_dict contract").  Every function cites the reference
file:line it restates.  No pytorch3d / timm / omegaconf dependency: the camera algebra is written in
closed form (pinned against the reference import by oracle/make_golden.py and by the known-answer tests).

Parity status: PINNED -- checked against the imported reference on the fixtures under tests/golden/
(tests/test_cpu_oracle_and_host.py, tolerance 1e-5 relative-to-max per tensor).
"""
import math

import torch
import torch.nn.functional as F

# ---------------------------------------------------------------------------------------------
# schedule tables
# ---------------------------------------------------------------------------------------------


def ddpm_tables(num_timesteps=1000):
    """mvdfusion/scheduler.py:11-36 (fp32 linspace**2, fp32 cumprod)."""
    betas = torch.linspace(0.00085 ** 0.5, 0.0120 ** 0.5, num_timesteps, dtype=torch.float32) ** 2
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, dim=0)
    return {
        "betas": betas,
        "alphas": alphas,
        "alphas_cumprod": ac,
        "sqrt_alphas_cumprod": torch.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1 - ac),
    }


def ddim_schedule(tables, ddim_num_steps=50, eta=1.0):
    """mvdfusion/sampler.py:25-39 + external/sd1/ldm/modules/diffusionmodules/util.py:46-60."""
    T = tables["alphas_cumprod"].shape[0]
    c = T // ddim_num_steps
    ts = torch.arange(0, T, c, dtype=torch.int64) + 1
    ac = tables["alphas_cumprod"]
    a = ac[ts].double()
    a_prev = torch.cat([ac[0:1], ac[ts[:-1]]], 0)
    sig = eta * torch.sqrt((1 - a_prev) / (1 - a) * (1 - a / a_prev))
    return {
        "timesteps": ts,
        "alphas": a.float(),
        "alphas_prev": a_prev.float(),
        "sigmas": sig.float(),
        "sqrt_one_minus_alphas": torch.sqrt(1.0 - a.float()).float(),
    }


def timestep_embedding(timesteps, dim, max_period=10000):
    """external/sd1/.../diffusionmodules/util.py:152-172 and mvdfusion/embedder.py:114-134 (cos first)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


# ---------------------------------------------------------------------------------------------
# camera algebra (pytorch3d semantics, closed form)
# ---------------------------------------------------------------------------------------------


def camera_center(R, T):
    """C = -T R^T (row-vector convention)."""
    return -torch.einsum("ni,nji->nj", T, R)


def project_ndc(R, T, f, p, pts):
    """pts (P,3) world -> (N,P,3): X_cam = X R + T; (fx X/Z + px, fy Y/Z + py, 1/Z)."""
    xc = torch.einsum("pi,nij->npj", pts, R) + T[:, None, :]
    z = xc[..., 2:3]
    xy = f[:, None, :] * xc[..., :2] / z + p[:, None, :]
    return torch.cat([xy, 1.0 / z], dim=-1)


def unproject_ndc(R, T, f, p, xy, depth):
    """xy (N,P,2), depth (N,P) -> world (N,P,3): X_cam = ((x-px) d / fx, (y-py) d / fy, d); X_w = (X_cam - T) R^T."""
    xc = torch.cat([(xy - p[:, None, :]) * depth[..., None] / f[:, None, :], depth[..., None]], dim=-1)
    return torch.einsum("npi,nji->npj", xc - T[:, None, :], R)


def harmonic_embedding(x, n_harmonic=7, omega0=0.1):
    """utils/common_utils.py:229-244: [sin(x w_k), cos(x w_k), x], index = dim*7 + k."""
    freqs = (2.0 ** torch.arange(n_harmonic, dtype=torch.float32)) * omega0
    e = (x[..., None] * freqs).reshape(*x.shape[:-1], -1)
    return torch.cat((e.sin(), e.cos(), x), dim=-1)


# ---------------------------------------------------------------------------------------------
# GridAttn (mvdfusion/view_attn_efficient2.py)
# ---------------------------------------------------------------------------------------------


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _dit_block(sd, pre, x, c, num_heads=8):
    """DiTBlock.forward view_attn_efficient2.py:63-67 with timm Attention / Mlp (restated in oracle/shims.py)."""
    mod = _lin(sd, pre + "adaLN_modulation.1", F.silu(c))
    sh1, sc1, g1, sh2, sc2, g2 = mod.chunk(6, dim=1)
    C = x.shape[-1]

    def modulate(h, shift, scale):
        return h * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)

    h = modulate(F.layer_norm(x, (C,), eps=1e-6), sh1, sc1)
    B, N, _ = h.shape
    hd = C // num_heads
    qkv = _lin(sd, pre + "attn.qkv", h).reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    attn = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
    a = (attn @ v).transpose(1, 2).reshape(B, N, C)
    x = x + g1.unsqueeze(1) * _lin(sd, pre + "attn.proj", a)
    h = modulate(F.layer_norm(x, (C,), eps=1e-6), sh2, sc2)
    h = _lin(sd, pre + "mlp.fc2", F.gelu(_lin(sd, pre + "mlp.fc1", h)))
    return x + g2.unsqueeze(1) * h


def gridattn_tokens(feat, in_feat, cams, in_cam, depth, S):
    """view_attn_efficient2.py:269-370: the (V_ref, V_query, S*S*D, 723) token tensor.

    feat (V,256,S,S) z-embedded noisy latents, in_feat (1,256,S,S), cams/in_cam = dict(R,T,f,p),
    depth (V,D,S,S) metric depth samples.
    """
    V, D = depth.shape[0], depth.shape[1]
    R, T, f, p = cams["R"], cams["T"], cams["f"], cams["p"]
    # ray grid utils/ray_utils.py:263-269 (x along columns, y along rows, +X left / +Y up)
    half = 1.0 / float(S)
    lin = torch.linspace(1.0 - half, -1.0 + half, S, dtype=torch.float32).to(R.dtype)
    yy, xx = torch.meshgrid(lin, lin, indexing="ij")
    xy = torch.stack([xx, yy], dim=-1).reshape(1, S * S, 2).expand(V, -1, -1)
    # utils/ray_utils.py:175-202: unproject planes z=1 and z=2
    ones = torch.ones(V, S * S, dtype=R.dtype)
    p1 = unproject_ndc(R, T, f, p, xy, ones)
    p2 = unproject_ndc(R, T, f, p, xy, 2.0 * ones)
    dirs = p2 - p1
    orig = p1 - dirs
    lengths = depth.permute(0, 2, 3, 1).reshape(V, S * S, D)           # ray_utils.py:367-369
    xyz = orig[:, :, None, :] + lengths[..., None] * dirs[:, :, None, :]  # (V, S*S, D, 3)
    pts = xyz.reshape(V * S * S * D, 3)

    def gather(fmap, cam):
        ndc = project_ndc(cam["R"], cam["T"], cam["f"], cam["p"], pts)       # :303,:321
        grid = -ndc[..., :2].unsqueeze(2)                                   # :312
        out = F.grid_sample(fmap, grid, align_corners=True, mode="bilinear", padding_mode="border")
        n = fmap.shape[0]
        return out[..., 0].reshape(n, -1, V, S * S * D).permute(0, 2, 3, 1)  # 'v c (b n) -> v b n c'

    ref_feat = gather(feat, cams)                                           # (V, V, n, 256)
    inp_feat = gather(in_feat, in_cam).expand(V, -1, -1, -1)
    C = camera_center(R, T)                                                 # (V,3)
    ref_dir = pts[None, :, :] - C[:, None, :]                               # (V, V*n, 3)
    ref_dir = ref_dir.reshape(V, V, S * S * D, 3)
    ref_depth = harmonic_embedding(torch.linalg.norm(ref_dir, dim=-1, keepdim=True))
    ref_dir = F.normalize(ref_dir, dim=-1)
    o = C[:, None, None, :].expand_as(ref_dir)
    ref_pl = harmonic_embedding(torch.cat((ref_dir, torch.cross(o, ref_dir, dim=-1)), dim=-1))
    qdir = F.normalize(dirs, dim=-1)                                        # (V, S*S, 3)
    qdir = qdir[:, :, None, :].expand(-1, -1, D, -1).reshape(1, V, S * S * D, 3)
    qo = C[None, :, None, :].expand_as(qdir)
    q_pl = harmonic_embedding(torch.cat((qdir, torch.cross(qo, qdir, dim=-1)), dim=-1)).expand(V, -1, -1, -1)
    q_dep = harmonic_embedding(lengths.reshape(1, V, S * S * D, 1)).expand(V, -1, -1, -1)
    mask = torch.ones(V, V, S * S * D, 1, dtype=R.dtype)
    return torch.cat((ref_feat, inp_feat, ref_pl, ref_depth, q_pl, q_dep, mask), dim=-1)


def gridattn_forward(sd, pre, noisy_latents, cams, t_embed, t, tables, depth_noise, input_latents, in_cam,
                     n_pts_per_ray=1, depth_scale=2.0, depth_shift=0.5, return_tokens=False, overwrite_attn_depth=None):
    """GridAttn.forward view_attn_efficient2.py:413-442 + aggregate_features :269-410.

    depth_noise (V,D,S,S) ~ N(0,1) replaces the in-place torch.normal draw (:431): on CPU
    torch.normal(mean,std) == mean + std*randn bit-for-bit from the same stream.
    """
    V, _, S, _ = noisy_latents.shape
    D = n_pts_per_ray
    sac = tables["sqrt_alphas_cumprod"][t]
    std = tables["sqrt_one_minus_alphas_cumprod"][t] / sac / 10.0
    if overwrite_attn_depth is None:
        dch = noisy_latents[:, 4:] / sac[:, None, None, None]
    else:                                       # :423-426 (feed_prev_depth): the given depth map replaces the x0-style estimate
        dch = overwrite_attn_depth
    dch = dch.expand(-1, D, -1, -1)
    samples = dch + std[:, None, None, None] * depth_noise
    depth = torch.clip((samples + 1.0) / 2.0, 0.0, 1.0) * depth_scale + depth_shift

    def zemb(x):
        return F.gelu(_lin(sd, pre + "z_embedder.0", x.permute(0, 2, 3, 1))).permute(0, 3, 1, 2)

    feat, in_feat = zemb(noisy_latents), zemb(input_latents)
    z = gridattn_tokens(feat, in_feat, cams, in_cam, depth, S)             # (Vref, Vq, n, 723)
    n = z.shape[2]
    x = z.permute(1, 2, 0, 3).reshape(V * n, V, -1)                        # 'v (b n) c -> (b n) v c'
    if return_tokens:
        return x
    x = F.gelu(_lin(sd, pre + "pre_layer_b.0", x))
    c = t_embed[:1]
    for li in range(3):
        x = _dit_block(sd, f"{pre}aggregation_transformer.layer_list.{li}.", x, c)
    w = _lin(sd, pre + "aggregation_transformer.weight_layer", x).softmax(dim=-2)
    agg = (x * w).sum(dim=-2)
    out = _lin(sd, pre + "final_layer_b", agg)
    return out.reshape(V, S, S, D, -1)


# ---------------------------------------------------------------------------------------------
# UNet (mvdfusion/unet.py + external/sd1 blocks)
# ---------------------------------------------------------------------------------------------


def _conv(sd, name, x, stride=1, padding=1):
    return F.conv2d(x, sd[name + ".weight"], sd.get(name + ".bias"), stride=stride, padding=padding)


def _resblock(sd, pre, x, emb):
    """ResBlock._forward openaimodel.py:255-275 (no updown, no scale-shift)."""
    h = F.silu(F.group_norm(x, 32, sd[pre + "in_layers.0.weight"], sd[pre + "in_layers.0.bias"], eps=1e-5))
    h = _conv(sd, pre + "in_layers.2", h)
    e = _lin(sd, pre + "emb_layers.1", F.silu(emb))
    h = h + e[:, :, None, None]
    h = F.silu(F.group_norm(h, 32, sd[pre + "out_layers.0.weight"], sd[pre + "out_layers.0.bias"], eps=1e-5))
    h = _conv(sd, pre + "out_layers.3", h)
    if pre + "skip_connection.weight" in sd:
        x = _conv(sd, pre + "skip_connection", x, padding=0)
    return x + h


def _cross_attention(sd, pre, x, context, heads):
    """CrossAttention.forward external/sd1/ldm/modules/attention.py:170-193."""
    ctx = x if context is None else context
    q, k, v = _lin(sd, pre + "to_q", x), _lin(sd, pre + "to_k", ctx), _lin(sd, pre + "to_v", ctx)
    b, n, c = q.shape
    d = c // heads

    def split(t):
        return t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3)

    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * d ** -0.5
    out = torch.einsum("bhij,bhjd->bhid", sim.softmax(dim=-1), v)
    out = out.permute(0, 2, 1, 3).reshape(b, n, c)
    return _lin(sd, pre + "to_out.0", out)


def _geglu_ff(sd, pre, x):
    """FeedForward/GEGLU attention.py:37-64."""
    h = _lin(sd, pre + "net.0.proj", x)
    a, g = h.chunk(2, dim=-1)
    return _lin(sd, pre + "net.2", a * F.gelu(g))


def _ln(sd, name, x):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps=1e-5)


def _spatial_transformer(sd, pre, x, context, heads):
    """SpatialTransformer.forward attention.py:268-287 + BasicTransformerBlock._forward :219-223."""
    b, c, h, w = x.shape
    x_in = x
    x = F.group_norm(x, 32, sd[pre + "norm.weight"], sd[pre + "norm.bias"], eps=1e-6)
    x = _conv(sd, pre + "proj_in", x, padding=0)
    x = x.permute(0, 2, 3, 1).reshape(b, h * w, c)
    tb = pre + "transformer_blocks.0."
    x = _cross_attention(sd, tb + "attn1.", _ln(sd, tb + "norm1", x), None, heads) + x
    x = _cross_attention(sd, tb + "attn2.", _ln(sd, tb + "norm2", x), context, heads) + x
    x = _geglu_ff(sd, tb + "ff.", _ln(sd, tb + "norm3", x)) + x
    x = x.reshape(b, h, w, c).permute(0, 3, 1, 2)
    x = _conv(sd, pre + "proj_out", x, padding=0)
    return x + x_in


def _vaft(sd, pre, x, volume_levels, heads, image_size):
    """ViewAlignedFeatureTransformer.forward mvdfusion/attention.py:119-145 + DualAttnetionBlock._forward :43-66."""
    b, c, h, w = x.shape
    level = {image_size: 0, image_size // 2: 1, image_size // 4: 2, image_size // 8: 3}[h]
    ctx = volume_levels[level]                      # (b,h,w,d,768)
    ctx = ctx.reshape(b * h * w, ctx.shape[3], ctx.shape[4])
    x_in = x
    x = F.group_norm(x, 32, sd[pre + "aligned_attn_norm.weight"], sd[pre + "aligned_attn_norm.bias"], eps=1e-6)
    x = x.permute(0, 2, 3, 1).reshape(b, h * w, c)
    x = _lin(sd, pre + "aligned_attn_proj_in", x)
    tb = pre + "aligned_attn_transformer_blocks.0."
    x = _cross_attention(sd, tb + "attn1.", _ln(sd, tb + "norm1", x), None, heads) + x
    xp = x.reshape(b * h * w, 1, c)
    xp = _cross_attention(sd, tb + "attn2.", _ln(sd, tb + "norm2", xp), ctx, heads) + xp
    x = xp[:, 0].reshape(b, h * w, c)
    x = _geglu_ff(sd, tb + "ff.", _ln(sd, tb + "norm3", x)) + x
    x = _lin(sd, pre + "aligned_attn_proj_out", x)
    x = x.reshape(b, h, w, c).permute(0, 3, 1, 2)
    return x + x_in


def unet_layout(model_channels=320, channel_mult=(1, 2, 4, 4), num_res_blocks=2, attention_resolutions=(4, 2, 1)):
    """Block layout of mvdfusion/unet.py:320-494 as lists of ('res'|'st'|'vaft'|'down'|'up'|'conv') per block."""
    inp = [["conv"]]
    ds = 1
    for level, mult in enumerate(channel_mult):
        for _ in range(num_res_blocks):
            layers = ["res"]
            if ds in attention_resolutions:
                layers.append("st")
            inp.append(layers)
        if level != len(channel_mult) - 1:
            inp.append(["down"])
            ds *= 2
    mid = ["res", "st", "vaft", "res"]
    out = []
    for level, mult in list(enumerate(channel_mult))[::-1]:
        for i in range(num_res_blocks + 1):
            layers = ["res"]
            if ds in attention_resolutions:
                layers += ["st", "vaft"]
            if level and i == num_res_blocks:
                layers.append("up")
                ds //= 2
            out.append(layers)
    return inp, mid, out


def unet_forward(sd, pre, x, timesteps, context, volume_levels, model_channels=320, heads=8, image_size=32,
                 channel_mult=(1, 2, 4, 4), num_res_blocks=2, attention_resolutions=(4, 2, 1)):
    """UNetModel.forward mvdfusion/unet.py:524-556."""
    inp, mid, out = unet_layout(model_channels, channel_mult, num_res_blocks, attention_resolutions)
    emb = timestep_embedding(timesteps, model_channels).to(x.dtype)
    emb = _lin(sd, pre + "time_embed.2", F.silu(_lin(sd, pre + "time_embed.0", emb)))

    def run(block_pre, layers, h):
        for li, kind in enumerate(layers):
            p = f"{block_pre}{li}."
            if kind == "conv":
                h = _conv(sd, p[:-1], h)
            elif kind == "res":
                h = _resblock(sd, p, h, emb)
            elif kind == "st":
                h = _spatial_transformer(sd, p, h, context, heads)
            elif kind == "vaft":
                h = _vaft(sd, p, h, volume_levels, heads, image_size)
            elif kind == "down":
                h = _conv(sd, p + "op", h, stride=2)
            elif kind == "up":
                h = F.interpolate(h, scale_factor=2, mode="nearest")
                h = _conv(sd, p + "conv", h)
        return h

    hs = []
    h = x
    for bi, layers in enumerate(inp):
        h = run(f"{pre}input_blocks.{bi}.", layers, h)
        hs.append(h)
    h = run(f"{pre}middle_block.", mid, h)
    for bi, layers in enumerate(out):
        h = torch.cat([h, hs.pop()], dim=1)
        h = run(f"{pre}output_blocks.{bi}.", layers, h)
    h = F.silu(F.group_norm(h, 32, sd[pre + "out.0.weight"], sd[pre + "out.0.bias"], eps=1e-5))
    return _conv(sd, pre + "out.2", h)


def volume_pyramid(volume_feats, num_levels=4):
    """UNetWrapper.get_volume_feats_pyramid mvdfusion/unet.py:198-209 (area pooling x{1,1/2,1/4,1/8})."""
    b, h, w, d, c = volume_feats.shape
    v = volume_feats.permute(0, 3, 4, 1, 2).reshape(b * d, c, h, w)
    levels = []
    for i in range(num_levels):
        lf = F.interpolate(v, scale_factor=0.5 ** i, mode="area")
        hh, ww = lf.shape[-2:]
        levels.append(lf.reshape(b, d, c, hh, ww).permute(0, 3, 4, 1, 2))
    return levels


def unet_cfg(sd, pre, x, t1, clip_embed, volume_feats, x_concat, scale, **kw):
    """UNetWrapper.predict_with_unconditional_scale mvdfusion/unet.py:167-196 (use_zero_123=True)."""
    xc = x_concat.clone()
    xc[:, :4] = xc[:, :4] / 0.18215
    x_ = torch.cat([x, xc], 1)
    x_null = torch.cat([x, torch.zeros_like(xc)], 1)
    s = unet_forward(sd, pre, x_, t1, clip_embed, volume_pyramid(volume_feats), **kw)
    s_uc = unet_forward(sd, pre, x_null, t1, torch.zeros_like(clip_embed),
                        volume_pyramid(torch.zeros_like(volume_feats)), **kw)
    return s_uc + scale * (s - s_uc)


def unet_train_forward(sd, pre, x, t1, clip_embed, volume_feats, x_concat, **kw):
    """UNetWrapper.forward mvdfusion/unet.py:129-164 with is_train=False-equivalent (no condition dropout)."""
    xc = x_concat.clone()
    xc[:, :4] = xc[:, :4] / 0.18215
    return unet_forward(sd, pre, torch.cat([x, xc], 1), t1, clip_embed, volume_pyramid(volume_feats), **kw)


# ---------------------------------------------------------------------------------------------
# ViewFusion.apply_model + DDIM step
# ---------------------------------------------------------------------------------------------


def embed_time(sd, t, dim=256):
    """ViewFusion.embed_time viewfusion_zero_depth_rgb.py:276-279."""
    e = timestep_embedding(t, dim).to(sd["time_embed.0.weight"].dtype)
    return _lin(sd, "time_embed.2", F.silu(_lin(sd, "time_embed.0", e)))


def cc_projection(sd, clip_v_embed):
    """viewfusion_zero_depth_rgb.py:110,322: Linear-SiLU-Linear-SiLU-Linear."""
    h = F.silu(_lin(sd, "cc_projection.0", clip_v_embed))
    h = F.silu(_lin(sd, "cc_projection.2", h))
    return _lin(sd, "cc_projection.4", h)


def apply_model(sd, noisy_latents, cams, input_latents, in_cam, clip_v_embed, t, tables, depth_noise,
                cfg_scale=2.5, n_pts_per_ray=1, unet_kw=None, prev_depth=None):
    """ViewFusion.apply_model viewfusion_zero_depth_rgb.py:282-345."""
    unet_kw = unet_kw or {}
    V = noisy_latents.shape[0]
    t_embed = embed_time(sd, t)
    vol = gridattn_forward(sd, "view_attn.", noisy_latents, cams, t_embed, t, tables, depth_noise,
                           input_latents, in_cam, n_pts_per_ray=n_pts_per_ray, overwrite_attn_depth=prev_depth)
    il = input_latents.expand(V, -1, -1, -1)
    clip_embed = cc_projection(sd, clip_v_embed)
    pre = "unet_model.unet_model."
    if cfg_scale == 1.0:
        return unet_train_forward(sd, pre, noisy_latents, t[:1], clip_embed, vol, il, **unet_kw)
    return unet_cfg(sd, pre, noisy_latents, t[:1], clip_embed, vol, il, cfg_scale, **unet_kw)


def ddim_update(x, eps, ddim, index, noise):
    """DDIMSampler.denoise_apply_impl mvdfusion/sampler.py:43-66; noise=None at index 0."""
    a_t = ddim["alphas"][index]
    a_prev = ddim["alphas_prev"][index]
    s1m = ddim["sqrt_one_minus_alphas"][index]
    sig = ddim["sigmas"][index]
    pred_x0 = (x - s1m * eps) / a_t.sqrt()
    dir_xt = torch.clamp(1.0 - a_prev - sig ** 2, min=1e-7).sqrt() * eps
    x_prev = a_prev.sqrt() * pred_x0 + dir_xt
    if noise is not None:
        x_prev = x_prev + sig * noise
    return x_prev, pred_x0


def denoise_step(sd, x, cams, input_latents, in_cam, clip_v_embed, tables, ddim, index, depth_noise, step_noise,
                 cfg_scale=2.5, n_pts_per_ray=1, unet_kw=None, prev_depth=None):
    """DDIMSampler.denoise_apply mvdfusion/sampler.py:69-88."""
    V = x.shape[0]
    t = torch.full((V,), int(ddim["timesteps"][index]), dtype=torch.long)
    eps = apply_model(sd, x, cams, input_latents, in_cam, clip_v_embed, t, tables, depth_noise,
                      cfg_scale=cfg_scale, n_pts_per_ray=n_pts_per_ray, unet_kw=unet_kw, prev_depth=prev_depth)
    return ddim_update(x, eps, ddim, index, step_noise if index > 0 else None)


# ---------------------------------------------------------------------------------------------
# VAE decode (the caller right after the sampling loop)
# external/sd1/ldm/models/autoencoder.py:331-334, external/sd1/ldm/modules/diffusionmodules/model.py:462-577
# ---------------------------------------------------------------------------------------------


def _vae_norm(sd, name, x):
    """Normalize = GroupNorm(32, C, eps=1e-6, affine) model.py:37-38."""
    return F.group_norm(x, 32, sd[name + ".weight"], sd[name + ".bias"], eps=1e-6)


def _swish(x):
    """nonlinearity model.py:31-33 -- spelled x*sigmoid(x) as the reference does (F.silu rounds differently by an ulp,
    enough to flip fp16 roundings in the decoder tail)."""
    return x * torch.sigmoid(x)


def _vae_resnet(sd, pre, x):
    """ResnetBlock.forward with temb=None, dropout 0 (model.py:121-141); 1x1 nin_shortcut when the width changes."""
    h = _conv(sd, pre + "conv1", _swish(_vae_norm(sd, pre + "norm1", x)))
    h = _conv(sd, pre + "conv2", _swish(_vae_norm(sd, pre + "norm2", h)))
    if pre + "nin_shortcut.weight" in sd:
        x = _conv(sd, pre + "nin_shortcut", x, padding=0)
    return x + h


def _vae_attn(sd, pre, x):
    """AttnBlock.forward model.py:178-205: single-head attention over the h*w positions, scale c^-0.5."""
    h = _vae_norm(sd, pre + "norm", x)
    q, k, v = (_conv(sd, pre + n, h, padding=0) for n in ("q", "k", "v"))
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)            # (b, i, c)
    k = k.reshape(b, c, hh * ww)                             # (b, c, j)
    w = torch.softmax(torch.bmm(q, k) * (int(c) ** -0.5), dim=2)
    o = torch.bmm(v.reshape(b, c, hh * ww), w.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(sd, pre + "proj_out", o, padding=0)


def vae_decoder_layout(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2):
    """[(level, [(block index, cin, cout)], has_upsample)] in execution order (Decoder.__init__ model.py:473-525)."""
    block_in = ch * ch_mult[-1]
    out = []
    for lvl in reversed(range(len(ch_mult))):
        blocks = []
        for i in range(num_res_blocks + 1):
            blocks.append((i, block_in, ch * ch_mult[lvl]))
            block_in = ch * ch_mult[lvl]
        out.append((lvl, blocks, lvl != 0))
    return out


def vae_decode(sd, pre, z, ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2):
    """AutoencoderKL.decode autoencoder.py:331-334 -> Decoder.forward model.py:535-577 (attn_resolutions=[], tanh_out False).

    The tail of the reference is `h_fake = norm_out(h).half(); h = (h-mean)/std; h = h + (h_fake-h).detach()`
    (model.py:564-570): in the forward pass that is the GroupNorm output ROUNDED TO fp16, re-expressed in fp32 through the
    instance-normalised tensor (the two roundings of the add are kept)."""
    d = pre + "decoder."
    h = _conv(sd, pre + "post_quant_conv", z, padding=0)
    h = _conv(sd, d + "conv_in", h)
    h = _vae_resnet(sd, d + "mid.block_1.", h)
    h = _vae_attn(sd, d + "mid.attn_1.", h)
    h = _vae_resnet(sd, d + "mid.block_2.", h)
    for lvl, blocks, up in vae_decoder_layout(ch, ch_mult, num_res_blocks):
        for i, _, _ in blocks:
            h = _vae_resnet(sd, f"{d}up.{lvl}.block.{i}.", h)
        if up:
            h = _conv(sd, f"{d}up.{lvl}.upsample.conv", F.interpolate(h, scale_factor=2.0, mode="nearest"))
    h_fake = _vae_norm(sd, d + "norm_out", h).to(torch.float16)
    hn = (h - h.mean([2, 3], keepdim=True)) / h.std([2, 3], keepdim=True)
    h = hn + (h_fake - hn)
    return _conv(sd, d + "conv_out", _swish(h))


def viewfusion_decode(sd, z, z_scale_factor=0.18215, **kw):
    """ViewFusion.decode viewfusion_zero_depth_rgb.py:161-163: unnormalize(vae.decode(z / scale)).clip(0, 1)."""
    return ((vae_decode(sd, "vae.", z * 1 / z_scale_factor, **kw) + 1.0) / 2.0).clip(0.0, 1.0)


# ---------------------------------------------------------------------------------------------
# VAE encode (the VAE half): autoencoder.py:325-329, model.py:60-79,368-459
# ---------------------------------------------------------------------------------------------


def vae_encoder_layout(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2):
    """[(level, [(block index, cin, cout)], has_downsample)] in execution order (Encoder.__init__ model.py:391-412)."""
    in_mult = (1,) + tuple(ch_mult)
    out = []
    for lvl in range(len(ch_mult)):
        cin, blocks = ch * in_mult[lvl], []
        for i in range(num_res_blocks):
            blocks.append((i, cin, ch * ch_mult[lvl]))
            cin = ch * ch_mult[lvl]
        out.append((lvl, blocks, lvl != len(ch_mult) - 1))
    return out


def vae_encode_moments(sd, pre, x, ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2):
    """Encoder.forward model.py:434-459 + quant_conv (autoencoder.py:326-327): (B,3,H,W) in [-1,1] -> moments (B,8,H/8,W/8).
    Downsample = F.pad(x, (0,1,0,1)) + conv3x3 stride 2 padding 0 (model.py:72-76)."""
    e = pre + "encoder."
    h = _conv(sd, e + "conv_in", x)
    for lvl, blocks, down in vae_encoder_layout(ch, ch_mult, num_res_blocks):
        for i, _, _ in blocks:
            h = _vae_resnet(sd, f"{e}down.{lvl}.block.{i}.", h)
        if down:
            h = _conv(sd, f"{e}down.{lvl}.downsample.conv", F.pad(h, (0, 1, 0, 1), mode="constant", value=0), stride=2, padding=0)
    h = _vae_resnet(sd, e + "mid.block_1.", h)
    h = _vae_attn(sd, e + "mid.attn_1.", h)
    h = _vae_resnet(sd, e + "mid.block_2.", h)
    h = _conv(sd, e + "conv_out", _swish(_vae_norm(sd, e + "norm_out", h)))
    return _conv(sd, pre + "quant_conv", h, padding=0)


def viewfusion_encode(sd, images, z_scale_factor=0.18215, **kw):
    """ViewFusion.encode viewfusion_zero_depth_rgb.py:158-159: vae.encode(normalize(x)).mode() * scale, with
    normalize = clip(2x-1, -1, 1) (utils/common_utils.py:60-64) and mode() = the mean half of the moments
    (distributions.py:27,61-62)."""
    m = vae_encode_moments(sd, "vae.", torch.clip(images * 2 - 1.0, -1.0, 1.0), **kw)
    return torch.chunk(m, 2, dim=1)[0] * z_scale_factor


# ---------------------------------------------------------------------------------------------
# CLIP image encoder: FrozenCLIPImageEmbedder, external/sd1/ldm/modules/encoders/modules.py:402-441.
# The vision transformer itself lives in OpenAI's `clip` package (clip/model.py, NOT in the reference tree, version unpinned:
# requirements.txt "git+https://github.com/openai/CLIP.git"): restated here from its published architecture -- PARITY UNPINNED for
# that part; the reference's own preprocessing / call order is pinned through oracle/make_golden.py (clip_l14).
# ---------------------------------------------------------------------------------------------
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def clip_preprocess(x, size=224):
    """FrozenCLIPImageEmbedder.preprocess encoders/modules.py:422-431 (kornia.geometry.resize bicubic, align_corners=True,
    antialias=False == F.interpolate; kornia.enhance.normalize == (x - mean) / std)."""
    x = F.interpolate(x, size=(size, size), mode="bicubic", align_corners=True)
    x = (x + 1.0) / 2.0
    mean = torch.tensor(CLIP_MEAN, dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD, dtype=x.dtype).view(1, 3, 1, 1)
    return (x - mean) / std


def clip_encode_image(sd, pre, x, heads):
    """clip.model.VisionTransformer.forward (published CLIP model.py): conv1 patchify -> [class | patches] + positional
    embedding -> ln_pre -> ResidualAttentionBlocks (x + MHA(ln_1 x); x + c_proj(QuickGELU(c_fc(ln_2 x)))) -> ln_post(class) @ proj."""
    w = sd[pre + "conv1.weight"]
    width, patch = w.shape[0], w.shape[-1]
    h = F.conv2d(x, w, None, stride=patch)                                   # (B, width, g, g)
    B = h.shape[0]
    h = h.reshape(B, width, -1).permute(0, 2, 1)                             # (B, g*g, width)
    cls = sd[pre + "class_embedding"].to(h.dtype) + torch.zeros(B, 1, width, dtype=h.dtype)
    h = torch.cat([cls, h], dim=1) + sd[pre + "positional_embedding"]
    h = F.layer_norm(h, (width,), sd[pre + "ln_pre.weight"], sd[pre + "ln_pre.bias"], 1e-5)
    i = 0
    dh = width // heads
    while f"{pre}transformer.resblocks.{i}.ln_1.weight" in sd:
        b = f"{pre}transformer.resblocks.{i}."
        y = F.layer_norm(h, (width,), sd[b + "ln_1.weight"], sd[b + "ln_1.bias"], 1e-5)
        qkv = F.linear(y, sd[b + "attn.in_proj_weight"], sd[b + "attn.in_proj_bias"])
        q, k, v = (t.reshape(B, -1, heads, dh).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
        a = torch.softmax((q * dh ** -0.5) @ k.transpose(-2, -1), dim=-1) @ v     # nn.MultiheadAttention scaling
        a = a.transpose(1, 2).reshape(B, -1, width)
        h = h + F.linear(a, sd[b + "attn.out_proj.weight"], sd[b + "attn.out_proj.bias"])
        y = F.layer_norm(h, (width,), sd[b + "ln_2.weight"], sd[b + "ln_2.bias"], 1e-5)
        y = F.linear(y, sd[b + "mlp.c_fc.weight"], sd[b + "mlp.c_fc.bias"])
        y = y * torch.sigmoid(1.702 * y)                                      # QuickGELU
        h = h + F.linear(y, sd[b + "mlp.c_proj.weight"], sd[b + "mlp.c_proj.bias"])
        i += 1
    cls = F.layer_norm(h[:, 0, :], (width,), sd[pre + "ln_post.weight"], sd[pre + "ln_post.bias"], 1e-5)
    return cls @ sd[pre + "proj"]


def clip_image_embed(sd, images_pm1, heads=16, pre="clip_image_encoder.model.visual."):
    """FrozenCLIPImageEmbedder.encode encoders/modules.py:433-441: (B,3,H,W) in [-1,1] -> (B,1,768)."""
    return clip_encode_image(sd, pre, clip_preprocess(images_pm1), heads).float().unsqueeze(1)
