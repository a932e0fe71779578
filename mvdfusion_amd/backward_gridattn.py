"""Backward of GridAttn (mvdfusion/view_attn_efficient2.py:269-442) on the HIP path: from the gradient of the feature frustum
(V, S, S, D, 768) -- what the view-aligned transformers of the UNet hand back (backward_unet.py) -- to every `view_attn.*`
parameter and to the conditioning vector c (ViewFusion.time_embed).

Recompute-then-backward like the UNet blocks: the UNFUSED forward chain (token kernel, GEMMs, view attention, pooling) is re-run
keeping its intermediates, then

  final_layer_b  <- softmax-over-V pooling (weight_layer)  <- 3 x DiTBlock (adaLN-Zero: LayerNorm without affine + modulate, timm
  attention over the V reference views, GELU MLP, gates)   <- pre_layer_b (Linear 723 -> 256 + GELU)  <- token matrix
  <- bilinear gathers of the z-embedded latents (grid_sample backward = mvd_gridattn_tokens_backward)  <- z_embedder.

Matrix products run on the split-operand MFMA GEMM (backward.linear_backward), attention over V on mvd_attention_backward with
sequences of length V, LayerNorm+modulate on mvd_layernorm_backward with weight (1 + scale); elementwise glue (GELU', gates, the
(nseq, V, C) pooling algebra, the 5-channel z-embedding) is torch.
"""
import math

import torch
import torch.nn.functional as F

from . import backward as bw
from . import hip


def _gelu_grad(z):
    return 0.5 * (1.0 + torch.erf(z * 0.7071067811865476)) + z * torch.exp(-0.5 * z * z) * 0.3989422804014327


def _silu_grad(z):
    s = torch.sigmoid(z)
    return s * (1 + z * (1 - s))


def _dit_block_backward(tape, blk, h, c, dh2, T, V):
    """One DiTBlock (view_attn_efficient2.py:42-67).  h (T, C) block input, c (1, C) conditioning, dh2 gradient at the block output.
    Returns (dh, {name: grad}, dc (1, C))."""
    C, H = blk.hidden_size, blk.num_heads
    dh_ = C // H
    lin = blk.adaLN_modulation[1]
    sc_ = F.silu(c)
    mod = sc_ @ lin.weight.t() + lin.bias                                    # (1, 6C)   host glue
    sh1, s1, g1, sh2, s2, g2 = (mod[0, i * C:(i + 1) * C].contiguous() for i in range(6))
    dev = h.device
    # ---- forward (unfused)
    m1 = hip.planes_like(T, C, dev)
    hip.layernorm(h, m1, s1, sh1, T, C, eps=1e-6, w_plus_one=True)
    qkv = tape.linear(m1, blk.attn.qkv.weight, blk.attn.qkv.bias)           # (T, 3C): [q | k | v], each [heads][dhead]
    att = hip.planes_like(T, C, dev)
    hip.check(hip.lib().mvd_view_mha(hip.ptr(qkv), hip.ptr(att), T // V, V, H, dh_, hip.stream()))
    a_out = tape.linear(att, blk.attn.proj.weight, blk.attn.proj.bias)
    h1 = h + g1 * a_out
    m2 = hip.planes_like(T, C, dev)
    hip.layernorm(h1, m2, s2, sh2, T, C, eps=1e-6, w_plus_one=True)
    f1 = tape.linear(m2, blk.mlp.fc1.weight, blk.mlp.fc1.bias)              # pre-activation (T, hidden)
    gel, _ = bw.act_planes(f1, hip.ACT_GELU)                               # GELU + operand split in one pass (mvd_act_planes)
    f2 = tape.linear(gel, blk.mlp.fc2.weight, blk.mlp.fc2.bias)
    # ---- backward
    g = {}
    dg2 = bw.col_sum((dh2 * f2).contiguous(), T, C)
    dgel, g["mlp.fc2.weight"], g["mlp.fc2.bias"] = tape.linear_bwd(gel, blk.mlp.fc2.weight, dh2 * g2)
    dm2, g["mlp.fc1.weight"], g["mlp.fc1.bias"] = tape.linear_bwd(m2, blk.mlp.fc1.weight, bw.act_backward(dgel, f1, hip.ACT_GELU))
    dx, ds2, dsh2 = bw.layernorm_backward(h1.contiguous(), dm2.contiguous(), (1.0 + s2).contiguous(), 1e-6)
    dh1 = dh2 + dx
    dg1 = bw.col_sum((dh1 * a_out).contiguous(), T, C)
    datt, g["attn.proj.weight"], g["attn.proj.bias"] = tape.linear_bwd(att, blk.attn.proj.weight, dh1 * g1)
    q, k, v = (qkv[:, i * C:(i + 1) * C].contiguous() for i in range(3))
    dq, dk, dv = bw.attention_backward(q, k, v, datt.contiguous(), T // V, H, V, dh_)
    dm1, g["attn.qkv.weight"], g["attn.qkv.bias"] = tape.linear_bwd(m1, blk.attn.qkv.weight, torch.cat([dq, dk, dv], dim=1))
    dx, ds1, dsh1 = bw.layernorm_backward(h.contiguous(), dm1.contiguous(), (1.0 + s1).contiguous(), 1e-6)
    dh = dh1 + dx
    dmod = torch.cat([dsh1, ds1, dg1, dsh2, ds2, dg2])[None, :]               # (1, 6C)
    g["adaLN_modulation.1.weight"] = dmod.t() @ sc_
    g["adaLN_modulation.1.bias"] = dmod[0].clone()
    dc = (dmod @ lin.weight) * _silu_grad(c)
    return dh, g, dc


def gridattn_backward(ga, tape, eng, c, dvol, V, S, D):
    """ga: GridAttn; eng: the StepEngine whose buffers hold this step's inputs (x, depth noise, step table, cameras, input latents);
    c (1, 256) conditioning; dvol (V*S*S*D, 768) gradient of the frustum.  Returns ({view_attn-relative name: grad}, dc (1, 256))."""
    L = hip.lib()
    dev = dvol.device
    C = ga.hidden_size
    nseq = V * S * S * D
    T = nseq * V
    agg = ga.aggregation_transformer
    # ---- forward (unfused chain, view_attn_efficient2.py GridAttn.run)
    z = ga.z_embedder[0]
    feat = torch.empty(V, S, S, 256, dtype=torch.float32, device=dev)
    in_feat = torch.empty(1, S, S, 256, dtype=torch.float32, device=dev)
    hip.check(L.mvd_zembed(hip.ptr(eng.x), hip.ptr(z.weight), hip.ptr(z.bias), hip.ptr(feat), V, S, hip.stream()))
    hip.check(L.mvd_zembed(hip.ptr(eng.input_latents), hip.ptr(z.weight), hip.ptr(z.bias), hip.ptr(in_feat), 1, S, hip.stream()))
    half = 1.0 / float(S)
    grid_lin = torch.linspace(1.0 - half, -1.0 + half, S, dtype=torch.float32).to(dev)
    tokens = hip.planes_like(T, hip.TOKEN_LD, dev)
    dsrc, dsteps = eng.depth_geo()          # (the depth source the forward used: x itself, or an overwrite_attn_depth map)
    geo = (hip.ptr(dsrc), hip.ptr(eng.depth_noise), hip.ptr(dsteps), hip.ptr(eng.iter), hip.ptr(grid_lin))
    hip.check(L.mvd_gridattn_tokens(*geo, hip.ptr(feat), hip.ptr(in_feat), hip.ptr(eng.cams), hip.ptr(eng.in_cam), hip.ptr(tokens), V, 0, V,
                                    S, D, float(ga.depth_scale), float(ga.depth_shift), hip.stream()))
    pre = ga.pre_layer_b[0]
    z0 = tape.linear(tokens, pre.weight, pre.bias)                             # (T, 256) pre-activation
    hs = [bw.act_planes(z0, hip.ACT_GELU, planes=False, f32=True)[1]]
    for blk in agg.layer_list:
        hcur = hs[-1]
        # block output via the inference path's own kernels (DiTBlock.run mutates its buffers: use the unfused algebra here)
        hs.append(_dit_forward(tape, blk, hcur, c, T, V))
    hL = hs[-1].view(nseq, V, C)
    wl = agg.weight_layer
    lg = hL @ wl.weight[0] + wl.bias                                           # (nseq, V)        host glue
    p = torch.softmax(lg, dim=1)
    pooled = (p[:, :, None] * hL).sum(1)                                       # (nseq, C)
    poolp = tape.planes(pooled)
    # ---- backward
    g = {}
    fin = ga.final_layer_b
    dpool, g["final_layer_b.weight"], g["final_layer_b.bias"] = tape.linear_bwd(poolp, fin.weight, dvol)
    a = (dpool[:, None, :] * hL).sum(2)                                        # (nseq, V): d pooled . h_v
    dlg = p * (a - (p * a).sum(1, keepdim=True))
    dh = (p[:, :, None] * dpool[:, None, :] + dlg[:, :, None] * wl.weight[0]).reshape(T, C)
    g["aggregation_transformer.weight_layer.weight"] = (dlg[:, :, None] * hL).sum((0, 1))[None, :]
    g["aggregation_transformer.weight_layer.bias"] = dlg.sum().reshape(1)
    dc = torch.zeros_like(c)
    for bi in range(len(agg.layer_list) - 1, -1, -1):
        dh, gb, dcb = _dit_block_backward(tape, agg.layer_list[bi], hs[bi], c, dh.contiguous(), T, V)
        g.update({f"aggregation_transformer.layer_list.{bi}.{k}": v for k, v in gb.items()})
        dc += dcb
    dtok, g["pre_layer_b.0.weight"], g["pre_layer_b.0.bias"] = tape.linear_bwd(tokens, pre.weight, bw.act_backward(dh, z0, hip.ACT_GELU))
    # ---- grid_sample backward into the z-embedded feature maps, then the 5 -> 256 z-embedding
    dtok = dtok.contiguous() if dtok.is_contiguous() else dtok
    base = dtok if dtok.storage_offset() == 0 else dtok.contiguous()
    ldt = base.stride(0)
    mx = float(dtok[:, :512].abs().max())
    scale = 2.0 ** (40 - math.floor(math.log2(mx))) if mx > 0 and math.isfinite(mx) else 1.0
    dfeat_acc = torch.zeros(V, S, S, 256, dtype=torch.int64, device=dev)
    din_acc = torch.zeros(1, S, S, 256, dtype=torch.int64, device=dev)
    hip.check(L.mvd_gridattn_tokens_backward(*geo, hip.ptr(eng.cams), hip.ptr(eng.in_cam), hip.ptr(base), ldt, hip.ptr(dfeat_acc),
                                             hip.ptr(din_acc), float(scale), V, 0, V, S, D, float(ga.depth_scale), float(ga.depth_shift),
                                             hip.stream()))
    dW = torch.zeros_like(z.weight)
    db = torch.zeros_like(z.bias)
    for lat, acc, n in ((eng.x, dfeat_acc, V), (eng.input_latents, din_acc, 1)):
        xp = lat.reshape(n, 5, S * S).permute(0, 2, 1).reshape(n * S * S, 5)
        zz = xp @ z.weight.t() + z.bias
        dz = (acc.double() / scale).float().reshape(n * S * S, 256) * _gelu_grad(zz)
        dW += dz.t() @ xp
        db += dz.sum(0)
    g["z_embedder.0.weight"], g["z_embedder.0.bias"] = dW, db
    return g, dc


def _dit_forward(tape, blk, h, c, T, V):
    """Unfused DiTBlock forward returning the block output (same algebra as _dit_block_backward's forward half)."""
    C, H = blk.hidden_size, blk.num_heads
    lin = blk.adaLN_modulation[1]
    mod = F.silu(c) @ lin.weight.t() + lin.bias
    sh1, s1, g1, sh2, s2, g2 = (mod[0, i * C:(i + 1) * C].contiguous() for i in range(6))
    dev = h.device
    m1 = hip.planes_like(T, C, dev)
    hip.layernorm(h, m1, s1, sh1, T, C, eps=1e-6, w_plus_one=True)
    qkv = tape.linear(m1, blk.attn.qkv.weight, blk.attn.qkv.bias)
    att = hip.planes_like(T, C, dev)
    hip.check(hip.lib().mvd_view_mha(hip.ptr(qkv), hip.ptr(att), T // V, V, H, C // H, hip.stream()))
    h1 = tape.linear(att, blk.attn.proj.weight, blk.attn.proj.bias, res=h, colscale=g1)        # h + g1 * proj(att): the GEMM's own epilogue
    m2 = hip.planes_like(T, C, dev)
    hip.layernorm(h1, m2, s2, sh2, T, C, eps=1e-6, w_plus_one=True)
    f1 = tape.linear(m2, blk.mlp.fc1.weight, blk.mlp.fc1.bias)
    return tape.linear(bw.act_planes(f1, hip.ACT_GELU)[0], blk.mlp.fc2.weight, blk.mlp.fc2.bias, res=h1, colscale=g2)


def time_embed_backward(time_embed, t_sin, dc):
    """ViewFusion.time_embed = Linear(256,256) / SiLU / Linear(256,256) on the sinusoid of t; only row 0 feeds GridAttn
    (viewfusion_zero_depth_rgb.py:276-279, 303).  Host glue.  Returns {relative name: grad}."""
    l1, l2 = time_embed[0], time_embed[2]
    z1 = t_sin @ l1.weight.t() + l1.bias
    e1 = F.silu(z1)
    g = {"2.weight": dc.t() @ e1, "2.bias": dc[0].clone()}
    dz1 = (dc @ l2.weight) * _silu_grad(z1)
    g["0.weight"], g["0.bias"] = dz1.t() @ t_sin, dz1[0].clone()
    return g
