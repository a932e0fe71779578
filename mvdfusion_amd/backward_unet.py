"""Backward walk over the whole view-conditioned UNet on the HIP path (reference: mvdfusion/unet.py:524-556 forward;
train.py:90-95 `loss.backward()`).

The forward (UNetModel.run with ``_record``) keeps a copy of every block's INPUT (activation checkpointing at block granularity,
as the reference's use_checkpoint=True does); the walk below visits the blocks in reverse, recomputes the layer inputs inside a
block on the inference path, and chains the block backwards of backward_blocks.py:

    head <- output_blocks[11..0] (cat = [h | skip]: the skip half of the gradient is parked until the matching input block)
         <- middle_block <- input_blocks[n..1] (+ parked skip gradient) <- stem conv (weight gradient only).

Strided and upsampling convolutions reuse the stride-1 backward: a stride-2 conv's dgrad / wgrad equal the stride-1 ones on the
zero-stuffed output gradient (Z[2i, 2j] = dY[i, j]); nearest-2x upsampling + conv is the stride-1 backward on the explicitly
upsampled input followed by a 2x2 block sum.  The per-step vectors (time embedding MLP, cc_projection) are a few (1 x 128) / (V x 768)
products: host glue in torch.  The gradient w.r.t. the volume features (the output of GridAttn) is returned: GridAttn's own
backward continues from it (backward_gridattn.py).
"""
import torch
import torch.nn.functional as F

from . import backward as bw
from . import backward_blocks as bb
from .attention import SpatialTransformer, ViewAlignedFeatureTransformer
from .unet import Downsample, ResBlock, Upsample, _StemConv


def _silu_grad(z):
    s = torch.sigmoid(z)
    return s * (1 + z * (1 - s))


def downsample_backward(tape, layer, x, dout, B, H, W):
    """Downsample = conv3x3 stride 2 pad 1 (openaimodel.py:133-151).  x (B*H*W, C) input, dout (B*H/2*W/2, Cout)."""
    Co = dout.shape[-1]
    z = torch.zeros(B, H, W, Co, dtype=torch.float32, device=dout.device)
    z[:, ::2, ::2] = dout.view(B, H // 2, W // 2, Co)                          # zero-stuffed gradient on the input grid
    dx, dW, db = tape.conv_bwd(tape.planes(x), layer.op.weight, z.view(B * H * W, Co), B, H, W)
    return dx, {"op.weight": dW, "op.bias": db}


def upsample_backward(tape, layer, x, dout, B, H, W):
    """Upsample = nearest 2x + conv3x3 (openaimodel.py:100-131).  x (B*H*W, C) input, dout (B*2H*2W, Cout)."""
    C = x.shape[-1]
    xu = x.view(B, H, W, C).repeat_interleave(2, 1).repeat_interleave(2, 2).reshape(B * 4 * H * W, C)
    du, dW, db = tape.conv_bwd(tape.planes(xu), layer.conv.weight, dout, B, 2 * H, 2 * W)
    dx = du.reshape(B, H, 2, W, 2, C).sum((2, 4)).reshape(B * H * W, C)
    return dx, {"conv.weight": dW, "conv.bias": db}


def _pooled_volumes(vol, B, S, D, levels):
    """fp32 volume features per pyramid level (UNetWrapper.get_volume_feats_pyramid, unet.py:198-209: area interpolation by an integer
    factor == average pooling): {image size: (B*h*w*D, 768)}."""
    out = {S: vol.reshape(B * S * S * D, -1)}
    v = vol.view(B, S, S, D, -1).permute(0, 3, 4, 1, 2).reshape(B * D, -1, S, S)          # (B*D, 768, S, S)
    for f in levels:
        p = F.avg_pool2d(v, f)                                                       # host glue (the forward used mvd_area_pool)
        h = S // f
        out[h] = p.view(B, D, -1, h, h).permute(0, 3, 4, 1, 2).reshape(B * h * h * D, -1).contiguous()
    return out


def unet_backward(unet, ctx, tape, record, dh, B, S, D, emb, t_sin, context, vol):
    """dh: gradient at the input of the output head (B*S*S, mc).  record: UNetModel._record of the forward.  emb (1, 4 mc) time
    embedding, t_sin (1, mc) its sinusoid input, context (B, 768), vol (B, S, S, D, 768) fp32 volume features (after dropout).
    Returns ({UNetModel-relative parameter name: gradient}, dcontext (B, 768), dvol (B, S, S, D, 768))."""
    grads = {}
    n_in = len(unet.input_blocks)
    names = {}
    for i, blk in enumerate(unet.input_blocks):
        names[id(blk)] = f"input_blocks.{i}."
    names[id(unet.middle_block)] = "middle_block."
    for i, blk in enumerate(unet.output_blocks):
        names[id(blk)] = f"output_blocks.{i}."
    vols = _pooled_volumes(vol, B, S, D, [2, 4, 8])
    dvols = {}
    demb = torch.zeros_like(emb)
    dcontext = torch.zeros_like(context)
    skip_grad = {}
    out_index = {id(blk): i for i, blk in enumerate(unet.output_blocks)}
    d = dh
    for blk, x_in, H, W, ca in reversed(record):
        pre = names[id(blk)]
        if isinstance(blk[0], _StemConv):
            d = d + skip_grad.pop(0)
            _, dW, db = tape.conv_bwd(x_in, blk[0].weight, d, B, H, W, need_dx=False)
            grads[pre + "0.weight"], grads[pre + "0.bias"] = dW, db
            continue
        is_input = pre.startswith("input_blocks.")
        if is_input:
            d = d + skip_grad.pop(int(pre.split(".")[1]))
        # layer inputs of this block, recomputed on the inference path
        layers = list(blk)
        xs, hw = [x_in], [(H, W)]
        for li, layer in enumerate(layers[:-1]):
            h_, w_ = hw[-1]
            if isinstance(layer, ResBlock):
                y = layer.run(ctx, xs[-1], h_, w_)
            else:
                y = layer.run(ctx, xs[-1], h_, w_)
            assert not isinstance(layer, (Upsample, Downsample)), "resampling layers close a block"
            xs.append(y.clone())
            hw.append((h_, w_))
        for li in range(len(layers) - 1, -1, -1):
            layer, x, (h_, w_) = layers[li], xs[li], hw[li]
            lp = f"{pre}{li}."
            if isinstance(layer, ResBlock):
                d, g, de = bb.resblock_backward(tape, layer, x, emb, d, B, h_, w_)
                demb += de
            elif isinstance(layer, SpatialTransformer):
                d, g, dc = bb.spatial_transformer_backward(tape, layer, x, context, d, B, h_, w_)
                dcontext += dc
            elif isinstance(layer, ViewAlignedFeatureTransformer):
                d, g, dv = bb.view_aligned_transformer_backward(tape, layer, x, vols[h_], d, B, h_, w_, D)
                dvols[h_] = dvols[h_] + dv if h_ in dvols else dv
            elif isinstance(layer, Upsample):
                d, g = upsample_backward(tape, layer, x, d, B, h_, w_)
            elif isinstance(layer, Downsample):
                d, g = downsample_backward(tape, layer, x, d, B, h_, w_)
            else:
                raise TypeError(type(layer))
            grads.update({lp + k: v for k, v in g.items()})
        if id(blk) in out_index:                       # d is the gradient of cat = [h | skip]
            skip_grad[n_in - 1 - out_index[id(blk)]] = d[:, ca:].contiguous()
            d = d[:, :ca].contiguous()
    assert not skip_grad
    # ---- time embedding MLP (unet.py:537-538; openaimodel time_embed): emb = L2(silu(L1(t_sin)))      host glue, (1 x 4 mc)
    l1, l2 = unet.time_embed[0], unet.time_embed[2]
    z1 = t_sin @ l1.weight.t() + l1.bias
    e1 = F.silu(z1)
    grads["time_embed.2.weight"], grads["time_embed.2.bias"] = demb.t() @ e1, demb[0].clone()
    dz1 = (demb @ l2.weight) * _silu_grad(z1)
    grads["time_embed.0.weight"], grads["time_embed.0.bias"] = dz1.t() @ t_sin, dz1[0].clone()
    # ---- volume pyramid: area pooling backward (each fine cell receives 1 / f^2 of its coarse cell's gradient)
    dvol = torch.zeros(B, S, S, D, vol.shape[-1], dtype=torch.float32, device=vol.device)
    for h_, dv in dvols.items():
        f = S // h_
        t = dv.view(B, h_, h_, D, -1)
        if f > 1:
            t = t.repeat_interleave(f, 1).repeat_interleave(f, 2) / float(f * f)
        dvol += t
    return grads, dcontext, dvol


def cc_projection_backward(proj, clip_v_embed, dcontext):
    """ViewFusion.cc_projection = Linear(796,768) / SiLU / Linear / SiLU / Linear (viewfusion_zero_depth_rgb.py:96-104) on the (V, 796)
    pose-augmented CLIP embedding.  Host glue: three (V x 768) products.  Returns {relative name: gradient}."""
    l1, l2, l3 = proj[0], proj[2], proj[4]
    z1 = clip_v_embed @ l1.weight.t() + l1.bias
    a1 = F.silu(z1)
    z2 = a1 @ l2.weight.t() + l2.bias
    a2 = F.silu(z2)
    g = {"4.weight": dcontext.t() @ a2, "4.bias": dcontext.sum(0)}
    dz2 = (dcontext @ l3.weight) * _silu_grad(z2)
    g["2.weight"], g["2.bias"] = dz2.t() @ a1, dz2.sum(0)
    dz1 = (dz2 @ l2.weight) * _silu_grad(z1)
    g["0.weight"], g["0.bias"] = dz1.t() @ clip_v_embed, dz1.sum(0)
    return g
