"""View-conditioned SD1.x UNet executed by HIP kernels -- mirror of ``mvdfusion/unet.py``.

Classes keep the reference's names, constructor kwargs and state_dict keys:
  ``ResBlock`` / ``Downsample`` / ``Upsample``   external/sd1/ldm/modules/diffusionmodules/openaimodel.py:91-275
  ``TimestepEmbedSequential``                    mvdfusion/unet.py:36-52
  ``UNetModel``                                  mvdfusion/unet.py:215-576
  ``UNetWrapper``                                mvdfusion/unet.py:56-209

Data layout in HBM: activations are fp32 channels-last, (B, H, W, C) == a (B*H*W, C) row-major matrix, so the conv
path (implicit GEMM, K = 9*C contiguous per tap) and the transformer path ((hw, C) tokens) share buffers with no
transposes.  The classifier-free-guidance pair is ONE batch of 2V views (rows [0,V) conditional, [V,2V) null), so the
4 GB of weights stream once per step instead of twice.
"""
import math

import torch
import torch.nn as nn

from . import hip
from .attention import SpatialTransformer, ViewAlignedFeatureTransformer
from .engine import Ctx


class GroupNorm32(nn.GroupNorm):
    pass


def normalization(channels):
    return GroupNorm32(32, channels)  # eps 1e-5 (diffusionmodules/util.py:200-217)


class TimestepBlock(nn.Module):
    pass


class Upsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels or channels
        self.conv = nn.Conv2d(self.channels, self.out_channels, 3, padding=padding)
        self._p = None

    def run(self, ctx, x, H, W, out=None):
        if self._p is None:
            self._p = hip.pack_conv3x3(self.conv.weight, self.conv.bias)
        if out is None:
            out = ctx.act((ctx.B * 4 * H * W, self.out_channels))
        xp = hip.split_planes(x, ctx.ws.planes("updown.x", ctx.B * H * W, self.channels))   # raw residual stream -> planes
        ctx.gemm(xp, self._p, out, conv=dict(B=ctx.B, Hin=H, Win=W, Cin=self.channels, Hout=2 * H, Wout=2 * W,
                                            stride=1, upsample=1), gn=(ctx.B, 4 * H * W), kind="conv")
        return out, 2 * H, 2 * W


class Downsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels or channels
        self.op = nn.Conv2d(self.channels, self.out_channels, 3, stride=2, padding=padding)
        self._p = None

    def run(self, ctx, x, H, W, out=None):
        if self._p is None:
            self._p = hip.pack_conv3x3(self.op.weight, self.op.bias)
        Ho, Wo = H // 2, W // 2
        if out is None:
            out = ctx.act((ctx.B * Ho * Wo, self.out_channels))
        xp = hip.split_planes(x, ctx.ws.planes("updown.x", ctx.B * H * W, self.channels))
        ctx.gemm(xp, self._p, out, conv=dict(B=ctx.B, Hin=H, Win=W, Cin=self.channels, Hout=Ho, Wout=Wo, stride=2,
                                            upsample=0), gn=(ctx.B, Ho * Wo), kind="conv")
        return out, Ho, Wo


class ResBlock(TimestepBlock):
    """GN+SiLU -> conv3x3 (+bias +time-embedding) -> GN+SiLU -> conv3x3 (+bias +skip)  (openaimodel.py:255-275)."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, dims=2, use_checkpoint=False,
                 use_scale_shift_norm=False):
        super().__init__()
        assert not use_scale_shift_norm
        self.channels, self.out_channels = channels, out_channels or channels
        self.in_layers = nn.Sequential(normalization(channels), nn.SiLU(),
                                       nn.Conv2d(channels, self.out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(normalization(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        nn.Conv2d(self.out_channels, self.out_channels, 3, padding=1))
        self.skip_connection = nn.Identity() if self.out_channels == channels else \
            nn.Conv2d(channels, self.out_channels, 1)
        self._p = None

    def packed(self):
        if self._p is None:
            c1, c2 = self.in_layers[2], self.out_layers[3]
            sk = None if isinstance(self.skip_connection, nn.Identity) else \
                hip.pack_linear(self.skip_connection.weight, self.skip_connection.bias)
            self._p = (hip.pack_conv3x3(c1.weight, None), hip.pack_conv3x3(c2.weight, c2.bias), sk)
        return self._p

    def run(self, ctx, x, H, W, out=None, x_planes=None):
        """x: fp32 (M, Cin) residual stream; x_planes: its split-bf16 planes if the producer already wrote them."""
        B, Ci, Co = ctx.B, self.channels, self.out_channels
        M = B * H * W
        w1, w2, wsk = self.packed()
        geo = dict(B=B, Hin=H, Win=W, Hout=H, Wout=W, stride=1, upsample=0)
        a = ctx.ws.planes("res.a", M, Ci)
        ctx.groupnorm(x, a, self.in_layers[0], B, H * W, Ci, silu=True)
        h = ctx.ws.get("res.h", (M, Co))
        # conv bias + Linear(SiLU(emb)) are folded into one per-step bias vector (UNetModel._time_biases)
        a2 = ctx.ws.planes("res.a2", M, Co)
        # GroupNorm + SiLU of h right behind the convolution (inside its split-K reduce when it splits); h itself has no other reader
        ctx.gemm(a, w1, h, conv=dict(Cin=Ci, **geo), bias=False, bias_b=ctx.emb_bias[self], rows_per_batch=M, gn=(B, H * W), kind="conv",
                 gn_apply=(self.out_layers[0], a2, True, True))
        skip = x
        if wsk is not None:
            if x_planes is None:
                x_planes = hip.split_planes(x, ctx.ws.planes("res.xp", M, Ci))
            skip = ctx.ws.get("res.skip", (M, Co))
            ctx.gemm(x_planes, wsk, skip, kind="skip")
        if out is None:
            out = ctx.act((M, Co))
        ctx.gemm(a2, w2, out, conv=dict(Cin=Co, **geo), res=skip, gn=(B, H * W), kind="conv")
        return out


def gn_hint(layer):
    """(GroupNorm module, name of its planes buffer, silu) of the GroupNorm a layer starts with -- what the PREVIOUS layer's output GEMM
    needs to apply it on the way out (Ctx.next_gn); None for layers that do not start with one."""
    from .attention import SpatialTransformer, ViewAlignedFeatureTransformer
    if isinstance(layer, tuple):                      # an explicit hint (the UNet head's GroupNorm)
        return layer
    if isinstance(layer, ResBlock):
        return layer.in_layers[0], "res.a", True
    if isinstance(layer, SpatialTransformer):
        return layer.norm, "tf.n", False
    if isinstance(layer, ViewAlignedFeatureTransformer):
        return layer.aligned_attn_norm, "tf.n", False
    return None


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    def run(self, ctx, x, H, W, out=None, x_planes=None, next_layer=None, next_cat=None):
        """Run the layers in order; `out` (optional) is the buffer the LAST layer must write (skip tensors);
        `x_planes`: split-bf16 planes of x when the caller has them (the concat kernel writes both); next_layer: the layer that
        consumes this block's output directly (its leading GroupNorm is applied by the last layer's output GEMM); next_cat: the block
        output goes into the decoder's next concat (Ctx.next_cat: the last layer's output GEMM produces that concat's planes)."""
        n = len(self)
        for i, layer in enumerate(self):
            dst = out if i == n - 1 else None
            ctx.next_gn = gn_hint(self[i + 1] if i + 1 < n else next_layer)
            ctx.next_cat = next_cat if i == n - 1 else None
            if isinstance(layer, (Upsample, Downsample)):
                x, H, W = layer.run(ctx, x, H, W, out=dst)
            elif isinstance(layer, ResBlock):
                x = layer.run(ctx, x, H, W, out=dst, x_planes=x_planes if i == 0 else None)
            else:
                x = layer.run(ctx, x, H, W, out=dst)
            ctx.next_gn = None
            ctx.next_cat = None
        return x, H, W


class _StemConv(nn.Conv2d):
    """input_blocks.0.0: Conv2d(in_channels -> model_channels); input is zero-padded to 32 channels."""
    _p = None

    def run(self, ctx, x, H, W, out):
        if self._p is None:
            self._p = hip.pack_conv3x3(self.weight, self.bias)
        ctx.gemm(x, self._p, out, conv=dict(B=ctx.B, Hin=H, Win=W, Cin=self._p.conv_cin, Hout=H, Wout=W, stride=1,
                                            upsample=0), gn=(ctx.B, H * W), kind="conv")
        return out


class UNetModel(nn.Module):
    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None,
                 use_checkpoint=False, use_fp16=False, num_heads=-1, num_head_channels=-1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False,
                 use_spatial_transformer=True, use_view_aligned_transformer=True, transformer_depth=1,
                 context_dim=None, n_embed=None, legacy=True):
        super().__init__()
        assert use_view_aligned_transformer and use_spatial_transformer and context_dim is not None
        assert dims == 2 and num_classes is None and not resblock_updown and n_embed is None and conv_resample
        assert num_heads != -1 and num_head_channels == -1, "this build supports the num_heads form (configs/*.yaml)"
        if isinstance(context_dim, (list, tuple)) or type(context_dim).__name__ == "ListConfig":
            context_dim = list(context_dim)[0]
        channel_mult = tuple(channel_mult)
        attention_resolutions = tuple(attention_resolutions)
        self.image_size, self.in_channels, self.model_channels = image_size, in_channels, model_channels
        self.out_channels, self.num_res_blocks, self.channel_mult = out_channels, num_res_blocks, channel_mult
        self.attention_resolutions, self.num_heads, self.context_dim = attention_resolutions, num_heads, context_dim
        ted = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))

        def res(cin, cout):
            return ResBlock(cin, ted, dropout, out_channels=cout, use_checkpoint=use_checkpoint)

        def st(c):
            return SpatialTransformer(c, num_heads, c // num_heads, depth=transformer_depth, context_dim=context_dim)

        def vaft(c):
            return ViewAlignedFeatureTransformer(c, num_heads, c // num_heads, depth=transformer_depth,
                                                 context_dim=context_dim, image_size=image_size)

        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(_StemConv(in_channels, model_channels, 3, padding=1))])
        skip_chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [res(ch, mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(st(ch))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                skip_chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, True, out_channels=ch)))
                skip_chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(res(ch, ch), st(ch), vaft(ch), res(ch, ch))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                layers = [res(ch + skip_chans.pop(), model_channels * mult)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers += [st(ch), vaft(ch)]
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch, True, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(normalization(ch), nn.SiLU(), nn.Conv2d(model_channels, out_channels, 3, padding=1))
        self._head = None
        self._temb = None
        self._xattn = None
        self._head_saved = None
        self._record = None

    # ------------------------------------------------------------------ reference API (unet.py:558-576)
    def get_cross_attn_parameters(self, finetune_cross_attn, finetune_view_attn):
        out = []
        for name, p in self.named_parameters():
            if finetune_cross_attn and any(k in name for k in (".norm.", ".proj_in.", ".transformer_blocks.", ".proj_out.")):
                out.append(p)
            if finetune_view_attn and ".aligned_attn_" in name:
                out.append(p)
        return out

    def disable_unet_grad(self):
        for name, p in self.named_parameters():
            if ".aligned_attn_" not in name:
                p.requires_grad_(False)

    # ------------------------------------------------------------------ HIP execution
    def _resblocks(self):
        return [m for m in self.modules() if isinstance(m, ResBlock)]

    def time_biases(self, ctx, t_sin):
        """emb = time_embed(t_sin); per ResBlock bias = in_layers.2.bias + emb_layers.1(SiLU(emb))  -- one GEMV
        over the row-concatenated emb_layers weights (unet.py:537-538; openaimodel.py:264-273)."""
        if self._temb is None:
            blocks = self._resblocks()
            w = torch.cat([b.emb_layers[1].weight.detach() for b in blocks], 0).contiguous()
            bias = torch.cat([b.emb_layers[1].bias.detach() + b.in_layers[2].bias.detach() for b in blocks], 0).contiguous()
            offs, o = {}, 0
            for b in blocks:
                offs[b] = (o, o + b.out_channels)
                o += b.out_channels
            self._temb = (w, bias, offs, torch.empty(1, o, dtype=torch.float32, device=w.device))
        w, bias, offs, out = self._temb
        ted = self.model_channels * 4
        e1 = ctx.ws.get("temb.e1", (1, ted))
        hip.gemv(self.time_embed[0].weight, self.time_embed[0].bias, t_sin, e1, act_out=hip.ACT_SILU)
        emb = ctx.ws.get("temb.emb", (1, ted))
        hip.gemv(self.time_embed[2].weight, self.time_embed[2].bias, e1, emb)
        hip.gemv(w, bias, emb, out, act_in=hip.ACT_SILU)
        ctx.emb_bias = {b: out[0, lo:hi] for b, (lo, hi) in offs.items()}

    def cross_attn_vectors(self, ctx):
        """attn2 of every SpatialTransformer sees the length-1 CLIP context: softmax over one key is 1, so its output is
        the per-view vector to_out(to_v(context)) (attention.py:170-193 with kv_len 1).  The two linear maps are composed
        once (W = Wo Wv in fp64, rounded to fp32 -- weight preprocessing like packing) and all 16 layers are evaluated
        by ONE GEMV per step over the row-concatenated W; each layer takes its column slice as the GEMM bias_b."""
        if self._xattn is None:
            sts = [m for m in self.modules() if isinstance(m, SpatialTransformer)]
            ws, bs, offs, o = [], [], {}, 0
            for st in sts:
                a2 = st.transformer_blocks[0].attn2
                wo, wv = a2.to_out[0].weight.detach().double(), a2.to_v.weight.detach().double()
                ws.append((wo @ wv).float())
                bs.append(a2.to_out[0].bias.detach().float())
                offs[st] = (o, o + st.in_channels)
                o += st.in_channels
            self._xattn = (torch.cat(ws, 0).contiguous(), torch.cat(bs, 0).contiguous(), offs, o)
        w, bias, offs, total = self._xattn
        out = ctx.ws.get("xattn.vec", (ctx.context.shape[0], total))
        ctx.gemv_rows(w, bias, ctx.context, out)
        ctx.xattn_vec = {st: out[:, lo:hi] for st, (lo, hi) in offs.items()}

    def run(self, ctx, x_in, t_sin, S):
        """x_in: split planes (B*S*S, 2*32) of the channels-last zero-padded input; t_sin: (1, model_channels) sinusoid; returns the
        (B*S*S, 8) head output (first out_channels columns valid)."""
        B = ctx.B
        self.time_biases(ctx, t_sin)
        self.cross_attn_vectors(ctx)
        hs = []
        H = W = S
        h = x_in
        rec = self._record          # training: a list that receives (block, copy of the block input, H, W, h width of a cat input)
        for bi, blk in enumerate(self.input_blocks):
            if rec is not None:
                rec.append((blk, h.clone(), H, W, 0))
            if bi == 0:
                o = ctx.ws.get("hs0", (B * H * W, self.model_channels))
                ctx.next_gn = gn_hint(self.input_blocks[1][0])
                h = blk[0].run(ctx, h, H, W, o)
                ctx.next_gn = None
            else:
                last = blk[-1]
                Ho, Wo = (H // 2, W // 2) if isinstance(last, Downsample) else (H, W)
                co = last.out_channels if isinstance(last, (ResBlock, Downsample)) else last.in_channels
                o = ctx.ws.get(f"hs{bi}", (B * Ho * Wo, co))      # skip tensors get their own static buffers
                nxt = self.input_blocks[bi + 1][0] if bi + 1 < len(self.input_blocks) else self.middle_block[0]
                h, H, W = blk.run(ctx, h, H, W, out=o, next_layer=nxt)
            hs.append((h, H, W))
        if rec is not None:
            rec.append((self.middle_block, h.clone(), H, W, 0))
        keep = rec is not None or ctx.keep_fp32               # a backward will read the blocks' fp32 inputs

        def cat_hint(nblk, co, Hn, Wn):
            """Ctx.next_cat for the concat in front of decoder block `nblk`, whose first operand (co channels at Hn x Wn) the current
            block is about to produce: (skip, cat buffer, raw planes buffer, norm, planes name, silu); None where the concat kernel stays."""
            first = nblk[0]
            if keep or not isinstance(first, ResBlock) or isinstance(first.skip_connection, nn.Identity):
                return None                                  # (the fp32 concatenation has a reader there)
            nsk = hs[-1][0]
            Mn = B * Hn * Wn
            return (nsk, ctx.ws.get("cat", (Mn, co + nsk.shape[-1])), ctx.ws.planes("catp", Mn, co + nsk.shape[-1])) + gn_hint(first)

        def out_geometry(blk, H, W):
            last = blk[-1]
            co = last.out_channels if isinstance(last, (ResBlock, Upsample)) else last.in_channels
            return (co, 2 * H, 2 * W) if isinstance(last, Upsample) else (co, H, W)

        # (the middle block's output goes into the first concat, a decoder block's into the next one: no direct GroupNorm consumer, but
        #  the GEMM that produces it can write that concat's normalised + raw planes -- Ctx.next_cat)
        h, H, W = self.middle_block.run(ctx, h, H, W, next_cat=cat_hint(self.output_blocks[0], *out_geometry(self.middle_block, H, W)))
        for bi, blk in enumerate(self.output_blocks):
            sk, _, _ = hs.pop()
            M = B * H * W
            ca, cb = h.shape[-1], sk.shape[-1]
            cat = ctx.ws.get("cat", (M, ca + cb))
            catp = ctx.ws.planes("catp", M, ca + cb)          # planes for the ResBlock's 1x1 skip conv
            first = blk[0]
            if cat.data_ptr() in ctx._cat_done:               # produced by the previous block's output GEMM
                ctx._cat_done.discard(cat.data_ptr())
            else:
                # concat + the block's leading GroupNorm / SiLU in one launch; the fp32 concatenation itself is only read by a ResBlock
                # without a skip convolution (none in this decoder) and by the training recorder
                need_cat = keep or not isinstance(first, ResBlock) or isinstance(first.skip_connection, nn.Identity)
                ctx.concat(h, ca, sk, cb, cat, catp, B, H * W, gn_apply=gn_hint(first), need_out=need_cat)
            if rec is not None:
                rec.append((blk, cat.clone(), H, W, ca))
            last_blk = bi + 1 == len(self.output_blocks)
            # (a decoder block's output goes into the next concat; the last one feeds the head's GroupNorm + SiLU)
            h, H, W = blk.run(ctx, cat, H, W, x_planes=catp, next_layer=(self.out[0], "res.a", True) if last_blk else None,
                              next_cat=None if last_blk else cat_hint(self.output_blocks[bi + 1], *out_geometry(blk, H, W)))
        if self._head is None:
            self._head = hip.pack_conv3x3(self.out[2].weight, self.out[2].bias)
        M = B * H * W
        a = ctx.ws.planes("res.a", M, self.model_channels)
        ctx.groupnorm(h, a, self.out[0], B, H * W, self.model_channels, silu=True)
        y = ctx.ws.get("head", (M, 8))
        ctx.gemm(a, self._head, y, conv=dict(B=B, Hin=H, Win=W, Cin=self.model_channels, Hout=H, Wout=W, stride=1,
                                             upsample=0), ldo=8, kind="conv")
        self._head_saved = (h, a)          # input of the head and planes of SiLU(GN(h)): what backward.unet_head_backward needs
        return y


class UNetWrapper(nn.Module):
    """mvdfusion/unet.py:56-209.  ``unet_config`` is the yaml node {target, params} (or a params dict)."""

    def __init__(self, unet_config, unet_path=None, drop_conditions=False, drop_scheme="default", use_zero_123=False,
                 finetune_unet=False, finetune_cross_attn=False, finetune_view_attn=True, remove_keys=()):
        super().__init__()
        params = unet_config.get("params", unet_config) if hasattr(unet_config, "get") else unet_config
        self.unet_model = UNetModel(**dict(params))
        if unet_path:
            from .load_model import load_unet_checkpoint
            load_unet_checkpoint(self.unet_model, unet_path, remove_keys=remove_keys)
        if not finetune_unet:
            self.unet_model.disable_unet_grad()
        self.drop_conditions, self.drop_scheme, self.use_zero_123 = drop_conditions, drop_scheme, use_zero_123
        self.finetune_unet, self.finetune_cross_attn, self.finetune_view_attn = \
            finetune_unet, finetune_cross_attn, finetune_view_attn

    def get_trainable_parameters(self):
        if self.finetune_unet:
            return self.unet_model.parameters()
        return self.unet_model.get_cross_attn_parameters(finetune_cross_attn=self.finetune_cross_attn,
                                                         finetune_view_attn=self.finetune_view_attn)

    def level_channels(self):
        """Channel count of the ViewAlignedFeatureTransformers at each pyramid level (resolution S / 2^i)."""
        u = self.unet_model
        return [u.model_channels * m for m in u.channel_mult]

    def level0_operand(self, ctx, B, S, D):
        """The buffer the feature frustum's planes are written to (by GridAttn's last GEMM) and its first column.
        D == 1: the level-0 [attention output | volume features] operand (rows, C0 + 768) of the merged to_out /
        cross-attention GEMM (attention.py), volume columns at C0; D > 1: plain (rows * D, 768) planes."""
        rows = B * S * S * D
        if D == 1:
            c0 = self.level_channels()[0]
            return ctx.ws.get("ovol0", (rows, 2 * (c0 + 768)), torch.int16, zero=True), c0
        return ctx.ws.get("vol0", (rows, 2 * 768), torch.int16, zero=True), 0

    def volume_pyramid(self, ctx, vol, B, S, D):
        """get_volume_feats_pyramid (unet.py:198-209): area pooling at x{1, 1/2, 1/4, 1/8}; vol (B,S,S,D,768) fp32, whose
        planes already sit in level0_operand().  Returns [(planes buffer, first volume column)] per level; the pooled levels
        only feed GEMMs, so they are written as planes straight into their consumers' operand buffers."""
        levels = [self.level0_operand(ctx, B, S, D)]
        Cc = vol.shape[-1]
        chans = self.level_channels()
        for i in range(1, len(self.unet_model.channel_mult)):
            f = 2 ** i
            rows = B * (S // f) * (S // f) * D
            if D == 1:
                ld, col = chans[i] + Cc, chans[i]
                o = ctx.ws.get(f"ovol{i}", (rows, 2 * ld), torch.int16, zero=True)
            else:
                ld, col = Cc, 0
                o = ctx.ws.planes(f"vol{i}", rows, Cc)
            hip.check(hip.lib().mvd_area_pool(hip.ptr(vol), hip.c_void_p(o.data_ptr() + 4 * col), B, S, D, Cc, f, ld, hip.stream()))
            levels.append((o, col))
        return levels
