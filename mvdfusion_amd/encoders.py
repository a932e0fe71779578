"""CLIP image encoder of the conditioning branch, executed by HIP kernels -- mirror of
``external.sd1.ldm.modules.encoders.modules.FrozenCLIPImageEmbedder`` (encoders/modules.py:402-441).

The reference wraps OpenAI's ``clip`` package (``clip.load('ViT-L/14')``, text tower deleted, encoders/modules.py:415-417), which
is NOT part of the reference tree; the vision transformer below follows its published architecture (CLIP ``model.py``:
``VisionTransformer`` / ``ResidualAttentionBlock`` / ``QuickGELU``) and keeps its parameter names, so a checkpoint's
``clip_image_encoder.model.visual.*`` keys load unchanged (the text-side leftovers ``token_embedding``, ``positional_embedding``,
``ln_final``, ``text_projection``, ``logit_scale`` are held too).

Per call (once per sample, viewfusion_zero_depth_rgb.py:266): bicubic resize to 224^2 + CLIP normalisation (torch-ROCm,
plumbing) -> 14x14 patch embedding as a GEMM -> [class | patches] + positions, ln_pre -> 24 x [LN -> fused QKV GEMM (+bias)
-> flash attention (16 heads x 64, 257 keys in 260-row sequences) -> out_proj GEMM (+residual) -> LN -> c_fc GEMM + QuickGELU
-> c_proj GEMM (+residual)] -> ln_post on the class token -> x @ proj.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from . import hip
from .engine import Ctx

_CONFIGS = {   # name -> (image, patch, width, layers, heads, out_dim, text_width)
    "ViT-L/14": (224, 14, 1024, 24, 16, 768, 768),
    "ViT-B/16": (224, 16, 768, 12, 12, 512, 512),
    "tiny-test": (224, 14, 128, 2, 2, 64, 64),        # two 64-wide heads: same kernels, seconds on the CPU oracle
}


class _ResidualAttentionBlock(nn.Module):
    def __init__(self, width, heads):
        super().__init__()
        self.attn = nn.MultiheadAttention(width, heads)
        self.ln_1 = nn.LayerNorm(width)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(width, width * 4)), ("gelu", nn.Identity()),
                                              ("c_proj", nn.Linear(width * 4, width))]))
        self.ln_2 = nn.LayerNorm(width)
        self.width, self.heads = width, heads
        self._p = None

    def packed(self):
        if self._p is None:
            self._p = (hip.pack_linear(self.attn.in_proj_weight, self.attn.in_proj_bias),
                       hip.pack_linear(self.attn.out_proj.weight, self.attn.out_proj.bias),
                       hip.pack_linear(self.mlp.c_fc.weight, self.mlp.c_fc.bias),
                       hip.pack_linear(self.mlp.c_proj.weight, self.mlp.c_proj.bias))
        return self._p


class _Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.resblocks = nn.Sequential(*[_ResidualAttentionBlock(width, heads) for _ in range(layers)])


class _VisionTransformer(nn.Module):
    def __init__(self, image, patch, width, layers, heads, out_dim):
        super().__init__()
        self.image, self.patch, self.width, self.heads, self.out_dim = image, patch, width, heads, out_dim
        scale = width ** -0.5
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch, stride=patch, bias=False)
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((image // patch) ** 2 + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = _Transformer(width, layers, heads)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, out_dim))
        self._p = None

    def packed(self):
        if self._p is None:
            self._p = hip.pack_linear(self.conv1.weight.detach().reshape(self.width, -1))
        return self._p


class _CLIP(nn.Module):
    """Parameter layout of clip.model.CLIP after ``del model.transformer`` (encoders/modules.py:417)."""

    def __init__(self, image, patch, width, layers, heads, out_dim, text_width):
        super().__init__()
        self.visual = _VisionTransformer(image, patch, width, layers, heads, out_dim)
        self.token_embedding = nn.Embedding(49408, text_width)
        self.positional_embedding = nn.Parameter(torch.empty(77, text_width))
        self.ln_final = nn.LayerNorm(text_width)
        self.text_projection = nn.Parameter(torch.empty(text_width, out_dim))
        self.logit_scale = nn.Parameter(torch.ones([]))


class FrozenCLIPImageEmbedder(nn.Module):
    def __init__(self, model="ViT-L/14", jit=False, device="cpu", antialias=False, precision="f16x4"):
        super().__init__()
        assert not jit and not antialias, "the reference constructs it with jit=False, antialias=False"
        # `model` is a CLIP model name or -- as in configs/*.yaml (clip_path: weights/clip_vit_14.ckpt) -- a checkpoint file that
        # clip.load() would open: the shipped configs all use ViT-L/14
        arch = model if model in _CONFIGS else "ViT-L/14"
        self.model = _CLIP(*_CONFIGS[arch])
        if model not in _CONFIGS:
            import os
            if not os.path.exists(model):      # clip.load() raises on an unknown name / missing file: never run on random weights silently
                raise FileNotFoundError(f"FrozenCLIPImageEmbedder: '{model}' is neither a known CLIP architecture {sorted(_CONFIGS)} nor "
                                        "an existing checkpoint file")
            try:
                sd = torch.jit.load(model, map_location="cpu").state_dict()       # OpenAI ships TorchScript archives
            except Exception:
                sd = torch.load(model, map_location="cpu")
                sd = sd.get("state_dict", sd)
            missing, _ = self.model.load_state_dict({k: v.float() for k, v in sd.items()}, strict=False)
            lost = [k for k in missing if k.startswith("visual.")]
            if lost:
                raise KeyError(f"FrozenCLIPImageEmbedder: checkpoint '{model}' lacks {len(lost)} vision-tower tensors (first: {lost[:3]})")
        self.antialias = antialias
        self.register_buffer("mean", torch.Tensor([0.48145466, 0.4578275, 0.40821073]), persistent=False)
        self.register_buffer("std", torch.Tensor([0.26862954, 0.26130258, 0.27577711]), persistent=False)
        self.precision = hip.parse_precision(precision)[1]       # (a per-class policy addresses the UNet's layer classes only)
        self._ctx, self._packed_sig = None, None

    def preprocess(self, x):
        """encoders/modules.py:422-431: kornia bicubic resize (align_corners=True, no antialias) == F.interpolate, then CLIP's
        mean / std (host-side tensor plumbing on the GPU, once per sample)."""
        v = self.model.visual
        x = torch.nn.functional.interpolate(x, size=(v.image, v.image), mode="bicubic", align_corners=True)
        x = (x + 1.0) / 2.0
        return (x - self.mean.view(1, 3, 1, 1)) / self.std.view(1, 3, 1, 1)

    @torch.no_grad()
    def forward(self, x):
        """x in [-1, 1], (B, 3, H, W) -> (B, out_dim) fp32 (encoders/modules.py:433-439)."""
        if isinstance(x, list):     # [""] = condition dropout of the unconditional branch (:435-438)
            return torch.zeros(1, self.model.visual.out_dim, device=self.model.visual.conv1.weight.device)
        if not x.is_cuda:
            raise RuntimeError("FrozenCLIPImageEmbedder runs on the HIP path only (no CPU fallback)")
        sig = hip.params_signature(self)
        if sig != self._packed_sig:
            if self._packed_sig is not None:
                hip.drop_packed_caches(self)
            self._packed_sig = sig
        if self._ctx is None:
            self._ctx = Ctx(x.device, self.precision)
        return self._encode_image(self._ctx, self.preprocess(x.float()))

    def encode(self, im):
        return self(im).unsqueeze(1)

    def _encode_image(self, ctx, x):
        v = self.model.visual
        B, W, H = x.shape[0], v.width, v.heads
        g = v.image // v.patch
        P, L = g * g, g * g + 1
        Lr = (L + 3) // 4 * 4                     # rows per sequence: the QKV epilogue stores 4 tokens at a time
        M = B * Lr
        # patch embedding: Conv2d(3, W, patch, stride patch, no bias) == GEMM over the (c, ky, kx)-ordered patches
        K = 3 * v.patch * v.patch
        patches = x.unfold(2, v.patch, v.patch).unfold(3, v.patch, v.patch).permute(0, 2, 3, 1, 4, 5).reshape(B * P, K)
        pp = hip.split_planes(patches.contiguous(), ctx.ws.planes("clip.patches", B * P, (K + 31) // 32 * 32))
        emb = ctx.ws.get("clip.emb", (B * P, W))
        ctx.gemm(pp, v.packed(), emb)
        tok = ctx.ws.get("clip.tok", (B, Lr, W), zero=True)
        tok[:, 0] = v.class_embedding + v.positional_embedding[0]
        tok[:, 1:L] = emb.view(B, P, W) + v.positional_embedding[1:]
        xa, xb = ctx.ws.get("clip.xa", (M, W)), ctx.ws.get("clip.xb", (M, W))
        hip.layernorm(tok.view(M, W), None, v.ln_pre.weight, v.ln_pre.bias, M, W, v.ln_pre.eps, y_f32=xa)
        ln = ctx.ws.planes("clip.ln", M, W)
        o = ctx.ws.planes("clip.o", M, W)
        hid = ctx.ws.planes("clip.hid", M, 4 * W)
        planes = ctx.ws.attn_planes(B, H, Lr, W // H)
        for blk in v.transformer.resblocks:
            w_qkv, w_out, w_fc, w_proj = blk.packed()
            hip.layernorm(xa, ln, blk.ln_1.weight, blk.ln_1.bias, M, W, blk.ln_1.eps)
            ctx.gemm(ln, w_qkv, None, epi=hip.EPI_QKV, qkv=dict(planes=planes, heads=H, dhead=W // H, L=Lr))
            hip.attention(planes, o, B, H, Lr, W // H, prec=ctx.prec, Lkeys=L)       # padding rows never act as keys
            ctx.gemm(o, w_out, xb, res=xa)
            hip.layernorm(xb, ln, blk.ln_2.weight, blk.ln_2.bias, M, W, blk.ln_2.eps)
            ctx.gemm(ln, w_fc, None, act=hip.ACT_QUICKGELU, out_planes=hid)
            ctx.gemm(hid, w_proj, xa, res=xb)
        cls = xa.view(B, Lr, W)[:, 0].contiguous()
        post = ctx.ws.get("clip.post", (B, W))
        hip.layernorm(cls, None, v.ln_post.weight, v.ln_post.bias, B, W, v.ln_post.eps, y_f32=post)
        out = torch.empty(B, v.out_dim, device=x.device)
        wt = ctx.ws.get("clip.projT", (v.out_dim, W))
        wt.copy_(v.proj.detach().t())
        ctx.gemv_rows(wt, None, post, out)
        return out
