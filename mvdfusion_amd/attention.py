"""Transformer blocks of the view-conditioned UNet, executed by HIP kernels.

Mirrors (same attribute names => same state_dict keys):
  * ``CrossAttention``, ``FeedForward``/``GEGLU``, ``BasicTransformerBlock``, ``SpatialTransformer``
    (external/sd1/ldm/modules/attention.py:37-64, 152-287)
  * ``DualAttnetionBlock`` [sic], ``ViewAlignedFeatureTransformer`` (mvdfusion/attention.py:16-145)

The nn.Linear / nn.Conv2d / nn.LayerNorm / nn.GroupNorm members only HOLD the fp32 parameters under the reference's
names; ``run(ctx, x, H, W)`` is the forward pass on channels-last activations (a (B*H*W, C) fp32 matrix):

  GroupNorm -> proj_in GEMM -> LN -> fused QKV GEMM (epilogue writes the split-bf16 attention operands) ->
  flash attention (MFMA) -> to_out GEMM (+bias +residual [+ per-view cross-attention vector]) -> LN ->
  GEGLU GEMM (gate fused in the epilogue) -> ONE GEMM for ff-out (+residual) followed by proj_out (+residual).

Cross-attention against a length-1 context (the CLIP vector; the D==1 depth sample) is softmax over one key == 1, so
it reduces exactly to to_out(to_v(ctx)); to_q / to_k / norm2 are dead there and are skipped.

Launch merging by operand concatenation along K (exact algebra; the composed weights are formed once in fp64 and rounded to
fp32 -- weight preprocessing like packing):
  * ff-out and proj_out have no nonlinearity between them:  out = (g W2^T + b2 + t) Wp^T + bp + x
      = [g | t] [Wp W2 | Wp]^T + (Wp b2 + bp) + x      -- the GEGLU epilogue and the producer of t write their planes side by
    side into one (M, 5C) operand buffer;
  * ViewAlignedFeatureTransformer with D == 1:  t2b = o Wo^T + bo + t + to_out2(to_v(vol))
      = [o | vol] [Wo | Wo2 Wv]^T + (bo + bo2) + t     -- the attention kernel writes o next to the level's volume features.
"""
import torch
import torch.nn as nn

from . import hip


def _normalize(c):
    return nn.GroupNorm(num_groups=32, num_channels=c, eps=1e-6, affine=True)


class CrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner = dim_head * heads
        context_dim = query_dim if context_dim is None else context_dim
        self.heads, self.dim_head, self.scale = heads, dim_head, dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))
        self._p = {}

    def packed(self, key):
        """Lazily packed MFMA operand images: 'qkv' (fused), 'q', 'k', 'v', 'out'."""
        if key not in self._p:
            if key == "qkv":
                self._p[key] = hip.pack_linear_cat([self.to_q.weight, self.to_k.weight, self.to_v.weight])
            elif key == "out":
                self._p[key] = hip.pack_linear(self.to_out[0].weight, self.to_out[0].bias)
            else:
                self._p[key] = hip.pack_linear(getattr(self, "to_" + key).weight)
        return self._p[key]

    def packed_qkv_ln(self, norm):
        """The fused QKV weight with the LayerNorm in front of it folded in (hip.LnFold)."""
        if getattr(self, "_lnf", None) is None:
            self._lnf = hip.LnFold(torch.cat([self.to_q.weight.detach(), self.to_k.weight.detach(), self.to_v.weight.detach()], 0), None, norm)
        return self._lnf


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4, dropout=0.0):
        super().__init__()
        inner = int(dim * mult)
        self.net = nn.Sequential(GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim))
        self._p = None

    def packed(self):
        """(GEGLU weight, ff-out weight): the second is only used by callers that do not merge it with their proj_out."""
        if self._p is None:
            self._p = (hip.pack_linear(self.net[0].proj.weight, self.net[0].proj.bias, geglu=True),
                       hip.pack_linear(self.net[2].weight, self.net[2].bias))
        return self._p

    def packed_geglu_ln(self, norm):
        """The GEGLU projection with the LayerNorm in front of it folded in (hip.LnFold)."""
        if getattr(self, "_lnf", None) is None:
            self._lnf = hip.LnFold(self.net[0].proj.weight, self.net[0].proj.bias, norm, geglu=True)
        return self._lnf

    def packed_geglu(self):
        if self._p is not None:
            return self._p[0]
        if getattr(self, "_pg", None) is None:
            self._pg = hip.pack_linear(self.net[0].proj.weight, self.net[0].proj.bias, geglu=True)
        return self._pg


def compose_ff_out_proj(ff, proj_w, proj_b):
    """[Wp W2 | Wp] (C, 5C) and Wp b2 + bp: ff-out (attention.py:60) followed by proj_out (attention.py:259 /
    mvdfusion/attention.py:114) as one Linear over the concatenated operand [g | t]."""
    Wp = proj_w.detach().reshape(proj_w.shape[0], -1).double()
    W2, b2 = ff.net[2].weight.detach().double(), ff.net[2].bias.detach().double()
    Wm = torch.cat([Wp @ W2, Wp], dim=1).float().contiguous()
    bm = (Wp @ b2 + proj_b.detach().double()).float().contiguous()
    return hip.pack_linear(Wm, bm)


class _TransformerCore(nn.Module):
    """attn1 / attn2 / ff / norm1-3 holder shared by BasicTransformerBlock and DualAttnetionBlock."""

    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None):
        super().__init__()
        self.attn1 = CrossAttention(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout, context_dim=None)
        self.ff = FeedForward(dim, dropout=dropout)
        self.attn2 = CrossAttention(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head,
                                    dropout=dropout)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)
        self.dim, self.n_heads, self.d_head = dim, n_heads, d_head

    def self_attn(self, ctx, t, B, L, tag, o=None, t_planes=None, t_stats=None):
        """returns t + attn1(norm1(t)) pieces: the attention output `o` (pre to_out); `o` may be a wider planes buffer whose
        first C columns receive it.  t_planes / t_stats: the producer of t wrote its split planes and row statistics -- norm1 is then
        folded into the QKV GEMM (no LayerNorm kernel)."""
        C, M = self.dim, B * L
        planes = ctx.ws.attn_planes(B, self.n_heads, L, self.d_head)
        qkv = dict(planes=planes, heads=self.n_heads, dhead=self.d_head, L=L)
        if t_planes is not None:
            fold = self.attn1.packed_qkv_ln(self.norm1)
            ctx.gemm(t_planes, fold.w, None, epi=hip.EPI_QKV, qkv=qkv, ln=(t_stats, fold), kind="qkv")
        else:
            ln = ctx.ws.planes(tag + ".ln", M, C)
            ctx.layernorm(t, ln, self.norm1, M, C)
            ctx.gemm(ln, self.attn1.packed("qkv"), None, epi=hip.EPI_QKV, qkv=qkv, kind="qkv")
        if o is None:
            o = ctx.ws.planes(tag + ".o", M, C)
        hip.attention(planes, o, B, self.n_heads, L, self.d_head, prec=ctx.prec_of("attn"))
        return o

    def cat5(self, ctx, M, tag):
        """The (M, 5C) operand [g | t] of the merged ff-out / proj_out GEMM: the producer of t fills columns [4C, 5C)."""
        return ctx.ws.planes(tag + ".cat5", M, 5 * self.dim)

    def feed_forward_proj(self, ctx, t2, cat5, w_merged, x, out, M, tag, gn=None, t2_stats=None):
        """out = proj_out(t2 + ff(norm3(t2))) + x with t2's planes already in cat5[:, 4C:] (see module docstring).  t2_stats: the row
        statistics its producer emitted -- norm3 is then folded into the GEGLU GEMM, which reads t2's planes where they already are."""
        C = self.dim
        if t2_stats is not None:
            fold = self.ff.packed_geglu_ln(self.norm3)
            ctx.gemm(cat5[:, 2 * 4 * C:], fold.w, None, M=M, lda=5 * C, epi=hip.EPI_GEGLU, out_planes=cat5, ln=(t2_stats, fold), kind="geglu")
        else:
            ln = ctx.ws.planes(tag + ".ln", M, C)
            ctx.layernorm(t2, ln, self.norm3, M, C)
            ctx.gemm(ln, self.ff.packed_geglu(), None, epi=hip.EPI_GEGLU, out_planes=cat5, kind="geglu")      # columns [0, 4C)
        ctx.gemm(cat5, w_merged, out, res=x, gn=gn, kind="ffproj")
        return out


class BasicTransformerBlock(_TransformerCore):
    pass


class DualAttnetionBlock(_TransformerCore):
    pass


class SpatialTransformer(nn.Module):
    """external/sd1/ldm/modules/attention.py:225-287 (use_linear=False: proj_in/out are 1x1 convs)."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None):
        super().__init__()
        assert depth == 1
        inner = n_heads * d_head
        self.in_channels = in_channels
        self.norm = _normalize(in_channels)
        self.proj_in = nn.Conv2d(in_channels, inner, kernel_size=1, stride=1, padding=0)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, n_heads, d_head, dropout, context_dim)])
        self.proj_out = nn.Conv2d(inner, in_channels, kernel_size=1, stride=1, padding=0)
        self._p = None

    def packed(self):
        if self._p is None:
            self._p = (hip.pack_linear(self.proj_in.weight, self.proj_in.bias),
                       compose_ff_out_proj(self.transformer_blocks[0].ff, self.proj_out.weight, self.proj_out.bias))
        return self._p

    def run(self, ctx, x, H, W, out=None):
        B, C, L = ctx.B, self.in_channels, H * W
        M = B * L
        tb = self.transformer_blocks[0]
        w_in, w_ffproj = self.packed()
        n = ctx.ws.planes("tf.n", M, C)
        ctx.groupnorm(x, n, self.norm, B, L, C, silu=False)
        t = ctx.ws.get("tf.t", (M, C))
        fold = ctx.ln_fold_enabled()
        tp, rs1, rs2 = (ctx.ws.planes("tf.tp", M, C), ctx.row_stats("tf.rs1", M, C), ctx.row_stats("tf.rs2", M, C)) if fold else (None,) * 3
        ctx.gemm(n, w_in, t, out_planes=tp, row_stats=rs1, kind="proj")
        o = tb.self_attn(ctx, t, B, L, "tf", t_planes=tp, t_stats=rs1)
        # attn2 on the length-1 CLIP context: per-view vector to_out(to_v(ctx_b)), broadcast over the pixels
        a2 = tb.attn2
        xvec = getattr(ctx, "xattn_vec", None)
        if xvec is not None and self in xvec:
            vec = xvec[self]       # slice of the one batched GEMV over all 16 layers (UNetModel.cross_attn_vectors)
        else:
            v1 = ctx.ws.get("tf.v1", (B, C))
            ctx.gemv_rows(a2.to_v.weight, None, ctx.context, v1)
            vec = ctx.ws.get("tf.vec", (B, C))
            ctx.gemv_rows(a2.to_out[0].weight, a2.to_out[0].bias, v1, vec)
        # (with the LayerNorm fold the fp32 copy of t2 has no reader: the GEGLU GEMM takes t2's planes + row statistics, the merged
        #  ff-out / proj_out GEMM takes the planes from cat5 -- one (M, C) fp32 write less per transformer)
        t2 = None if fold else ctx.ws.get("tf.t2", (M, C))
        cat5 = tb.cat5(ctx, M, "tf")
        ctx.gemm(o, tb.attn1.packed("out"), t2, res=t, bias_b=vec, rows_per_batch=L, out_planes=cat5, out_planes_col=4 * C, row_stats=rs2, kind="out")
        if out is None:
            out = ctx.act((M, C))
        return tb.feed_forward_proj(ctx, t2, cat5, w_ffproj, x, out, M, "tf", gn=(B, L), t2_stats=rs2)


class ViewAlignedFeatureTransformer(nn.Module):
    """mvdfusion/attention.py:72-145 (use_linear=True).  Parameters are prefixed ``aligned_attn_``."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None, image_size=None):
        super().__init__()
        assert depth == 1
        inner = n_heads * d_head
        self.in_channels, self.image_size = in_channels, image_size
        self.aligned_attn_norm = _normalize(in_channels)
        self.aligned_attn_proj_in = nn.Linear(in_channels, inner)
        self.aligned_attn_transformer_blocks = nn.ModuleList(
            [DualAttnetionBlock(inner, n_heads, d_head, dropout, context_dim)])
        self.aligned_attn_proj_out = nn.Linear(in_channels, inner)
        self.level_mapper = {image_size: 0, image_size // 2: 1, image_size // 4: 2, image_size // 8: 3}
        self._p = None

    def packed(self):
        if self._p is None:
            self._p = (hip.pack_linear(self.aligned_attn_proj_in.weight, self.aligned_attn_proj_in.bias),
                       compose_ff_out_proj(self.aligned_attn_transformer_blocks[0].ff, self.aligned_attn_proj_out.weight,
                                           self.aligned_attn_proj_out.bias))
        return self._p

    def packed_ovol(self):
        """D == 1: [Wo | Wo2 Wv] (C, C + 768) with bias bo + bo2 -- attn1.to_out and the length-1 cross attention
        to_out2(to_v(vol)) (mvdfusion/attention.py:52-62) as one Linear over the concatenated operand [o | vol]."""
        if getattr(self, "_fused", None) is None:
            tb = self.aligned_attn_transformer_blocks[0]
            Wo, bo = tb.attn1.to_out[0].weight.detach().double(), tb.attn1.to_out[0].bias.detach().double()
            Wo2, bo2 = tb.attn2.to_out[0].weight.detach().double(), tb.attn2.to_out[0].bias.detach().double()
            Wv = tb.attn2.to_v.weight.detach().double()
            self._fused = hip.pack_linear(torch.cat([Wo, Wo2 @ Wv], dim=1).float().contiguous(), (bo + bo2).float().contiguous())
        return self._fused

    def run(self, ctx, x, H, W, out=None):
        B, C, L, D = ctx.B, self.in_channels, H * W, ctx.D
        M = B * L
        tb = self.aligned_attn_transformer_blocks[0]
        w_in, w_ffproj = self.packed()
        # D == 1: the level's (M, C + 768) operand buffer [attention output | volume features]; D > 1: plain (M*D, 768) planes
        vol, vol_col = ctx.vol_levels[self.level_mapper[H]]
        n = ctx.ws.planes("tf.n", M, C)
        ctx.groupnorm(x, n, self.aligned_attn_norm, B, L, C, silu=False)
        t = ctx.ws.get("tf.t", (M, C))
        fold = ctx.ln_fold_enabled()
        tp, rs1, rs2 = (ctx.ws.planes("tf.tp", M, C), ctx.row_stats("tf.rs1", M, C), ctx.row_stats("tf.rs2", M, C)) if fold else (None,) * 3
        ctx.gemm(n, w_in, t, out_planes=tp, row_stats=rs1, kind="proj")
        t2b = None if fold else ctx.ws.get("tf.t2b", (M, C))
        cat5 = tb.cat5(ctx, M, "tf")
        if D == 1:
            assert vol_col == C and vol.shape[-1] == 2 * (C + 768), (vol_col, C, vol.shape)
            tb.self_attn(ctx, t, B, L, "tf", o=vol, t_planes=tp, t_stats=rs1)            # o -> columns [0, C) of the [o | vol] operand
            ctx.gemm(vol, self.packed_ovol(), t2b, res=t, out_planes=cat5, out_planes_col=4 * C, row_stats=rs2, kind="out")
        else:
            o = tb.self_attn(ctx, t, B, L, "tf", t_planes=tp, t_stats=rs1)
            t2 = ctx.ws.get("tf.t2", (M, C))
            ctx.gemm(o, tb.attn1.packed("out"), t2, res=t, kind="out")
            a2 = tb.attn2.packed
            ln2 = ctx.ws.planes("tf.ln", M, C)
            ctx.layernorm(t2, ln2, tb.norm2, M, C)
            q = ctx.ws.get("tf.q2", (M, C))
            ctx.gemm(ln2, a2("q"), q, kind="xattn")
            k = ctx.ws.get("tf.k2", (M * D, C))
            v = ctx.ws.get("tf.v2", (M * D, C))
            ctx.gemm(vol, a2("k"), k, kind="xattn")
            ctx.gemm(vol, a2("v"), v, kind="xattn")
            o2 = ctx.ws.planes("tf.o2", M, C)
            hip.check(hip.lib().mvd_pixel_cross_attn(hip.ptr(q), hip.ptr(k), hip.ptr(v), hip.ptr(o2), M, D, tb.n_heads, tb.d_head, hip.stream()))
            ctx.gemm(o2, a2("out"), t2b, res=t2, out_planes=cat5, out_planes_col=4 * C, row_stats=rs2, kind="xattn")
        if out is None:
            out = ctx.act((M, C))
        return tb.feed_forward_proj(ctx, t2b, cat5, w_ffproj, x, out, M, "tf", gn=(B, L), t2_stats=rs2)
