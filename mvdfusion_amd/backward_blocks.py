"""Block-level backward of the view-conditioned UNet on the HIP path: ResBlock, SpatialTransformer, ViewAlignedFeatureTransformer
(reference: openaimodel.py:255-275, sd1 attention.py:225-287, mvdfusion/attention.py:16-145; train.py:90-95 `loss.backward()`).

Activation checkpointing, as the reference does (``checkpoint(self._forward, ...)``, use_checkpoint=True): a block's backward takes
the block INPUT saved by the forward, re-runs the block's forward UNFUSED on the HIP ops keeping every intermediate, then walks the
chain rule with the backward kernels of mvdfusion_amd/backward.py (dgrad / wgrad on the split-operand MFMA GEMM; GroupNorm, LayerNorm,
GEGLU, self-attention and per-pixel cross-attention backward in csrc/backward.hip).  Tiny per-view algebra (the length-1 CLIP
cross-attention vector: (B, C) x (C, 768) products, the time-embedding outer product) is host glue in torch.

Returned gradients are keyed by the parameter names RELATIVE to the block (``in_layers.2.weight``, ``transformer_blocks.0.attn1.to_q.weight`` ...),
i.e. state_dict key = block prefix + name.  Parameters the loss does not depend on get exact zeros, as autograd gives them
(attn2.to_q / to_k / norm2 of a BasicTransformerBlock whose context has length 1: softmax over one key is constant).
"""
import torch

from . import backward as bw
from . import hip


class Tape:
    """Workspace + thin forward helpers shared by the block backwards (fresh tensors per call: the training path is not the
    allocation-free inference path)."""

    def __init__(self, device, prec=hip.PREC_X4, workspace=None, only_trainable=False):
        self.device = torch.device(device)
        self.prec = prec
        self.only_trainable = only_trainable     # skip the weight gradients of frozen parameters (requires_grad False)
        self.ws = workspace if workspace is not None else torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=self.device)

    # ---- forward pieces
    def planes(self, x):
        return hip.split_planes(x.contiguous())

    def linear(self, a_planes, weight, bias=None, res=None, M=None, colscale=None):
        """res + colscale * (A W^T + bias) (colscale: per-column gate, e.g. adaLN's g1 / g2; needs N % 16 == 0)."""
        w = hip.pack_linear(weight, bias)
        M = a_planes.shape[0] if M is None else M
        out = torch.empty(M, w.N, dtype=torch.float32, device=self.device)
        assert colscale is None or w.N == weight.shape[0]
        hip.gemm(a_planes, w, out, prec=self.prec, res=res, workspace=self.ws, colscale=colscale)
        return out[:, :weight.shape[0]] if w.N != weight.shape[0] else out

    def conv3x3(self, a_planes, weight, bias, B, H, W, res=None):
        w = hip.pack_conv3x3(weight, bias)
        out = torch.empty(B * H * W, w.N, dtype=torch.float32, device=self.device)
        hip.gemm(a_planes, w, out, prec=self.prec, res=res, workspace=self.ws,
                 conv=dict(B=B, Hin=H, Win=W, Cin=w.conv_cin, Hout=H, Wout=W, stride=1, upsample=0))
        return out[:, :weight.shape[0]] if w.N != weight.shape[0] else out

    def groupnorm(self, x, norm, B, HW, silu):
        C = x.shape[-1]
        y = hip.planes_like(B * HW, C, self.device)
        ws = torch.empty(B * hip.lib().mvd_groupnorm_chunks(HW) * 32 * 2, dtype=torch.float64, device=self.device)
        hip.groupnorm(x, y, norm.weight, norm.bias, B, HW, C, norm.eps, silu, ws)
        return y

    def layernorm(self, x, norm):
        rows, C = x.shape
        y = hip.planes_like(rows, C, self.device)
        hip.layernorm(x, y, norm.weight, norm.bias, rows, C, norm.eps)
        return y

    def self_attention(self, ln_planes, attn, B, L):
        """q, k, v (fp32, for the backward) and the attention output planes (forward path: fused QKV epilogue + flash attention)."""
        M, C = B * L, attn.heads * attn.dim_head
        q, k, v = (self.linear(ln_planes, getattr(attn, "to_" + n).weight) for n in "qkv")
        planes = hip.alloc_attn_planes(B, attn.heads, L, attn.dim_head, self.device)
        hip.gemm(ln_planes, hip.pack_linear_cat([attn.to_q.weight, attn.to_k.weight, attn.to_v.weight]), None, prec=self.prec,
                 epi=hip.EPI_QKV, qkv=dict(planes=planes, heads=attn.heads, dhead=attn.dim_head, L=L), workspace=self.ws)
        o = hip.planes_like(M, C, self.device)
        hip.attention(planes, o, B, attn.heads, L, attn.dim_head, prec=self.prec)
        return q, k, v, o

    # ---- backward pieces
    def wants(self, weight):
        """Whether the weight gradient of this parameter (or view of a parameter) has to be computed."""
        if not self.only_trainable:
            return True
        base = weight._base if weight._base is not None else weight
        return bool(weight.requires_grad or base.requires_grad)

    def linear_bwd(self, a_planes, weight, dy, need_dx=True, need_db=True):
        w = self.wants(weight)
        return bw.linear_backward(a_planes, weight, dy.contiguous(), self.ws, need_dx=need_dx, need_db=need_db and w, prec=self.prec,
                                  need_dw=w)

    def conv_bwd(self, a_planes, weight, dy, B, H, W, need_dx=True):
        w = self.wants(weight)
        return bw.conv3x3_backward(a_planes, weight, dy.contiguous(), B, H, W, self.ws, need_dx=need_dx, need_db=w, prec=self.prec,
                                   need_dw=w)


def _c(t):
    return t.contiguous()


def resblock_backward(tape, rb, x, emb, dout, B, H, W):
    """ResBlock (openaimodel.py:255-275; no scale-shift norm, dropout 0).  x (M, Cin) fp32 block input, emb (1, ted) the shared
    time embedding, dout (M, Cout).  Returns (dx, {name: grad}, demb (1, ted))."""
    gn1, conv1 = rb.in_layers[0], rb.in_layers[2]
    gn2, conv2 = rb.out_layers[0], rb.out_layers[3]
    lin_e = rb.emb_layers[1]
    Ci, Co = rb.channels, rb.out_channels
    HW = H * W
    # ---- forward (unfused)
    a1 = tape.groupnorm(x, gn1, B, HW, True)
    se = torch.nn.functional.silu(emb)                                     # (1, ted)   host glue
    e = se @ lin_e.weight.t() + lin_e.bias                                 # (1, Co)
    h1 = tape.conv3x3(a1, conv1.weight, conv1.bias + e[0], B, H, W)
    a2 = tape.groupnorm(h1, gn2, B, HW, True)
    has_skip = not isinstance(rb.skip_connection, torch.nn.Identity)
    xp = tape.planes(x) if has_skip else None
    # ---- backward
    g = {}
    da2, g["out_layers.3.weight"], g["out_layers.3.bias"] = tape.conv_bwd(a2, conv2.weight, dout, B, H, W)
    dh1, g["out_layers.0.weight"], g["out_layers.0.bias"] = bw.groupnorm_backward(_c(h1), _c(da2), gn2.weight, gn2.bias, B, HW, Co, gn2.eps, True)
    da1, g["in_layers.2.weight"], db1 = tape.conv_bwd(a1, conv1.weight, dh1, B, H, W)
    if db1 is None:                                                         # conv1 frozen: the column sum is still the gradient of the
        db1 = bw.col_sum(_c(dh1), B * HW, Co)                               # time-embedding vector (always needed)
    g["in_layers.2.bias"] = db1
    # the time-embedding vector is added per channel to every row: its gradient is the same column sum
    g["emb_layers.1.bias"] = db1.clone()
    g["emb_layers.1.weight"] = torch.outer(db1, se[0])
    demb = (db1[None, :] @ lin_e.weight) * (torch.sigmoid(emb) * (1 + emb * (1 - torch.sigmoid(emb))))
    dx, g["in_layers.0.weight"], g["in_layers.0.bias"] = bw.groupnorm_backward(_c(x), _c(da1), gn1.weight, gn1.bias, B, HW, Ci, gn1.eps, True)
    if has_skip:
        sk = rb.skip_connection
        dxs, dWs, g["skip_connection.bias"] = tape.linear_bwd(xp, sk.weight.reshape(Co, Ci), dout)
        g["skip_connection.weight"] = None if dWs is None else dWs.reshape(sk.weight.shape)
        dx = dx + dxs
    else:
        dx = dx + dout
    return dx, g, demb


def _feed_forward(tape, tb, t2):
    """ff(norm3(t2)) pieces: ln3 planes, pre-activation h (M, 8C), gated planes g (M, 4C)."""
    ff1, C = tb.ff.net[0].proj, tb.dim
    ln3 = tape.layernorm(t2, tb.norm3)
    hff = tape.linear(ln3, ff1.weight, ff1.bias)
    gp = hip.planes_like(t2.shape[0], 4 * C, tape.device)
    hip.gemm(ln3, hip.pack_linear(ff1.weight, ff1.bias, geglu=True), None, prec=tape.prec, epi=hip.EPI_GEGLU, out_planes=gp,
             workspace=tape.ws)
    return ln3, hff, gp


def _feed_forward_backward(tape, tb, t2, ln3, hff, gp, dt3, g, pre):
    """Backward of t3 = ff(norm3(t2)) + t2 given dt3; fills g[pre + ...]; returns dt2."""
    ff1, ff2 = tb.ff.net[0].proj, tb.ff.net[2]
    dg, g[pre + "ff.net.2.weight"], g[pre + "ff.net.2.bias"] = tape.linear_bwd(gp, ff2.weight, dt3)
    dh = bw.geglu_backward(_c(hff), _c(dg))
    dln3, g[pre + "ff.net.0.proj.weight"], g[pre + "ff.net.0.proj.bias"] = tape.linear_bwd(ln3, ff1.weight, dh)
    dt2, g[pre + "norm3.weight"], g[pre + "norm3.bias"] = bw.layernorm_backward(_c(t2), _c(dln3), tb.norm3.weight, tb.norm3.eps)
    return dt2 + dt3


def _self_attention_backward(tape, tb, t, ln1, q, k, v, o, dt1, B, L, g, pre):
    """Backward of t1 = attn1(norm1(t)) + t given dt1; returns dt."""
    a1 = tb.attn1
    do, g[pre + "attn1.to_out.0.weight"], g[pre + "attn1.to_out.0.bias"] = tape.linear_bwd(o, a1.to_out[0].weight, dt1)
    dq, dk, dv = bw.attention_backward(_c(q), _c(k), _c(v), _c(do), B, a1.heads, L, a1.dim_head)
    dln1 = None
    for n, d in (("q", dq), ("k", dk), ("v", dv)):
        dx, g[pre + f"attn1.to_{n}.weight"], _ = tape.linear_bwd(ln1, getattr(a1, "to_" + n).weight, d, need_db=False)
        dln1 = dx if dln1 is None else dln1 + dx
    dt, g[pre + "norm1.weight"], g[pre + "norm1.bias"] = bw.layernorm_backward(_c(t), _c(dln1), tb.norm1.weight, tb.norm1.eps)
    return dt + dt1


def spatial_transformer_backward(tape, st, x, context, dout, B, H, W):
    """SpatialTransformer (sd1 attention.py:225-287, 1x1-conv projections, depth 1) with a length-1 context (B, 768).
    Returns (dx, {name: grad}, dcontext (B, 768))."""
    tb = st.transformer_blocks[0]
    C, L = st.in_channels, H * W
    M = B * L
    pre = "transformer_blocks.0."
    w_in, w_out = st.proj_in.weight.reshape(C, C), st.proj_out.weight.reshape(C, C)
    # ---- forward (unfused)
    n = tape.groupnorm(x, st.norm, B, L, False)
    t = tape.linear(n, w_in, st.proj_in.bias)
    ln1 = tape.layernorm(t, tb.norm1)
    q, k, v, o = tape.self_attention(ln1, tb.attn1, B, L)
    t1 = tape.linear(o, tb.attn1.to_out[0].weight, tb.attn1.to_out[0].bias, res=t)
    a2 = tb.attn2
    vctx = context @ a2.to_v.weight.t()                                        # (B, C): the single value row per view (host glue)
    vec = vctx @ a2.to_out[0].weight.t() + a2.to_out[0].bias                   # softmax over one key == 1
    t2 = (t1.view(B, L, C) + vec[:, None, :]).reshape(M, C)
    ln3, hff, gp = _feed_forward(tape, tb, t2)
    ff2 = tb.ff.net[2]
    t3 = tape.linear(gp, ff2.weight, ff2.bias, res=t2)
    t3p = tape.planes(t3)
    # ---- backward
    g = {}
    dt3, dWo, g["proj_out.bias"] = tape.linear_bwd(t3p, w_out, dout)
    g["proj_out.weight"] = None if dWo is None else dWo.reshape(st.proj_out.weight.shape)
    dt2 = _feed_forward_backward(tape, tb, t2, ln3, hff, gp, dt3, g, pre)
    # attn2 with one key: only to_v / to_out see a gradient; to_q, to_k and norm2 get exact zeros
    dvec = dt2.view(B, L, C).sum(1)                                            # (B, C)
    g[pre + "attn2.to_out.0.bias"] = dvec.sum(0)
    g[pre + "attn2.to_out.0.weight"] = dvec.t() @ vctx
    dvctx = dvec @ a2.to_out[0].weight
    g[pre + "attn2.to_v.weight"] = dvctx.t() @ context
    dcontext = dvctx @ a2.to_v.weight
    for name in ("attn2.to_q.weight", "attn2.to_k.weight", "norm2.weight", "norm2.bias"):
        p = dict(tb.named_parameters())[name]
        g[pre + name] = torch.zeros_like(p)
    dt = _self_attention_backward(tape, tb, t, ln1, q, k, v, o, dt2, B, L, g, pre)
    dn, dWi, g["proj_in.bias"] = tape.linear_bwd(n, w_in, dt)
    g["proj_in.weight"] = None if dWi is None else dWi.reshape(st.proj_in.weight.shape)
    dx, g["norm.weight"], g["norm.bias"] = bw.groupnorm_backward(_c(x), _c(dn), st.norm.weight, st.norm.bias, B, L, C, st.norm.eps, False)
    return dx + dout, g, dcontext


def view_aligned_transformer_backward(tape, vt, x, vol, dout, B, H, W, D):
    """ViewAlignedFeatureTransformer (mvdfusion/attention.py:72-145, Linear projections, depth 1).  vol: (B*H*W*D, 768) fp32 volume
    features of this level.  Returns (dx, {name: grad}, dvol (B*H*W*D, 768))."""
    tb = vt.aligned_attn_transformer_blocks[0]
    C, L = vt.in_channels, H * W
    M = B * L
    pre = "aligned_attn_transformer_blocks.0."
    pin, pout, norm = vt.aligned_attn_proj_in, vt.aligned_attn_proj_out, vt.aligned_attn_norm
    a2 = tb.attn2
    # ---- forward (unfused)
    n = tape.groupnorm(x, norm, B, L, False)
    t = tape.linear(n, pin.weight, pin.bias)
    ln1 = tape.layernorm(t, tb.norm1)
    q, k, v, o = tape.self_attention(ln1, tb.attn1, B, L)
    t1 = tape.linear(o, tb.attn1.to_out[0].weight, tb.attn1.to_out[0].bias, res=t)
    ln2 = tape.layernorm(t1, tb.norm2)
    volp = tape.planes(vol)
    q2 = tape.linear(ln2, a2.to_q.weight)
    k2 = tape.linear(volp, a2.to_k.weight)
    v2 = tape.linear(volp, a2.to_v.weight)
    o2 = hip.planes_like(M, C, tape.device)
    hip.check(hip.lib().mvd_pixel_cross_attn(hip.ptr(q2), hip.ptr(k2), hip.ptr(v2), hip.ptr(o2), M, D, a2.heads, a2.dim_head, hip.stream()))
    t2 = tape.linear(o2, a2.to_out[0].weight, a2.to_out[0].bias, res=t1)
    ln3, hff, gp = _feed_forward(tape, tb, t2)
    ff2 = tb.ff.net[2]
    t3 = tape.linear(gp, ff2.weight, ff2.bias, res=t2)
    t3p = tape.planes(t3)
    # ---- backward
    g = {}
    dt3, g["aligned_attn_proj_out.weight"], g["aligned_attn_proj_out.bias"] = tape.linear_bwd(t3p, pout.weight, dout)
    dt2 = _feed_forward_backward(tape, tb, t2, ln3, hff, gp, dt3, g, pre)
    do2, g[pre + "attn2.to_out.0.weight"], g[pre + "attn2.to_out.0.bias"] = tape.linear_bwd(o2, a2.to_out[0].weight, dt2)
    dq2, dk2, dv2 = bw.pixel_cross_attn_backward(_c(q2), _c(k2), _c(v2), _c(do2), M, D, a2.heads, a2.dim_head)
    dln2, g[pre + "attn2.to_q.weight"], _ = tape.linear_bwd(ln2, a2.to_q.weight, dq2, need_db=False)
    dvk, g[pre + "attn2.to_k.weight"], _ = tape.linear_bwd(volp, a2.to_k.weight, dk2, need_db=False)
    dvv, g[pre + "attn2.to_v.weight"], _ = tape.linear_bwd(volp, a2.to_v.weight, dv2, need_db=False)
    dvol = dvk + dvv
    dt1, g[pre + "norm2.weight"], g[pre + "norm2.bias"] = bw.layernorm_backward(_c(t1), _c(dln2), tb.norm2.weight, tb.norm2.eps)
    dt1 = dt1 + dt2
    dt = _self_attention_backward(tape, tb, t, ln1, q, k, v, o, dt1, B, L, g, pre)
    dn, g["aligned_attn_proj_in.weight"], g["aligned_attn_proj_in.bias"] = tape.linear_bwd(n, pin.weight, dt)
    dx, g["aligned_attn_norm.weight"], g["aligned_attn_norm.bias"] = bw.groupnorm_backward(_c(x), _c(dn), norm.weight, norm.bias, B, L, C,
                                                                                          norm.eps, False)
    return dx + dout, g, dvol


def output_block_backward(tape, ctx, blk, cat, cat_planes, emb, context, vol, dout, B, H, W, D):
    """One level-0 output block ``TimestepEmbedSequential(ResBlock, SpatialTransformer, ViewAlignedFeatureTransformer)``
    (mvdfusion/unet.py:440-470) from its saved input ``cat`` = [h | skip]: the layer inputs are recomputed on the inference path
    (ctx still holds the step's time-embedding biases and cross-attention vectors), then the three block backwards run in reverse.
    Returns (dcat, {"0.": ..., "1.": ..., "2.": ...}-prefixed gradients, dcontext (B, 768), dvol, demb (1, ted))."""
    rb, st, vt = blk[0], blk[1], blk[2]
    r = rb.run(ctx, cat, H, W, x_planes=cat_planes).clone()
    s_ = st.run(ctx, r, H, W).clone()
    g = {}
    ds, gv, dvol = view_aligned_transformer_backward(tape, vt, s_, vol, dout, B, H, W, D)
    dr, gs, dcontext = spatial_transformer_backward(tape, st, r, context, ds, B, H, W)
    dcat, gr, demb = resblock_backward(tape, rb, cat, emb, dr, B, H, W)
    for pre, gg in (("0.", gr), ("1.", gs), ("2.", gv)):
        g.update({pre + k: v for k, v in gg.items()})
    return dcat, g, dcontext, dvol, demb
