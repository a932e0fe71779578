"""Depth-conditioned cross-view aggregation, executed by HIP kernels -- mirror of
``mvdfusion.view_attn_efficient2.GridAttn`` (mvdfusion/view_attn_efficient2.py:96-442).

Same constructor kwargs and state_dict keys (``z_embedder.0``, ``t_embedder.mlp.{0,2}`` [present but unused, as in the
reference], ``pre_layer_b.0``, ``aggregation_transformer.layer_list.{i}.{attn.qkv,attn.proj,mlp.fc1,mlp.fc2,
adaLN_modulation.1}``, ``aggregation_transformer.weight_layer``, ``final_layer_b``).

Per step (one launch each unless noted):
  z_embed (x2) -> fused depth-sample/unproject/reproject/gather/embed token kernel -> pre_layer GEMM(+GELU)
  -> 3 x [adaLN GEMV, LN+modulate, QKV GEMM, attention over the V views, proj GEMM (+gate +residual),
          LN+modulate, fc1 GEMM(+GELU), fc2 GEMM (+gate +residual)]
  -> weight-softmax pooling over V -> final GEMM -> (V, S, S, D, 768) feature frustum.
"""
import torch
import torch.nn as nn

from . import hip
from .cameras import pack_cameras


# ------------------------------------------------------------------------------------------------
# weight stream of the fused aggregation kernel (csrc/gridattn_fused.hip)
# ------------------------------------------------------------------------------------------------
_G4_VEC_BLOCK, _G4_VEC_MISC = 3328, 3 * 3328


_MT_INDEX = {}


def _microtile_index(dev):
    """Index tensors of _microtiles on `dev` (built once per device)."""
    t = _MT_INDEX.get(str(dev))
    if t is None:
        perm = torch.tensor([(4 * c + j) if j < 4 else (16 + 4 * c + j - 4) for c in range(4) for j in range(8)], device=dev)
        R, c8 = torch.arange(16, device=dev), torch.arange(8, device=dev)
        gran = (R >> 3)[:, None].expand(16, 8)
        pos = ((R & 7) * 8)[:, None] + (c8[None, :] ^ ((R >> 1) & 7)[:, None])
        t = _MT_INDEX[str(dev)] = (perm, gran, pos)
    return t


def _microtiles(W, scale, bf16):
    """(N, K) fp32 -> (N/16, K/32, 1024) int16: every (16 weight rows x 32 k) micro-tile as its 2 KiB LDS image --
    x*scale ~= hi + lo in the MFMA operand type, k inside the 32-block permuted to the fragment order of the kernel's
    register-resident activations (slot (c, j): k = 4c + j for j < 4, 16 + 4c + j - 4 otherwise), 16-byte chunk c (0-3 hi,
    4-7 lo) of row R at position (R & 7) * 8 + (c ^ ((R >> 1) & 7)) of granule R >> 3 (the swizzle of csrc/gemm_device.hpp).
    Runs on W's device with torch indexing only -- no host copy: a training step re-packs the stream after every optimizer step."""
    N, K = W.shape
    assert N % 16 == 0 and K % 32 == 0, (N, K)
    w = (W.detach().double() * scale).float()
    dt = torch.bfloat16 if bf16 else torch.float16
    hi = w.to(dt)
    lo = (w - hi.float()).to(dt)
    perm, gran, pos = _microtile_index(W.device)

    def chunks(t):
        return t.contiguous().view(torch.int16).view(N // 16, 16, K // 32, 32)[..., perm].reshape(N // 16, 16, K // 32, 4, 8)

    data = torch.cat([chunks(hi), chunks(lo)], dim=3).permute(0, 2, 1, 3, 4)            # (nt, ks, R, c8, 8)
    img = torch.empty(N // 16, K // 32, 2, 64, 8, dtype=torch.int16, device=W.device)
    img[:, :, gran, pos, :] = data
    return img.reshape(N // 16, K // 32, 1024)


def pack_fused_stream(ga, bf16=False):
    """GridAttn parameters -> (stream int16 (215 * 16, 1024) = 215 slots of 32 KiB in the kernel's consumption order,
    vecs fp32 (11264,) with everything but the per-step adaLN modulation filled in), both on the parameters' device.  The only host
    values are the power-of-two scales, which come from the parameter-maximum registry when it knows the tensors (hip._pack_scale(like=):
    training) and from one reduction per weight otherwise (inference packs once)."""
    C = ga.hidden_size
    blocks = list(ga.aggregation_transformer.layer_list)
    assert C == 256 and len(blocks) == 3 and all(b.num_heads == 8 and b.mlp.fc1.out_features == 512 for b in blocks)
    dev = ga.pre_layer_b[0].weight.device
    vecs = torch.zeros(hip.lib().mvd_gridattn_fused_vec_floats(), dtype=torch.float32, device=dev)
    tiles = []

    def scaled(Wt, like):
        sc = hip._pack_scale(Wt.detach(), like=like)
        return _microtiles(Wt, sc, bf16), 1.0 / sc

    def put(o, t):
        t = t.detach().float().reshape(-1)
        vecs[o:o + t.numel()] = t

    wpre = torch.zeros(C, 736, device=dev)
    wpre[:, :723] = ga.pre_layer_b[0].weight.detach().float()
    pre, s_pre = scaled(wpre, ga.pre_layer_b[0].weight)
    tiles.append(pre.permute(1, 0, 2).reshape(-1, 1024))                                # for ks: for nt
    put(_G4_VEC_MISC, ga.pre_layer_b[0].bias)
    wl = ga.aggregation_transformer.weight_layer
    put(_G4_VEC_MISC + 256, wl.weight)
    put(_G4_VEC_MISC + 512, wl.bias)
    host = {_G4_VEC_MISC + 520: s_pre}
    for bi, blk in enumerate(blocks):
        q, s_q = scaled(blk.attn.qkv.weight.detach().float(), blk.attn.qkv.weight)          # (48, 8, 1024)
        pj, s_p = scaled(blk.attn.proj.weight.detach().float(), blk.attn.proj.weight)      # (16, 8, 1024)
        f1, s_1 = scaled(blk.mlp.fc1.weight.detach().float(), blk.mlp.fc1.weight)          # (32, 8, 1024)
        f2, s_2 = scaled(blk.mlp.fc2.weight.detach().float(), blk.mlp.fc2.weight)          # (16, 16, 1024)
        for hd in range(8):
            rows = [2 * hd, 2 * hd + 1, 16 + 2 * hd, 16 + 2 * hd + 1, 32 + 2 * hd, 32 + 2 * hd + 1]
            tiles.append(q[rows].permute(1, 0, 2).reshape(-1, 1024))                    # mt = 6 ks + tile
            tiles.append(pj[:, hd])                                                     # 16 output tiles of k-step hd
        for ch in range(8):
            tiles.append(f1[4 * ch:4 * ch + 4].permute(1, 0, 2).reshape(-1, 1024))      # for ks: the chunk's 4 tiles
            tiles.append(f2[:, 2 * ch:2 * ch + 2].permute(1, 0, 2).reshape(-1, 1024))   # for u: for nt
        o = bi * _G4_VEC_BLOCK
        put(o + 1536, blk.attn.qkv.bias)
        put(o + 2304, blk.attn.proj.bias)
        put(o + 2560, blk.mlp.fc1.bias)
        put(o + 3072, blk.mlp.fc2.bias)
        for k, v in enumerate((s_q, s_p, s_1, s_2)):
            host[_G4_VEC_MISC + 521 + 4 * bi + k] = v
    idx = sorted(host)                                                                  # the scales: one small host -> device copy
    vecs[torch.tensor(idx, device=dev)] = torch.tensor([host[k] for k in idx], dtype=torch.float32).to(dev)
    stream = torch.cat(tiles, 0).contiguous()
    assert stream.shape[0] == 16 * hip.lib().mvd_gridattn_fused_slots(), stream.shape
    return stream, vecs


class _TimmAttention(nn.Module):      # parameter holder: timm.models.vision_transformer.Attention(qkv_bias=True)
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class _TimmMlp(nn.Module):            # parameter holder: timm Mlp(fc1, GELU, fc2)
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class DiTBlock(nn.Module):
    """adaLN-Zero block (view_attn_efficient2.py:42-67); LN eps 1e-6 without affine."""

    def __init__(self, hidden_size, num_heads, cond_dim=None, mlp_ratio=4.0):
        super().__init__()
        cond_dim = hidden_size if cond_dim is None else cond_dim
        self.attn = _TimmAttention(hidden_size, num_heads)
        self.mlp = _TimmMlp(hidden_size, int(hidden_size * mlp_ratio))
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(cond_dim, 6 * hidden_size, bias=True))
        self.hidden_size, self.num_heads = hidden_size, num_heads
        self._p = None

    def packed(self):
        if self._p is None:
            self._p = (hip.pack_linear(self.attn.qkv.weight, self.attn.qkv.bias),
                       hip.pack_linear(self.attn.proj.weight, self.attn.proj.bias),
                       hip.pack_linear(self.mlp.fc1.weight, self.mlp.fc1.bias),
                       hip.pack_linear(self.mlp.fc2.weight, self.mlp.fc2.bias))
        return self._p

    def run(self, ctx, h, h_alt, c, T, V):
        """h (T, C) -> returns the updated stream (written back into h); h_alt is the ping-pong partner."""
        C = self.hidden_size
        wqkv, wproj, wfc1, wfc2 = self.packed()
        lin = self.adaLN_modulation[1]
        mod = ctx.ws.get("ga.mod", (1, 6 * C))
        hip.gemv(lin.weight, lin.bias, c, mod, act_in=hip.ACT_SILU)
        sh1, sc1, g1, sh2, sc2, g2 = (mod[0, i * C:(i + 1) * C] for i in range(6))
        ln = ctx.ws.planes("ga.ln", T, C)
        hip.layernorm(h, ln, sc1, sh1, T, C, eps=1e-6, w_plus_one=True)
        qkv = ctx.ws.get("ga.qkv", (T, 3 * C))
        ctx.gemm(ln, wqkv, qkv, kind="ga")
        att = ctx.ws.planes("ga.att", T, C)
        hip.check(hip.lib().mvd_view_mha(hip.ptr(qkv), hip.ptr(att), T // V, V, self.num_heads, C // self.num_heads,
                                         hip.stream()))
        ctx.gemm(att, wproj, h_alt, res=h, colscale=g1, kind="ga")
        hip.layernorm(h_alt, ln, sc2, sh2, T, C, eps=1e-6, w_plus_one=True)
        f1 = ctx.ws.planes("ga.f1", T, wfc1.N)
        ctx.gemm(ln, wfc1, None, act=hip.ACT_GELU, out_planes=f1, kind="ga")
        ctx.gemm(f1, wfc2, h, res=h_alt, colscale=g2, kind="ga")
        return h


class AggregationTransformer(nn.Module):
    def __init__(self, hidden_size, num_layers=3, num_heads=8, mlp_ratio=2.0, use_t=False):
        super().__init__()
        if not use_t:
            raise NotImplementedError
        self.use_t = use_t
        self.layer_list = nn.ModuleList([DiTBlock(hidden_size, num_heads=num_heads, mlp_ratio=mlp_ratio)
                                         for _ in range(num_layers)])
        self.weight_layer = nn.Linear(hidden_size, 1)


class _TimestepEmbedder(nn.Module):   # holder for the unused-but-present t_embedder.mlp.{0,2} parameters
    def __init__(self, hidden_size, frequency_embedding_size=256):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(frequency_embedding_size, hidden_size, bias=True), nn.SiLU(),
                                 nn.Linear(hidden_size, hidden_size, bias=True))


class GridAttn(nn.Module):
    def __init__(self, input_size=32, in_channels=4, hidden_size=256, output_dim=768, num_heads=8, mlp_ratio=2.0,
                 num_layers=3, side_length=32, world_scale=0.6, z_near_far_scale=0.8, depth_scale=2.0,
                 depth_shift=0.5, n_pts_per_ray=3, use_t=True, keep_top_k_views=False, top_k=4, device="cpu"):
        super().__init__()
        assert not keep_top_k_views, "top-k view selection is dead code in the reference configs"
        assert hidden_size == 256, "the token kernel is specialised for 256-channel feature maps"
        assert in_channels == 5, "mvd_zembed reads 4 VAE + 1 depth latent channels (configs/*.yaml: in_channels: 5)"
        self.input_size, self.hidden_size, self.output_dim = input_size, hidden_size, output_dim
        self.depth_scale, self.depth_shift, self.n_pts_per_ray = depth_scale, depth_shift, n_pts_per_ray
        self.z_near_far_scale = z_near_far_scale
        self.z_embedder = nn.Sequential(nn.Linear(in_channels, 256), nn.GELU())
        self.t_embedder = _TimestepEmbedder(hidden_size)
        self.use_t = use_t
        self.pre_layer_b = nn.Sequential(nn.Linear(256 * 2 + 90 * 2 + 15 * 2 + 1, hidden_size), nn.GELU())
        self.aggregation_transformer = AggregationTransformer(hidden_size, num_layers, num_heads, mlp_ratio, use_t)
        self.final_layer_b = nn.Linear(hidden_size, output_dim)
        for blk in self.aggregation_transformer.layer_list:      # adaLN-Zero init (:174-176)
            nn.init.constant_(blk.adaLN_modulation[-1].weight, 0)
            nn.init.constant_(blk.adaLN_modulation[-1].bias, 0)
        self._p = None
        self._fused = None

    def packed(self):
        if self._p is None:
            self._p = (hip.pack_linear(self.pre_layer_b[0].weight, self.pre_layer_b[0].bias),
                       hip.pack_linear(self.final_layer_b.weight, self.final_layer_b.bias))
        return self._p

    def packed_fused(self, device):
        """(weight stream, vecs) of the fused aggregation kernel on `device` (packed once; the adaLN part of vecs is
        rewritten every step)."""
        if self._fused is None:
            stream, vecs = pack_fused_stream(self, bf16=hip.OPERAND_FORMAT == "bf16")
            self._fused = (stream.to(device), vecs.to(device))
        return self._fused

    def fused_supported(self, V, T):
        """Whether the single-launch aggregation kernel serves V reference views / T = nseq * V tokens: any 1 <= V <= 16 (the kernel
        pads the views of a 3-D point to the next power of two and masks the padding: the reference's 15 / 7 / 5 views included)."""
        blocks = self.aggregation_transformer.layer_list
        Vp = 1 << max(int(V) - 1, 0).bit_length()
        return (1 <= V <= 16 and (T // V * Vp) % 64 == 0 and T % V == 0 and self.hidden_size == 256 and len(blocks) == 3 and
                all(b.num_heads == 8 and b.mlp.fc1.out_features == 512 for b in blocks))

    def run(self, ctx, x, depth_noise, steps, it, cams_rec, in_cam_rec, input_latents, c, vol_out, V, S, D, q0=0, Vq=None,
            vol_planes=None, vol_planes_col=0, fused=None, depth_src=None, depth_steps=None):
        """x (V,5,S,S) noisy latents; c (1,256) time conditioning (t_embed[:1]); vol_out: (>=V*S*S*D, 768) buffer
        whose first Vq*S*S*D rows receive the feature frustum (row = ((v*S + y)*S + x)*D + d) of the query views
        [q0, q0+Vq) (all V views by default; a view-parallel rank passes the range it owns).  vol_planes: optional planes
        buffer receiving the frustum as well, in columns [vol_planes_col, vol_planes_col + 768) of its rows.
        fused: None = the single-launch aggregation kernel (mvd_gridattn_fused) whenever V <= 16, else the unfused chain
        of token kernel + GEMMs; True / False force one of them.
        depth_src / depth_steps (overwrite_attn_depth, view_attn_efficient2.py:418-426): a (V,5,S,S) buffer whose channel 4 is the depth
        map to sample around INSTEAD of the x0-style estimate x[:,4] / sqrt(alpha_bar), with a step table whose sqrt(alpha_bar) column
        is 1 (x / 1 is exact) and whose depth-std column is unchanged -- the kernels themselves are the same."""
        Vq = V if Vq is None else Vq
        L = hip.lib()
        assert x.shape[1] == 5, "depth wise efficient attention requires 4+1 channels"
        w_pre, w_fin = self.packed()
        z = self.z_embedder[0]
        feat = ctx.ws.get("ga.feat", (V, S, S, 256))
        in_feat = ctx.ws.get("ga.infeat", (1, S, S, 256))
        hip.check(L.mvd_zembed(hip.ptr(x), hip.ptr(z.weight), hip.ptr(z.bias), hip.ptr(feat), V, S, hip.stream()))
        hip.check(L.mvd_zembed(hip.ptr(input_latents), hip.ptr(z.weight), hip.ptr(z.bias), hip.ptr(in_feat), 1, S,
                               hip.stream()))
        nseq = Vq * S * S * D
        T = nseq * V
        dsrc = x if depth_src is None else depth_src
        dsteps = steps if depth_steps is None else depth_steps
        assert (depth_src is None) == (depth_steps is None) and dsrc.shape == x.shape
        grid_lin = ctx.ws.bufs.get(("ga.lin", S))
        if grid_lin is None:
            half = 1.0 / float(S)
            grid_lin = torch.linspace(1.0 - half, -1.0 + half, S, dtype=torch.float32).to(ctx.device)
            ctx.ws.bufs[("ga.lin", S)] = grid_lin
        if fused is None:
            fused = self.fused_supported(V, T)
        if fused:
            assert self.fused_supported(V, T), (V, T)
            stream, vecs = self.packed_fused(ctx.device)
            for bi, blk in enumerate(self.aggregation_transformer.layer_list):      # adaLN modulation of this step -> vecs
                lin = blk.adaLN_modulation[1]
                hip.gemv(lin.weight, lin.bias, c, vecs[bi * _G4_VEC_BLOCK:bi * _G4_VEC_BLOCK + 1536].view(1, 1536), act_in=hip.ACT_SILU)
            pool = ctx.ws.planes("ga.pool", nseq, self.hidden_size)
            hip.check(L.mvd_gridattn_fused(hip.ptr(dsrc), hip.ptr(depth_noise), hip.ptr(dsteps), hip.ptr(it), hip.ptr(grid_lin),
                                           hip.ptr(feat), hip.ptr(in_feat), hip.ptr(cams_rec), hip.ptr(in_cam_rec), hip.ptr(stream),
                                           hip.ptr(vecs), hip.ptr(pool), V, q0, Vq, S, D, float(self.depth_scale),
                                           float(self.depth_shift), 3 if ctx.prec_of("ga") == 3 else 4, hip.stream()))
            ctx.gemm(pool, w_fin, vol_out, M=nseq, out_planes=vol_planes, out_planes_col=vol_planes_col, kind="ga")
            return vol_out
        tokens = ctx.ws.planes("ga.tokens", T, hip.TOKEN_LD)
        hip.check(L.mvd_gridattn_tokens(hip.ptr(dsrc), hip.ptr(depth_noise), hip.ptr(dsteps), hip.ptr(it), hip.ptr(grid_lin),
                                        hip.ptr(feat), hip.ptr(in_feat), hip.ptr(cams_rec), hip.ptr(in_cam_rec),
                                        hip.ptr(tokens), V, q0, Vq, S, D, float(self.depth_scale), float(self.depth_shift),
                                        hip.stream()))
        h = ctx.ws.get("ga.h", (T, self.hidden_size))
        h_alt = ctx.ws.get("ga.h_alt", (T, self.hidden_size))
        ctx.gemm(tokens, w_pre, h, act=hip.ACT_GELU, kind="ga")
        for blk in self.aggregation_transformer.layer_list:
            h = blk.run(ctx, h, h_alt, c, T, V)
        wl = self.aggregation_transformer.weight_layer
        pool = ctx.ws.planes("ga.pool", nseq, self.hidden_size)
        hip.check(L.mvd_view_pool(hip.ptr(h), hip.ptr(wl.weight), hip.ptr(wl.bias), hip.ptr(pool), nseq, V, self.hidden_size,
                                  hip.stream()))
        # the frustum is consumed as fp32 (area pooling) and as planes (level-0 to_k / to_v GEMMs): write both
        ctx.gemm(pool, w_fin, vol_out, M=nseq, out_planes=vol_planes, out_planes_col=vol_planes_col, kind="ga")
        return vol_out
