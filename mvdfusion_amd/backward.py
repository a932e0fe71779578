"""Backward of the conv / linear / GroupNorm family on the HIP path -- the first slice of the training step
(reference train.py:90-95 ``loss.backward()``).

Every matrix product runs on the forward's split-operand MFMA GEMM (csrc/gemm.hip):

  dgrad  dX = dY W        mvd_gemm(planes(dY), packed W^T)                        (linear)
                          mvd_gemm conv mode over planes(dY) with the 180-degree rotated, channel-swapped filter   (3x3 / s1 / p1)
  wgrad  dW = dY^T X      mvd_gemm(A = planes(dY^T), B = MVD_B_PLANES planes(X^T))       -> (N, K)          nn.Linear.weight layout
                          ... B = planes(im2col(X)^T), rows ordered ci*9 + tap           -> (Cout, Cin*9)   nn.Conv2d.weight layout
  bgrad  db = column sums of dY (fp64 partials, fixed order)

plus ``GroupNorm(+SiLU)`` backward (csrc/backward.hip).  Weights change every optimizer step, so the transposed / rotated images
are packed per call (4 B per parameter, a fraction of the GEMM's own traffic).

These ops, LayerNorm / GEGLU / self-attention (fp32 matrix cores) / per-pixel cross-attention backward (op-level gradient tests against
torch autograd in fp32: tests/test_gpu_backward.py) are what the block-level backwards are made of: mvdfusion_amd/backward_blocks.py
(ResBlock, SpatialTransformer, ViewAlignedFeatureTransformer), backward_unet.py (the walk over the whole UNet, strided / upsampling
convolutions included) and backward_gridattn.py.  ``ViewFusion.forward(batch, cfg).backward()`` runs on them; all 994 parameter gradients
are pinned to the REFERENCE's ``loss.backward()`` (tests/golden/train_grads_*.npz, tests/test_gpu_vae.py).
"""
import os

import torch

from . import hip


def _pad32(n):
    return (n + 31) // 32 * 32


def _pad16(n):
    return (n + 15) // 16 * 16


def transpose_planes(x, rows, cols, src_planes=False, ldx=None, scale=None):
    """(rows, cols) matrix -- fp32 tensor, or split planes when `src_planes` -- to the split planes of its transpose:
    int16 (cols rounded up to 16, 2 * rows rounded up to 32); rows past `cols` stay zero (they are GEMM padding).  scale: device scalar an
    fp32 source is multiplied with on the way (the power-of-two gradient scale)."""
    ldo = _pad32(rows)
    # (the kernel writes every k-block of rows [0, cols), zeros included; only the GEMM padding rows [cols, ceil16(cols)) need a fill --
    #  it was a torch.zeros of the whole buffer: 600 fill launches per training step)
    out = torch.empty(_pad16(cols), 2 * ldo, dtype=torch.int16, device=x.device)
    if out.shape[0] > cols:
        out[cols:].zero_()
    if ldx is None:
        ldx = x.shape[-1] // 2 if src_planes else x.shape[-1]
    if scale is None:
        hip.check(hip.lib().mvd_transpose_planes(hip.ptr(x), int(bool(src_planes)), rows, cols, ldx, hip.ptr(out), ldo, hip.stream()))
    else:
        assert not src_planes
        hip.check(hip.lib().mvd_transpose_planes_scaled(hip.ptr(x), rows, cols, ldx, hip.ptr(out), ldo, hip.ptr(scale), hip.stream()))
    return out


def col_sum(x, rows, cols):
    out = torch.empty(cols, dtype=torch.float32, device=x.device)
    n = hip.lib().mvd_col_sum_workspace_doubles(rows, cols)
    ws = torch.empty(n, dtype=torch.float64, device=x.device)
    hip.check(hip.lib().mvd_col_sum(hip.ptr(x), rows, cols, x.shape[-1], hip.ptr(out), hip.ptr(ws), n, hip.stream()))
    return out


def act_planes(x, act, planes=True, f32=False):
    """act(x) of the fp32 matrix x (rows, cols) in one pass: (split planes (rows, 2 * ceil32(cols)) | None, fp32 (rows, cols) | None)
    (mvd_act_planes; act = hip.ACT_GELU / hip.ACT_SILU)."""
    x = x if x.is_contiguous() else x.contiguous()
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    ldp = _pad32(cols)
    sp = hip.planes_like(rows, ldp, x.device) if planes else None
    y = torch.empty(rows, cols, dtype=torch.float32, device=x.device) if f32 else None
    hip.check(hip.lib().mvd_act_planes(hip.ptr(x), hip.ptr(sp), hip.ptr(y), rows, cols, cols, ldp, cols, int(act), hip.stream()))
    return sp, y


def act_backward(dy, x, act):
    """dy * act'(x), elementwise, one pass (mvd_act_backward)."""
    dy = dy if dy.is_contiguous() else dy.contiguous()
    x = x if x.is_contiguous() else x.contiguous()
    assert dy.shape == x.shape and dy.dtype == x.dtype == torch.float32
    out = torch.empty_like(dy)
    hip.check(hip.lib().mvd_act_backward(hip.ptr(dy), hip.ptr(x), hip.ptr(out), dy.numel(), int(act), hip.stream()))
    return out


_POW2_SCRATCH = {}


def _pow2_scale(t):
    """Device scalars (s, 1/s), s the power of two that brings max|t| to [1024, 2048).  Gradients are small (1e-4 ... 1e-8): below 6e-5
    the fp16 hi + lo operand split only has an ABSOLUTE resolution of 2^-25 (include/mvd_hip.h, operand range contract), so they are
    scaled into the normal range before the split and the result is scaled back -- both exact (powers of two), both on the device:
    no host synchronisation.  One launch (mvd_pow2_scale; it was eight small torch kernels, 984 times per training step)."""
    t = t if t.is_contiguous() else t.contiguous()
    if t.data_ptr() % 16:           #  a contiguous slice at a storage offset that is not a multiple of 4 floats: the kernel reads float4
        t = t.clone()
    # the {running maximum, arrival counter} words are shared by the calls of ONE stream (the kernel re-zeroes them when it finishes and a
    # stream runs its kernels in order); another stream of the same device gets its own pair
    idx = t.device.index if t.device.index is not None else torch.cuda.current_device()
    try:
        key = (idx, torch._C._cuda_getCurrentRawStream(idx))          # (one C call: this runs ~1000 times per training step)
    except AttributeError:
        key = (idx, torch.cuda.current_stream(t.device).cuda_stream)
    scratch = _POW2_SCRATCH.get(key)
    if scratch is None:
        scratch = _POW2_SCRATCH[key] = torch.zeros(2, dtype=torch.int32, device=t.device)
    out = torch.empty(2, dtype=torch.float32, device=t.device)
    hip.check(hip.lib().mvd_pow2_scale(hip.ptr(t), t.numel(), hip.ptr(out), hip.ptr(scratch), hip.stream()))
    return out[0], out[1]


def _colsum_pow2_scale(dy, rows, cols):
    """(column sums of dy, s, 1/s): the bias gradient and the power-of-two operand scale of dy (see _pow2_scale) from ONE pass over dy
    (mvd_col_sum_pow2; round 6: they were two passes, ~330 launches and ~7 ms of the training step)."""
    assert dy.is_contiguous() and dy.shape[-1] == cols and dy.numel() == rows * cols
    idx = dy.device.index if dy.device.index is not None else torch.cuda.current_device()
    try:
        key = (idx, torch._C._cuda_getCurrentRawStream(idx), "cs")
    except AttributeError:
        key = (idx, torch.cuda.current_stream(dy.device).cuda_stream, "cs")
    scratch = _POW2_SCRATCH.get(key)
    if scratch is None:
        scratch = _POW2_SCRATCH[key] = torch.zeros(1, dtype=torch.int32, device=dy.device)
    out = torch.empty(cols, dtype=torch.float32, device=dy.device)
    out2 = torch.empty(2, dtype=torch.float32, device=dy.device)
    n = hip.lib().mvd_col_sum_workspace_doubles(rows, cols)
    ws = torch.empty(n, dtype=torch.float64, device=dy.device)
    hip.check(hip.lib().mvd_col_sum_pow2(hip.ptr(dy), rows, cols, cols, hip.ptr(out), hip.ptr(ws), n, hip.ptr(out2), hip.ptr(scratch), hip.stream()))
    return out, out2[0], out2[1]


FUSE_COLSUM_POW2 = os.environ.get("MVD_COLSUM_POW2", "1") != "0"      # (0: separate mvd_pow2_scale + mvd_col_sum passes, for A/B runs)


def _planes_padded(x, cols, scale=None):
    """fp32 (rows, cols) -> split planes (rows, 2 * ceil32(cols)) of x [* scale], padded columns zero."""
    return hip.split_planes(x.contiguous(), ldp=_pad32(cols), scale=scale)


def linear_backward(x_planes, weight, dy, workspace, need_dx=True, need_db=True, prec=hip.PREC_X4, need_dw=True):
    """y = x W^T + b  (nn.Linear / 1x1 conv).  x_planes: the forward's A operand, split planes (M, 2*ceil32(K)); weight (N, K) fp32;
    dy (M, N) fp32.  Returns (dx (M, K) | None, dW (N, K), db (N) | None)."""
    M, N = dy.shape
    K = weight.shape[1]
    dev = dy.device
    # dY * s on its way into the operand planes and 1 / s in the GEMM's accumulator scale: both exact (powers of two), no extra pass
    dyc = dy if dy.is_contiguous() else dy.contiguous()
    db = None
    if need_db and FUSE_COLSUM_POW2:
        db, sc, isc = _colsum_pow2_scale(dyc, M, N)
    else:
        sc, isc = _pow2_scale(dyc)
        db = col_sum(dyc, M, N) if need_db else None
    dx = None
    if need_dx:
        w2 = weight.detach().reshape(weight.shape[0], -1)                     # (a view for nn.Linear and 1x1 nn.Conv2d parameters)
        if w2.is_contiguous() and w2.dtype == torch.float32:
            wt = hip.pack_linear_t(w2, like=weight)                           # image of W^T (K, N): dX = dY W, no transposed copy
        else:
            wt = hip.pack_linear(w2.t().contiguous(), like=weight)   # (same elements: the registered max|w| serves)
        dx_full = torch.empty(M, wt.N, dtype=torch.float32, device=dev)
        hip.gemm(_planes_padded(dyc, N, scale=sc), wt, dx_full, prec=prec, bias=False, workspace=workspace, acc_scale_dev=isc)
        dx = dx_full[:, :K]
    dW = None
    if need_dw:                                                               # (frozen parameters: only the dgrad is needed)
        # dW = dY^T X : both operands activations, reduction over the M rows
        a = transpose_planes(dyc, M, N, scale=sc)                             # (ceil16(N), M) planes
        b = transpose_planes(x_planes, M, K, src_planes=True)                 # (ceil16(K), M) planes
        dw_full = torch.empty(a.shape[0], b.shape[0], dtype=torch.float32, device=dev)
        hip.gemm(a, hip.PlanesOperand(b, N=b.shape[0], K=_pad32(M)), dw_full, prec=prec, bias=False, workspace=workspace, acc_scale_dev=isc)
        dW = dw_full[:N, :K]
    return dx, dW, db


def conv3x3_backward(x_planes, weight, dy, B, H, W, workspace, need_dx=True, need_db=True, prec=hip.PREC_X4, need_dw=True):
    """y = conv3x3(x), stride 1, padding 1, channels-last.  x_planes: the forward's A operand (B*H*W, 2*ceil32(Cin)) split planes;
    weight (Cout, Cin, 3, 3) fp32; dy (B*H*W, Cout) fp32.  Returns (dx (M, Cin) | None, dW (Cout, Cin, 3, 3), db | None)."""
    M, Cout = dy.shape
    Cin = weight.shape[1]
    cin_p = x_planes.shape[-1] // 2
    dev = dy.device
    assert M == B * H * W and cin_p % 32 == 0 and cin_p >= Cin
    dyc = dy if dy.is_contiguous() else dy.contiguous()
    db = None
    if need_db and FUSE_COLSUM_POW2:   # (the scale is applied inside the plane conversions and the GEMMs' accumulator scale: linear_backward)
        db, sc, isc = _colsum_pow2_scale(dyc, M, Cout)
    else:
        sc, isc = _pow2_scale(dyc)
        db = col_sum(dyc, M, Cout) if need_db else None
    dx = None
    if need_dx:
        # dX = conv3x3(dY, W') with W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx]  (full correlation with the rotated filter)
        wr = hip.pack_conv3x3(weight.detach().flip(2, 3).transpose(0, 1).contiguous(), like=weight)
        dx_full = torch.empty(M, wr.N, dtype=torch.float32, device=dev)
        hip.gemm(_planes_padded(dyc, Cout, scale=sc), wr, dx_full, prec=prec, bias=False, workspace=workspace, acc_scale_dev=isc,
                 conv=dict(B=B, Hin=H, Win=W, Cin=_pad32(Cout), Hout=H, Wout=W, stride=1, upsample=0))
        dx = dx_full[:, :Cin]
    dW = None
    if need_dw:
        a = transpose_planes(dyc, M, Cout, scale=sc)                          # (ceil16(Cout), M)
        ldo = _pad32(M)
        cols_t = torch.empty(9 * cin_p, 2 * ldo, dtype=torch.int16, device=dev)     # (im2col X)^T, rows ci*9 + tap (every element written)
        hip.check(hip.lib().mvd_im2col3x3_t_planes(hip.ptr(x_planes), B, H, W, cin_p, hip.ptr(cols_t), ldo, hip.stream()))
        dw_full = torch.empty(a.shape[0], 9 * cin_p, dtype=torch.float32, device=dev)
        hip.gemm(a, hip.PlanesOperand(cols_t, N=9 * cin_p, K=ldo), dw_full, prec=prec, bias=False, workspace=workspace, acc_scale_dev=isc)
        dW = dw_full[:Cout, :9 * Cin].reshape(Cout, Cin, 3, 3)
    return dx, dW, db


def groupnorm_backward(x, dy, gamma, beta, B, HW, C, eps, silu, groups=32):
    """Backward of y = act(GroupNorm(x)), act = SiLU when `silu`.  x, dy: fp32 (B*HW, C).  Returns (dx, dgamma, dbeta)."""
    dev = x.device
    dx = torch.empty_like(x)
    dg = torch.empty(C, dtype=torch.float32, device=dev)
    db = torch.empty(C, dtype=torch.float32, device=dev)
    n = B * groups * 2 + B * C * 2
    ws = torch.empty(n, dtype=torch.float32, device=dev)
    hip.check(hip.lib().mvd_groupnorm_backward(hip.ptr(x), hip.ptr(dy.contiguous()), hip.ptr(gamma), hip.ptr(beta), B, HW, C, groups,
                                               float(eps), int(bool(silu)), hip.ptr(dx), hip.ptr(dg), hip.ptr(db), hip.ptr(ws), n,
                                               hip.stream()))
    return dx, dg, db


def layernorm_backward(x, dy, weight, eps):
    """Backward of y = LayerNorm(x) * weight + bias over the last dimension.  x, dy: fp32 (rows, C).  Returns (dx, dweight, dbias)."""
    rows, C = x.shape
    dx, t = torch.empty_like(x), torch.empty_like(x)
    dy = dy.contiguous()
    hip.check(hip.lib().mvd_layernorm_backward(hip.ptr(x), hip.ptr(dy), hip.ptr(weight), rows, C, float(eps), hip.ptr(dx), hip.ptr(t),
                                               hip.stream()))
    return dx, col_sum(t, rows, C), col_sum(dy, rows, C)


def geglu_backward(h, dy):
    """Backward of GEGLU y = a * gelu(g), [a | g] = h (rows, 2*half): dh (rows, 2*half)."""
    rows, half = dy.shape
    assert h.shape == (rows, 2 * half)
    dh = torch.empty_like(h)
    hip.check(hip.lib().mvd_geglu_backward(hip.ptr(h), hip.ptr(dy.contiguous()), rows, half, hip.ptr(dh), hip.stream()))
    return dh


def attention_backward(q, k, v, dout, B, heads, L, dhead):
    """Backward of softmax(Q K^T / sqrt(d)) V per (batch, head); all tensors token-major fp32 (B*L, heads*dhead).
    Returns (dq, dk, dv)."""
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    n = B * heads * L * 3
    stats = torch.empty(n, dtype=torch.float32, device=q.device)
    hip.check(hip.lib().mvd_attention_backward(hip.ptr(q), hip.ptr(k), hip.ptr(v), hip.ptr(dout.contiguous()), B, heads, L, dhead,
                                               hip.ptr(dq), hip.ptr(dk), hip.ptr(dv), hip.ptr(stats), n, hip.stream()))
    return dq, dk, dv


def pixel_cross_attn_backward(q, k, v, dout, P, D, heads, dhead):
    """Backward of the per-pixel cross attention over D context tokens (mvd_pixel_cross_attn): q, dout (P, C); k, v (P*D, C)."""
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    hip.check(hip.lib().mvd_pixel_cross_attn_backward(hip.ptr(q), hip.ptr(k), hip.ptr(v), hip.ptr(dout.contiguous()), P, D, heads, dhead,
                                                      hip.ptr(dq), hip.ptr(dk), hip.ptr(dv), hip.stream()))
    return dq, dk, dv


def unet_head_backward(unet, h, a_planes, pred_rows, target_rows, B, S, workspace):
    """MSE(pred, target) through the UNet output head ``out = [GroupNorm32, SiLU, conv3x3]`` (mvdfusion/unet.py:496-500).

    h: fp32 (B*S*S, mc) input of the head (output of the last output block); a_planes: split planes of SiLU(GN(h)) kept by the
    forward; pred_rows / target_rows: (B*S*S, Cout) channels-last prediction and target.  Returns ({parameter name: gradient},
    dh) with the names relative to the UNetModel (``out.0.weight`` ...) and dh = dL/dh, where the backward currently stops."""
    gn, conv = unet.out[0], unet.out[2]
    M, Cout = pred_rows.shape
    dy = (pred_rows - target_rows) * (2.0 / pred_rows.numel())                # d mean((pred - target)^2) / d pred
    da, dW, db = conv3x3_backward(a_planes, conv.weight, dy.contiguous(), B, S, S, workspace)
    dh, dg, dbeta = groupnorm_backward(h, da.contiguous(), gn.weight, gn.bias, B, S * S, h.shape[-1], gn.eps, True)
    return {"out.2.weight": dW, "out.2.bias": db, "out.0.weight": dg, "out.0.bias": dbeta}, dh
