"""Backward kernels of the conv / linear / GroupNorm family (mvdfusion_amd/backward.py, csrc/backward.hip) against torch autograd
in fp32 on the same seeded inputs: dgrad and wgrad run on the split-operand MFMA GEMM (f16x4: ~2^-22 operand error), the bias
gradient and GroupNorm backward in fp32 / fp64 VALU.  The chain through the real UNet head is pinned to the REFERENCE's gradients in
tests/test_gpu_vae.py::test_training_head_gradients_vs_reference_golden."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import planes_to_float, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from mvdfusion_amd import hip as h
    h.lib()
    return h


def g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("rows,cols", [(64, 32), (100, 48), (4096, 5), (33, 320)])
def test_transpose_planes(hip, rows, cols):
    from mvdfusion_amd import backward as bw
    x = torch.randn(rows, cols, generator=g(1)) * 3
    xd = x.cuda()
    t = bw.transpose_planes(xd, rows, cols)
    rp = (rows + 31) // 32 * 32
    got = planes_to_float(t)
    assert got.shape == ((cols + 15) // 16 * 16, rp)
    assert rel_err(got[:cols, :rows], x.t()) < 2e-6 and float(got[:cols, rows:].abs().max() if rp > rows else 0.0) == 0.0
    assert float(got[cols:].abs().max() if got.shape[0] > cols else 0.0) == 0.0
    # from split planes: a pure re-arrangement of the hi / lo halves (bit-exact against the fp32 route)
    xp = hip.split_planes(xd)
    t2 = bw.transpose_planes(xp, rows, cols, src_planes=True)
    assert torch.equal(t2.cpu(), t.cpu())


@pytest.mark.parametrize("M,N,K", [(1024, 320, 640), (512, 5, 96), (200, 48, 32), (4096, 96, 320)])
def test_linear_backward(hip, M, N, K):
    from mvdfusion_amd import backward as bw
    x = torch.randn(M, K, generator=g(2), requires_grad=True)
    w = (torch.randn(N, K, generator=g(3)) / math.sqrt(K)).requires_grad_()
    b = torch.randn(N, generator=g(4), requires_grad=True)
    dy = torch.randn(M, N, generator=g(5))
    F.linear(x, w, b).backward(dy)
    ws = torch.empty(16 * 1024 * 1024, device="cuda")
    xp = hip.split_planes(x.detach().cuda())
    dx, dW, db = bw.linear_backward(xp, w.detach().cuda(), dy.cuda(), ws)
    assert rel_err(dx.cpu(), x.grad) < 3e-6
    assert rel_err(dW.cpu(), w.grad) < 3e-6
    assert rel_err(db.cpu(), b.grad) < 2e-6


def test_backward_gradients_far_below_fp16_normal_range(hip):
    """Real gradients are 1e-5 ... 1e-8: below 6e-5 the fp16 operand split only resolves 2^-25 absolutely, so the backward scales dY by
    a power of two before the split (exactly undone by the GEMM's accumulator scale).  Same relative accuracy as at unit scale."""
    from mvdfusion_amd import backward as bw
    M, N, K = 1024, 96, 320
    x = torch.randn(M, K, generator=g(2), requires_grad=True)
    w = (torch.randn(N, K, generator=g(3)) / math.sqrt(K)).requires_grad_()
    dy = torch.randn(M, N, generator=g(5)) * 3e-7
    F.linear(x, w).backward(dy)
    ws = torch.empty(16 * 1024 * 1024, device="cuda")
    dx, dW, _ = bw.linear_backward(hip.split_planes(x.detach().cuda()), w.detach().cuda(), dy.cuda(), ws, need_db=False)
    assert rel_err(dx.cpu(), x.grad) < 3e-6 and rel_err(dW.cpu(), w.grad) < 3e-6
    B, H, Cin, Cout = 2, 16, 64, 32
    xc = torch.randn(B, Cin, H, H, generator=g(6), requires_grad=True)
    wc = (torch.randn(Cout, Cin, 3, 3, generator=g(7)) / math.sqrt(9 * Cin)).requires_grad_()
    dyc = torch.randn(B, Cout, H, H, generator=g(9)) * 3e-7
    F.conv2d(xc, wc, None, padding=1).backward(dyc)
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(B * H * H, -1).contiguous()
    dxc, dWc, _ = bw.conv3x3_backward(hip.split_planes(rows(xc.detach()).cuda()), wc.detach().cuda(), rows(dyc).cuda(), B, H, H, ws,
                                      need_db=False)
    assert rel_err(dxc.cpu(), rows(xc.grad)) < 3e-6 and rel_err(dWc.cpu(), wc.grad) < 3e-6


@pytest.mark.parametrize("B,H,Cin,Cout", [(2, 16, 64, 96), (4, 32, 32, 5), (1, 8, 320, 64), (2, 8, 10, 32)])
def test_conv3x3_backward(hip, B, H, Cin, Cout):
    from mvdfusion_amd import backward as bw
    x = torch.randn(B, Cin, H, H, generator=g(6), requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g(7)) / math.sqrt(9 * Cin)).requires_grad_()
    b = torch.randn(Cout, generator=g(8), requires_grad=True)
    dy = torch.randn(B, Cout, H, H, generator=g(9))
    F.conv2d(x, w, b, padding=1).backward(dy)
    ws = torch.empty(16 * 1024 * 1024, device="cuda")
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(B * H * H, -1).contiguous()
    xp = hip.split_planes(rows(x.detach()).cuda())          # channels padded to 32 with zeros
    dx, dW, db = bw.conv3x3_backward(xp, w.detach().cuda(), rows(dy).cuda(), B, H, H, ws)
    assert dW.shape == w.shape
    assert rel_err(dx.cpu(), rows(x.grad)) < 3e-6
    assert rel_err(dW.cpu(), w.grad) < 3e-6
    assert rel_err(db.cpu(), b.grad) < 2e-6


@pytest.mark.parametrize("B,HW,C,silu,eps", [(4, 1024, 320, True, 1e-5), (2, 256, 64, False, 1e-6), (3, 64, 1280, True, 1e-5),
                                             (4, 1024, 32, True, 1e-5)])
def test_groupnorm_backward(hip, B, HW, C, silu, eps):
    from mvdfusion_amd import backward as bw
    x = (torch.randn(B, HW, C, generator=g(10)) * 2 + 0.5).requires_grad_()
    gm = torch.randn(C, generator=g(11), requires_grad=True)
    bt = torch.randn(C, generator=g(12), requires_grad=True)
    dy = torch.randn(B, HW, C, generator=g(13))
    y = F.group_norm(x.permute(0, 2, 1), 32, gm, bt, eps=eps).permute(0, 2, 1)
    (F.silu(y) if silu else y).backward(dy)
    dx, dg, db = bw.groupnorm_backward(x.detach().reshape(B * HW, C).cuda(), dy.reshape(B * HW, C).cuda(), gm.detach().cuda(),
                                       bt.detach().cuda(), B, HW, C, eps, silu)
    assert rel_err(dx.cpu().view(B, HW, C), x.grad) < 5e-6
    assert rel_err(dg.cpu(), gm.grad) < 5e-6
    assert rel_err(db.cpu(), bt.grad) < 5e-6
    # deterministic: a second evaluation is bit-identical
    dx2, dg2, db2 = bw.groupnorm_backward(x.detach().reshape(B * HW, C).cuda(), dy.reshape(B * HW, C).cuda(), gm.detach().cuda(),
                                          bt.detach().cuda(), B, HW, C, eps, silu)
    assert torch.equal(dx, dx2) and torch.equal(dg, dg2) and torch.equal(db, db2)


@pytest.mark.parametrize("rows,C", [(1000, 320), (64, 1280), (4096, 32), (37, 640)])
def test_layernorm_backward(hip, rows, C):
    from mvdfusion_amd import backward as bw
    x = (torch.randn(rows, C, generator=g(20)) * 3 + 1).requires_grad_()
    w = torch.randn(C, generator=g(21), requires_grad=True)
    b = torch.randn(C, generator=g(22), requires_grad=True)
    dy = torch.randn(rows, C, generator=g(23))
    F.layer_norm(x, (C,), w, b, eps=1e-5).backward(dy)
    dx, dw, db = bw.layernorm_backward(x.detach().cuda(), dy.cuda(), w.detach().cuda(), 1e-5)
    assert rel_err(dx.cpu(), x.grad) < 5e-6
    assert rel_err(dw.cpu(), w.grad) < 5e-6
    assert rel_err(db.cpu(), b.grad) < 2e-6


def test_geglu_backward(hip):
    from mvdfusion_amd import backward as bw
    rows, half = 512, 1280
    h = (torch.randn(rows, 2 * half, generator=g(24)) * 2).requires_grad_()
    dy = torch.randn(rows, half, generator=g(25))
    a, gt = h.chunk(2, dim=-1)
    (a * F.gelu(gt)).backward(dy)
    dh = bw.geglu_backward(h.detach().cuda(), dy.cuda())
    assert rel_err(dh.cpu(), h.grad) < 3e-6


@pytest.mark.parametrize("B,H,L,d", [(2, 8, 256, 40), (1, 8, 1024, 4), (2, 4, 64, 160), (2, 8, 100, 80), (1, 8, 1024, 40), (3, 2, 300, 16), (5, 8, 15, 32),
                                      (1, 3, 130, 160)])
def test_attention_backward(hip, B, H, L, d):
    from mvdfusion_amd import backward as bw
    C = H * d
    q, k, v = [(torch.randn(B * L, C, generator=g(26 + i)) * (1.5 if i < 2 else 1.0)).requires_grad_() for i in range(3)]
    dout = torch.randn(B * L, C, generator=g(30))
    heads = lambda t: t.view(B, L, H, d).permute(0, 2, 1, 3)
    o = torch.softmax(heads(q) @ heads(k).transpose(-1, -2) * d ** -0.5, dim=-1) @ heads(v)
    o.permute(0, 2, 1, 3).reshape(B * L, C).backward(dout)
    dq, dk, dv = bw.attention_backward(q.detach().cuda(), k.detach().cuda(), v.detach().cuda(), dout.cuda(), B, H, L, d)
    assert rel_err(dq.cpu(), q.grad) < 1e-5
    assert rel_err(dk.cpu(), k.grad) < 1e-5
    assert rel_err(dv.cpu(), v.grad) < 1e-5


@pytest.mark.parametrize("D", [1, 3, 8])
def test_pixel_cross_attn_backward(hip, D):
    from mvdfusion_amd import backward as bw
    P, H, d = 300, 8, 40
    C = H * d
    q = torch.randn(P, C, generator=g(31), requires_grad=True)
    k = torch.randn(P * D, C, generator=g(32), requires_grad=True)
    v = torch.randn(P * D, C, generator=g(33), requires_grad=True)
    dout = torch.randn(P, C, generator=g(34))
    qh = q.view(P, 1, H, d).permute(0, 2, 1, 3)
    kh, vh = k.view(P, D, H, d).permute(0, 2, 1, 3), v.view(P, D, H, d).permute(0, 2, 1, 3)
    o = torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, dim=-1) @ vh
    o.permute(0, 2, 1, 3).reshape(P, C).backward(dout)
    dq, dk, dv = bw.pixel_cross_attn_backward(q.detach().cuda(), k.detach().cuda(), v.detach().cuda(), dout.cuda(), P, D, H, d)
    assert rel_err(dq.cpu(), q.grad) < 5e-6
    assert rel_err(dk.cpu(), k.grad) < 5e-6
    assert rel_err(dv.cpu(), v.grad) < 5e-6


def test_pow2_scale_one_launch(hip):
    """mvd_pow2_scale against the torch formula it replaced (max|t| -> power of two into [1024, 2048); 1 for zero / non-finite input)."""
    from mvdfusion_amd import backward as bw

    def ref(t):
        mx = torch.linalg.vector_norm(t.reshape(-1), ord=float("inf")).float()
        ok = torch.isfinite(mx) & (mx > 0)
        e = torch.floor(torch.log2(torch.where(ok, mx, torch.ones_like(mx))))
        return torch.where(ok, torch.exp2(10.0 - e), torch.ones_like(mx))

    cases = [torch.randn(1000, 37, generator=g(1)) * 1e-6, torch.randn(7, generator=g(2)) * 3e-5, torch.randn(1 << 20, generator=g(3)) * 40.0,
             torch.zeros(513), torch.full((5,), float("nan")), torch.tensor([1.0, float("inf"), -2.0]), torch.tensor([-1024.0]),
             torch.tensor([2047.99, -1.0e-30]), torch.randn(4097, generator=g(4)) * 2.0 ** -60]
    for i, t in enumerate(cases * 2):            # twice: the scratch words must come back zeroed
        td = t.cuda()
        s, inv = bw._pow2_scale(td)
        want = float(ref(td))
        assert float(s) == want and float(inv) == 1.0 / want, (i, float(s), want)
        if want != 1.0 or i % len(cases) == 6:
            assert 1024.0 <= float(td.abs().max()) * float(s) < 2048.0
    x = torch.randn(300, 300, generator=g(9)).cuda()[:, :299]          # a non-contiguous view is reduced over its own elements only
    assert float(bw._pow2_scale(x)[0]) == float(ref(x.contiguous()))


def test_hip_adamw_matches_torch_adamw(hip):
    """mvdfusion_amd.optim.HipAdamW (one mvd_adamw_multi launch) against torch.optim.AdamW on the same parameters and gradients over
    several steps, odd sizes and a second parameter group; state dicts interchange (train.py:150,178)."""
    from mvdfusion_amd.optim import HipAdamW
    shapes = [(5,), (4097,), (320, 320, 3, 3), (1280, 77), (3,), (8192,)]
    ps = [torch.nn.Parameter((torch.randn(*s, generator=g(20 + i)) * 0.05).cuda()) for i, s in enumerate(shapes)]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]

    def groups(lst):
        return [{"params": lst[:4], "lr": 1e-3}, {"params": lst[4:], "lr": 3e-4, "weight_decay": 0.1}]

    a, b = HipAdamW(groups(ps), lr=1e-3), torch.optim.AdamW(groups(qs), lr=1e-3)
    assert isinstance(a, torch.optim.AdamW)
    for step in range(4):
        for i, (p, q) in enumerate(zip(ps, qs)):
            gr = (torch.randn(*p.shape, generator=g(100 + 10 * step + i)) * 1e-3).cuda()
            p.grad, q.grad = gr.clone(), gr.clone()
        if step == 2:
            ps[1].grad = None                    # a parameter without a gradient this step is skipped (its step count stays behind)
            qs[1].grad = None
        a.step()
        b.step()
        for p, q in zip(ps, qs):
            assert float((p - q).detach().abs().max()) <= 2e-7 * max(1.0, float(q.detach().abs().max())), step
    sa, sb = a.state_dict(), b.state_dict()
    assert sa["param_groups"][1]["weight_decay"] == sb["param_groups"][1]["weight_decay"] and len(sa["state"]) == len(sb["state"])
    for k in sb["state"]:
        assert float(sa["state"][k]["step"]) == float(sb["state"][k]["step"])
        assert float((sa["state"][k]["exp_avg_sq"] - sb["state"][k]["exp_avg_sq"]).abs().max()) <= 1e-12
    import copy
    a2 = HipAdamW(groups(ps), lr=1e-3)
    a2.load_state_dict(copy.deepcopy(sb))         # a torch.optim.AdamW checkpoint resumes under HipAdamW (state_dict() hands out references)
    v0 = ps[0]._version
    for p, q in zip(ps, qs):
        gr = torch.ones_like(p) * 1e-3
        p.grad, q.grad = gr.clone(), gr.clone()
    a2.step()
    b.step()
    assert ps[0]._version > v0                     # the raw-pointer update is visible to version-keyed caches (packed weight images)
    for p, q in zip(ps, qs):
        assert float((p - q).detach().abs().max()) <= 4e-7 * max(1.0, float(q.detach().abs().max()))


@pytest.mark.parametrize("act", ["gelu", "silu"])
@pytest.mark.parametrize("rows,cols", [(1000, 1024), (37, 50), (4096, 256)])
def test_activation_forward_planes_and_backward(hip, act, rows, cols):
    """mvd_act_planes (act(x) as operand planes and / or fp32, one pass) and mvd_act_backward (dy * act'(x)): the training step's GELU / SiLU
    passes (view_attn_efficient2.py:42-67 Mlp / pre_layer_b) against torch (erf GELU; autograd for the derivative), ragged column counts."""
    from mvdfusion_amd import backward as bw
    from mvdfusion_amd import hip as H
    code = H.ACT_GELU if act == "gelu" else H.ACT_SILU
    fn = F.gelu if act == "gelu" else F.silu
    x = (torch.randn(rows, cols, generator=g(7)) * 2.5).requires_grad_(True)
    dy = torch.randn(rows, cols, generator=g(8))
    y = fn(x)
    y.backward(dy)
    sp, yf = bw.act_planes(x.detach().cuda(), code, planes=True, f32=True)
    assert rel_err(yf, y) < 2e-6
    got = planes_to_float(sp)
    pl = 2e-5 if H.OPERAND_FORMAT == "bf16" else 1e-6
    assert rel_err(got[:, :cols], y) < 2e-6 + pl and (got.shape[1] == cols or float(got[:, cols:].abs().max()) == 0.0)
    only_planes, none = bw.act_planes(x.detach().cuda(), code)
    assert none is None and torch.equal(only_planes, sp)
    dx = bw.act_backward(dy.cuda(), x.detach().cuda(), code)
    assert rel_err(dx, x.grad) < 3e-6


@pytest.mark.parametrize("N,K", [(320, 640), (1280, 320), (48, 200), (5, 96)])
def test_pack_linear_transposed_source_is_bit_identical(hip, N, K):
    """mvd_pack_linear_weight_t (round 6: the dgrad weight W^T packed straight from the parameter) writes the SAME image as packing an
    explicit transposed copy -- same elements, same power-of-two scale, same micro-tile layout, zero padding included."""
    w = (torch.randn(N, K, generator=g(70)) * 0.05).cuda()
    a = hip.pack_linear(w.t().contiguous(), like=w)          # image of W^T (K rows, N columns) from a transposed copy
    b = hip.pack_linear_t(w, like=w)                         # ... straight from W
    assert (a.N, a.K, a.n_real, a.acc_scale) == (b.N, b.K, b.n_real, b.acc_scale)
    assert torch.equal(a.data, b.data)


@pytest.mark.parametrize("rows,cols", [(4096, 320), (37, 50), (1024, 5), (8192, 1280)])
def test_col_sum_pow2_equals_the_two_passes(hip, rows, cols):
    """mvd_col_sum_pow2 (round 6): bias gradient and power-of-two operand scale of a dY from one pass == mvd_col_sum and mvd_pow2_scale,
    bit for bit (the same fp64 partials in the same order; the same maximum)."""
    from mvdfusion_amd import backward as bw
    dy = (torch.randn(rows, cols, generator=g(71)) * 3e-6).cuda()
    db, sc, isc = bw._colsum_pow2_scale(dy, rows, cols)
    sc2, isc2 = bw._pow2_scale(dy)
    assert torch.equal(db, bw.col_sum(dy, rows, cols))
    assert float(sc) == float(sc2) and float(isc) == float(isc2) and 1024.0 <= float(dy.abs().max()) * float(sc) < 2048.0
    assert rel_err(db.cpu(), dy.double().sum(0).float().cpu()) < 1e-5
    # the scratch word is left zero: a second call gives the same answer
    db3, sc3, _ = bw._colsum_pow2_scale(dy * 0.5, rows, cols)
    assert float(sc3) == 2.0 * float(sc)
