"""Per-kernel parity: every HIP op (called through the C ABI) against a plain fp32 PyTorch-CPU statement of the same
op.  Tolerances (relative to the max magnitude of the reference tensor):
  exact-fp32 VALU kernels ................ 2e-6
  bf16x3 MFMA kernels (~16-bit operands) . 3e-5
  bf16 MFMA kernels ...................... 2e-2
"""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import cfg_splitk_matrix, load_golden, planes_to_float, rel_err

pytestmark = pytest.mark.gpu

from mvdfusion_amd import hip as _hip
_BF = _hip.OPERAND_FORMAT == "bf16"
TOL = {1: 2e-2 if _BF else 3e-3, 3: 3e-5 if _BF else 3e-6, 4: 3e-5 if _BF else 2e-6}   # 1 product: 8 / 11 operand bits; x3/x4: ~16 / ~22 bits
PL = 2e-5 if _BF else 1e-6  # representation error of a value stored as split planes (~2^-17 bf16, ~2^-22 fp16)


@pytest.fixture(scope="module")
def hip():
    from mvdfusion_amd import hip as h
    assert h.lib().mvd_version() == 100
    return h


def g(seed):
    return torch.Generator().manual_seed(seed)


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("prec", [4, 3, 1])
@pytest.mark.parametrize("M,N,K", [(2048, 320, 320), (128, 1280, 2560), (100, 48, 96), (4096, 256, 736), (64, 16, 32)])
def test_gemm_dense(hip, prec, M, N, K):
    A = torch.randn(M, K, generator=g(1))
    W = torch.randn(N, K, generator=g(2)) / math.sqrt(K)
    b = torch.randn(N, generator=g(3))
    R = torch.randn(M, N, generator=g(4))
    ref = F.linear(A, W, b) + R
    Wp = hip.pack_linear(W.cuda(), b.cuda())
    out = torch.empty(M, N, device="cuda")
    ws = torch.empty(8 * 1024 * 1024, device="cuda")
    Ap, Rc = hip.split_planes(A.cuda()), R.cuda()
    hip.gemm(Ap, Wp, out, prec=prec, res=Rc, workspace=ws)
    assert rel_err(out, ref) < TOL[prec]
    # forced split-K and no split must agree with the reference as well
    for sk in (1, 4):
        out.zero_()
        hip.gemm(Ap, Wp, out, prec=prec, res=Rc, workspace=ws, splitk=sk)
        assert rel_err(out, ref) < TOL[prec], sk
    # every tile shape (plain loop, n-fastest order)
    for cfg in hip.GEMM_CONFIGS[::2]:       # every tile shape and loop variant (n-fastest order)
        out.zero_()
        hip.gemm(Ap, Wp, out, prec=prec, res=Rc, workspace=ws, splitk=1, cfg=cfg)
        assert rel_err(out, ref) < TOL[prec], cfg
    # loop variants removed in round 5 (include/mvd_hip.h: cfg) fail loudly instead of running some other kernel
    for loop in hip.REMOVED_LOOPS:
        with pytest.raises(RuntimeError, match="does not serve"):
            hip.gemm(Ap, Wp, out, prec=prec, res=Rc, workspace=ws, splitk=1, cfg=hip.make_cfg(1, loop))
    # plane output (feeds the next GEMM)
    if N % 32 == 0:
        op = hip.planes_like(M, N, "cuda")
        hip.gemm(Ap, Wp, None, prec=prec, res=Rc, workspace=ws, out_planes=op)
        assert rel_err(planes_to_float(op), ref) < TOL[prec] + PL
        # ... into columns [32, 32 + N) of a wider operand buffer (operand concatenation along K), fp32 output alongside
        wide = hip.split_planes(torch.full((M, N + 96), -3.0, device="cuda"))
        out.zero_()
        hip.gemm(Ap, Wp, out, prec=prec, res=Rc, workspace=ws, out_planes=wide, out_planes_col=32)
        got = planes_to_float(wide)
        assert rel_err(got[:, 32:32 + N], ref) < TOL[prec] + PL and rel_err(out, ref) < TOL[prec]
        assert bool((got[:, :32] == -3.0).all()) and bool((got[:, 32 + N:] == -3.0).all())


@pytest.mark.skipif(_BF, reason="the bf16 flavour has fp32's exponent range")
def test_gemm_operand_range_contract(hip):
    """include/mvd_hip.h "Operand range": operands are fp16 hi + lo.  In range (|x| < 65504) the relative operand error
    is ~2^-22 down to |x| ~ 6e-2 and the ABSOLUTE resolution is 2^-25 ~ 3e-8 below that; past 65504 the hi plane is inf and
    the result is non-finite, which hip.check_finite turns into an error at the end of a sample."""
    M, N, K = 256, 64, 320
    W = torch.randn(N, K, generator=g(5)) / math.sqrt(K)
    Wp = hip.pack_linear(W.cuda())
    out = torch.empty(M, N, device="cuda")
    base = torch.randn(M, K, generator=g(6))
    for scale, tol in ((3.0e4 / 4.5, 3e-6), (1.0, 3e-6), (1e-2, 3e-6)):      # max |A| ~ 4.5 * scale: up to ~3e4, well in range
        A = base * scale
        hip.gemm(hip.split_planes(A.cuda()), Wp, out, prec=4)
        assert rel_err(out, F.linear(A.double(), W.double()).float()) < tol, scale
    # tiny activations: error bounded by the absolute resolution of the split (2^-25 per element, summed over K with |w| ~ K^-1/2)
    A = base * 1e-6
    hip.gemm(hip.split_planes(A.cuda()), Wp, out, prec=4)
    ref = F.linear(A.double(), W.double())
    assert float((out.double().cpu() - ref).abs().max()) < 2.0 ** -25 * math.sqrt(K) * 4, "absolute floor of the fp16 split"
    # out of range: documented to overflow to non-finite, and detected
    A = base.clone()
    A[3, 7] = 1.0e5
    hip.gemm(hip.split_planes(A.cuda()), Wp, out, prec=4)
    assert not bool(torch.isfinite(out[3]).all())
    with pytest.raises(FloatingPointError):
        hip.check_finite(out, "range test")
    assert bool(torch.isfinite(out[4:]).all())                                # other rows are unaffected


@pytest.mark.parametrize("M,N,K", [(1024, 1024, 512), (512, 1024, 512), (200, 48, 96)])
def test_gemm_activation_b_operand(hip, M, N, K):
    """MVD_B_PLANES: out = A @ B^T with B an ACTIVATION in split planes (no packed weight): the VAE mid-block attention's
    Q K^T / P V, including a B operand that is a column view of a wider planes buffer and every tile shape."""
    A = torch.randn(M, K, generator=g(11))
    Bm = torch.randn(N, K, generator=g(12))
    bias = torch.randn(N, generator=g(13))
    ref = A @ Bm.t() * 0.5 + bias
    Ap = hip.split_planes(A.cuda())
    wide = hip.split_planes(torch.cat([torch.randn(N, 64, generator=g(14)), Bm], 1).cuda())      # B lives in columns [64, 64 + K)
    Bop = hip.PlanesOperand(wide[:, 2 * 64:], N=N, K=K, bias=bias.cuda(), acc_scale=0.5, ld=64 + K)
    out = torch.empty(M, N, device="cuda")
    ws = torch.empty(8 * 1024 * 1024, device="cuda")
    hip.gemm(Ap, Bop, out, workspace=ws)
    assert rel_err(out, ref) < TOL[4]
    for cfg in hip.GEMM_CONFIGS[::2]:
        for sk in (1, 2):
            out.zero_()
            hip.gemm(Ap, Bop, out, workspace=ws, cfg=cfg, splitk=sk)
            assert rel_err(out, ref) < TOL[4], (cfg, sk)


def test_gemm_epilogues(hip):
    M, C = 512, 320
    A = torch.randn(M, C, generator=g(5))
    W = torch.randn(C, C, generator=g(6)) / math.sqrt(C)
    b = torch.randn(C, generator=g(7))
    gate = torch.randn(C, generator=g(8))
    R = torch.randn(M, C, generator=g(9))
    bb = torch.randn(4, C, generator=g(10))
    Wp = hip.pack_linear(W.cuda(), b.cuda())
    out = torch.empty(M, C, device="cuda")
    ws = torch.empty(4 * 1024 * 1024, device="cuda")
    Ap, Rc, gc, bbc = hip.split_planes(A.cuda()), R.cuda(), gate.cuda(), bb.cuda()
    # gate * (acc + bias) + residual  (adaLN gate)
    hip.gemm(Ap, Wp, out, res=Rc, colscale=gc, workspace=ws)
    assert rel_err(out, R + gate * F.linear(A, W, b)) < TOL[4]
    # GELU / SiLU
    hip.gemm(Ap, Wp, out, act=hip.ACT_GELU, workspace=ws)
    assert rel_err(out, F.gelu(F.linear(A, W, b))) < TOL[4]
    hip.gemm(Ap, Wp, out, act=hip.ACT_SILU, workspace=ws)
    assert rel_err(out, F.silu(F.linear(A, W, b))) < TOL[4]
    # per-batch bias vector (kv_len == 1 cross attention)
    hip.gemm(Ap, Wp, out, bias_b=bbc, rows_per_batch=M // 4, workspace=ws)
    assert rel_err(out, F.linear(A, W, b) + bb.repeat_interleave(M // 4, 0)) < TOL[4]


# one configuration per tile shape (+ the role-split kernel): the tile epilogue is shared code, its chunk mapping depends on the wave tile
_EPI_CFGS = [_hip.make_cfg(0, 4), _hip.make_cfg(1, 0), _hip.make_cfg(1, 2), _hip.make_cfg(2, 4), _hip.make_cfg(3, 4), _hip.make_cfg(4, 4),
             _hip.make_cfg(2, _hip.WS_LOOP), _hip.make_cfg(4, _hip.WS_LOOP)]


@pytest.mark.parametrize("cfg", _EPI_CFGS)
@pytest.mark.parametrize("M,N", [(200, 328), (300, 50), (96, 320), (37, 4)])
def test_gemm_store_epilogue_matrix(hip, cfg, M, N):
    """Every option of the store epilogue (bias, per-view bias, activation, column scale, residual, fp32 and / or planes output) on ragged
    shapes: rows that end inside a wave tile, a column count that is not a multiple of 4 / 16 / the tile width (element path),
    per-view bias with FEWER rows per view than a wave tile -- for every wave-tile geometry."""
    K = 96
    A = torch.randn(M, K, generator=g(11))
    W = torch.randn(N, K, generator=g(12)) / math.sqrt(K)
    b = torch.randn(N, generator=g(13))
    R = torch.randn(M, N, generator=g(14))
    gate = torch.randn(N, generator=g(15))
    rpb = 8
    nb = (M + rpb - 1) // rpb
    bb = torch.randn(nb, N, generator=g(16))
    bbr = bb.repeat_interleave(rpb, 0)[:M]
    Ap = hip.split_planes(A.cuda())
    ws = torch.empty(4 * 1024 * 1024, device="cuda")
    Wb, W0 = hip.pack_linear(W.cuda(), b.cuda()), hip.pack_linear(W.cuda())
    lin, lin0 = F.linear(A, W, b), F.linear(A, W)
    Np = (N + 3) // 4 * 4                      # row pitch of fp32 outputs / residuals: a multiple of 4 (16-byte rows)
    out = torch.empty(M, Np, device="cuda")
    Rc = torch.zeros(M, Np, device="cuda")
    Rc[:, :N] = R.cuda()
    bbc = torch.zeros(nb, Np, device="cuda")
    bbc[:, :N] = bb.cuda()
    cases = [
        (dict(res=Rc), Wb, lin + R),
        (dict(), W0, lin0),
        (dict(bias_b=bbc, rows_per_batch=rpb), Wb, lin + bbr),
        (dict(bias_b=bbc, rows_per_batch=rpb, res=Rc, colscale=gate.cuda(), act=hip.ACT_QUICKGELU), Wb,
         R + gate * ((lin + bbr) * torch.sigmoid(1.702 * (lin + bbr)))),
        (dict(act=hip.ACT_SILU, colscale=gate.cuda()), Wb, gate * F.silu(lin)),
        (dict(act=hip.ACT_GELU, res=Rc), Wb, R + F.gelu(lin)),
    ]
    for kw, Wp, ref in cases:
        out.fill_(-7.0)
        hip.gemm(Ap, Wp, out, workspace=ws, cfg=cfg, splitk=1, **kw)
        assert rel_err(out[:, :N], ref) < TOL[4], (cfg, sorted(kw))
        assert bool((out[:, N:] == -7.0).all()), "columns past N must not be written"
    if N % 32 == 0:      # planes output next to / instead of the fp32 output
        op = hip.planes_like(M, N, "cuda")
        out.fill_(-7.0)
        hip.gemm(Ap, Wb, out, workspace=ws, cfg=cfg, splitk=1, res=Rc, out_planes=op)
        assert rel_err(out, lin + R) < TOL[4] and rel_err(planes_to_float(op), lin + R) < TOL[4] + PL
        op2 = hip.planes_like(M, N, "cuda")
        hip.gemm(Ap, Wb, None, workspace=ws, cfg=cfg, splitk=1, bias_b=bbc, rows_per_batch=rpb, out_planes=op2)
        assert rel_err(planes_to_float(op2), lin + bbr) < TOL[4] + PL


@pytest.mark.parametrize("splitk", [0, 1, 3])
def test_gemm_geglu(hip, splitk):
    M, C = 256, 64
    A = torch.randn(M, C, generator=g(11))
    W = torch.randn(8 * C, C, generator=g(12)) / math.sqrt(C)
    b = torch.randn(8 * C, generator=g(13))
    h = F.linear(A, W, b)
    a, gt = h.chunk(2, dim=-1)
    ref = a * F.gelu(gt)
    Wp = hip.pack_linear(W.cuda(), b.cuda(), geglu=True)
    out = torch.empty(M, 4 * C, device="cuda")
    ws = torch.empty(4 * 1024 * 1024, device="cuda")
    # GEGLU bias is addressed by logical column: pass the unpermuted bias
    Ap = hip.split_planes(A.cuda())
    hip.gemm(Ap, Wp, out, epi=hip.EPI_GEGLU, workspace=ws, splitk=splitk)
    assert rel_err(out, ref) < TOL[3]
    op = hip.planes_like(M, 4 * C, "cuda")
    hip.gemm(Ap, Wp, None, epi=hip.EPI_GEGLU, workspace=ws, splitk=splitk, out_planes=op)
    assert rel_err(planes_to_float(op), ref) < TOL[3] + PL
    # every kernel configuration that serves the GEGLU epilogue (incl. the wave-specialised kernel: 64 x 64 consumer tiles = two 32-column blocks)
    base = None
    for cfg in hip.gemm_configs(hip.EPI_GEGLU)[::2]:
        out.fill_(float("nan"))
        hip.gemm(Ap, Wp, out, epi=hip.EPI_GEGLU, workspace=ws, splitk=splitk, cfg=cfg)
        assert rel_err(out, ref) < TOL[3], cfg
        if splitk == 1:
            base = out.clone() if base is None else base
            assert torch.equal(out, base), cfg


@pytest.mark.parametrize("prec", [4, 3, 1])
@pytest.mark.parametrize("case", ["s1", "s2", "up", "smallM", "stem", "head"])
def test_conv3x3(hip, prec, case):
    B, H, Cin, Cout, stride, up = {"s1": (4, 16, 64, 96, 1, 0), "s2": (2, 16, 64, 64, 2, 0), "up": (2, 8, 64, 64, 1, 1),
                                   "smallM": (2, 4, 2560, 1280, 1, 0), "stem": (2, 32, 10, 320, 1, 0),
                                   "head": (2, 32, 320, 5, 1, 0)}[case]
    x = torch.randn(B, Cin, H, H, generator=g(20))
    w = torch.randn(Cout, Cin, 3, 3, generator=g(21)) / math.sqrt(9 * Cin)
    b = torch.randn(Cout, generator=g(22))
    xi = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    ref = F.conv2d(xi, w, b, stride=stride, padding=1)
    Ho = ref.shape[-1]
    cin_pad = (Cin + 31) // 32 * 32
    xn = torch.zeros(B, H, H, cin_pad)
    xn[..., :Cin] = x.permute(0, 2, 3, 1)
    Wp = hip.pack_conv3x3(w.cuda(), b.cuda())
    ldo = 8 if Cout < 8 else Cout
    out = torch.zeros(B * Ho * Ho, ldo, device="cuda")
    ws = torch.empty(16 * 1024 * 1024, device="cuda")
    xp = hip.split_planes(xn.view(-1, cin_pad).cuda())
    hip.gemm(xp, Wp, out, prec=prec, workspace=ws, ldo=ldo,
             conv=dict(B=B, Hin=H, Win=H, Cin=cin_pad, Hout=Ho, Wout=Ho, stride=stride, upsample=up))
    got = out[:, :Cout].view(B, Ho, Ho, Cout).permute(0, 3, 1, 2)
    assert rel_err(got, ref) < TOL[prec]
    # the input-patch kernel (stride-1 convolutions): every precision, bias, ragged N; bit-equal to the implicit-GEMM kernel
    served = 0
    for cfg in hip.PATCH_CONFIGS[::2]:
        out2 = torch.zeros_like(out)
        try:
            hip.gemm(xp, Wp, out2, prec=prec, workspace=ws, ldo=ldo, cfg=cfg, splitk=1,
                     conv=dict(B=B, Hin=H, Win=H, Cin=cin_pad, Hout=Ho, Wout=Ho, stride=stride, upsample=up))
        except RuntimeError as e:
            assert "does not serve" in str(e)
            continue
        served += 1
        base = torch.zeros_like(out)
        hip.gemm(xp, Wp, base, prec=prec, workspace=ws, ldo=ldo, cfg=1, splitk=1,
                 conv=dict(B=B, Hin=H, Win=H, Cin=cin_pad, Hout=Ho, Wout=Ho, stride=stride, upsample=up))
        assert torch.equal(out2, base), (case, cfg)
    assert served == (0 if case in ("s2", "up") else len(hip.PATCH_CONFIGS[::2])), (case, served)


_CASES = {}


def _case(key, make):
    """CPU operands / fp32 references of the configuration matrices, built once per module instead of once per parametrisation."""
    if key not in _CASES:
        _CASES[key] = make()
    return _CASES[key]


@pytest.mark.parametrize("cfg,splitk", cfg_splitk_matrix(_hip.GEMM_CONFIGS_CONV))
def test_gemm_configurations_agree(hip, cfg, splitk):
    """Every kernel configuration the autotuner may pick (tile x loop variant x tile order, include/mvd_hip.h `cfg`), with and
    without split-K, on conv and dense problems with even and odd k-tile counts: all must match the fp32 reference AND be
    repeatable bit for bit (a stale-accumulator miscompile of one instantiation showed up exactly here)."""
    ws = torch.empty(16 * 1024 * 1024, device="cuda")
    # nk = 18, 27, 90, 18, 9; the input-patch kernel (hip.PATCH_CONFIGS) meets tiles of 4 rows of one image, half an image, 2 images (one
    # past M), 8 whole 4x4 images and 2 rows of a 64-wide image (the last two need two patch DMAs per wave and k-tile at 128x80)
    for B, H, Cin, Cout in [(2, 32, 64, 64), (2, 16, 96, 320), (1, 8, 320, 48), (11, 4, 64, 80), (1, 64, 32, 80)]:
        def make_conv():
            x = torch.randn(B, Cin, H, H, generator=g(40))
            w = torch.randn(Cout, Cin, 3, 3, generator=g(41)) / math.sqrt(9 * Cin)
            ref = F.conv2d(x, w, None, padding=1).permute(0, 2, 3, 1).reshape(B * H * H, Cout)
            return ref, hip.pack_conv3x3(w.cuda(), None), hip.split_planes(x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().cuda())
        ref, Wp, xp = _case(("conv", B, H, Cin, Cout), make_conv)
        outs = []
        for rep in range(3):
            out = torch.full((B * H * H, Cout), float("nan"), device="cuda")
            hip.gemm(xp, Wp, out, prec=4, workspace=ws, cfg=cfg, splitk=splitk,
                     conv=dict(B=B, Hin=H, Win=H, Cin=Cin, Hout=H, Wout=H, stride=1, upsample=0))
            outs.append(out.cpu())
        assert rel_err(outs[0], ref) < TOL[4], (B, H, Cin, Cout)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        if splitk == 1:      # without split-K every tile / loop variant sums each output's k-tiles in the same order: bit-equal to cfg 1
            def make_base():
                base = torch.empty(B * H * H, Cout, device="cuda")
                hip.gemm(xp, Wp, base, prec=4, workspace=ws, cfg=1, splitk=1,
                         conv=dict(B=B, Hin=H, Win=H, Cin=Cin, Hout=H, Wout=H, stride=1, upsample=0))
                return base.cpu()
            assert torch.equal(_case(("conv-base", B, H, Cin, Cout), make_base), outs[0]), (B, H, Cin, Cout)
    for M, N, K in [(2048, 320, 320), (512, 640, 2592), (100, 48, 96)]:                   # nk = 10, 81, 3
        if cfg in hip.PATCH_CONFIGS:
            break                                                                          # (convolutions only)
        def make_dense():
            a = torch.randn(M, K, generator=g(42))
            w = torch.randn(N, K, generator=g(43)) / math.sqrt(K)
            return a @ w.t(), hip.pack_linear(w.cuda(), None), hip.split_planes(a.cuda())
        ref, Wp, ap = _case(("dense", M, N, K), make_dense)
        outs = []
        for rep in range(3):
            out = torch.full((M, N), float("nan"), device="cuda")
            hip.gemm(ap, Wp, out, prec=4, workspace=ws, cfg=cfg, splitk=splitk)
            outs.append(out.cpu())
        assert rel_err(outs[0], ref) < TOL[4], (M, N, K)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        if splitk == 1:
            def make_dbase():
                base = torch.empty(M, N, device="cuda")
                hip.gemm(ap, Wp, base, prec=4, workspace=ws, cfg=1, splitk=1)
                return base.cpu()
            assert torch.equal(_case(("dense-base", M, N, K), make_dbase), outs[0]), (M, N, K)


@pytest.mark.parametrize("cfg,splitk", cfg_splitk_matrix([0] + list(_hip.gemm_configs(_hip.EPI_STORE)), (0, 1, 3)))
def test_gemm_groupnorm_statistics(hip, cfg, splitk):
    """mvd_gemm_desc.gn_stats: the GEMM (tile epilogue or split-K reduce) emits the GroupNorm statistics of its output as integer
    atomics; mvd_groupnorm_from_stats must then reproduce F.group_norm of the stored output, and the statistics must be identical
    bit for bit across repeats (integer accumulation is order independent)."""
    ws = torch.empty(16 * 1024 * 1024, device="cuda")
    for B, HW, N, K in [(2, 256, 320, 320), (3, 64, 640, 1344), (2, 16, 1280, 96), (1, 1024, 96, 64), (4, 16, 32, 32)]:
        M = B * HW

        def make_gn():
            a = torch.randn(M, K, generator=g(50)) + 0.1
            w = torch.randn(N, K, generator=g(51)) / math.sqrt(K)
            b = torch.randn(N, generator=g(52))
            r = torch.randn(M, N, generator=g(53))
            gm, bt = torch.randn(N, generator=g(54)), torch.randn(N, generator=g(55))
            return (a @ w.t() + b + r, gm, bt, hip.pack_linear(w.cuda(), b.cuda()), hip.split_planes(a.cuda()), r.cuda(), gm.cuda(), bt.cuda())
        lin, gm, bt, Wp, ap, rc, gc, bc = _case(("gn", B, HW, N, K), make_gn)
        stats = []
        for rep in range(2):
            out = torch.full((M, N), float("nan"), device="cuda")
            st = torch.zeros(B, 32, 2, dtype=torch.int64, device="cuda")
            hip.gemm(ap, Wp, out, prec=4, workspace=ws, cfg=cfg, splitk=splitk, res=rc, gn_stats=st, gn_hw=HW)
            stats.append(st.cpu())
        assert torch.equal(stats[0], stats[1])
        assert rel_err(out, lin) < TOL[4]
        x = out.cpu().view(B, HW, N)
        xg = x.double().view(B, HW, 32, N // 32)
        sums = torch.stack([xg.sum(dim=(1, 3)), (xg * xg).sum(dim=(1, 3))], -1)
        mags = torch.stack([xg.abs().sum(dim=(1, 3)), (xg * xg).sum(dim=(1, 3))], -1)          # fp32 partial sums: error ~ sum |x|
        got = stats[0].double() / 2.0 ** 24
        assert float(((got - sums).abs() / (1.0 + mags)).max()) < 2e-6, (B, HW, N, K)
        y = hip.planes_like(M, N, "cuda")
        hip.groupnorm_from_stats(out, y, gc, bc, st, B, HW, N, 1e-5, True)
        ref = F.silu(F.group_norm(x.permute(0, 2, 1), 32, gm, bt, eps=1e-5).permute(0, 2, 1))
        assert rel_err(planes_to_float(y).view(B, HW, N), ref) < PL + 3e-6, (B, HW, N, K)


@pytest.mark.parametrize("splitk", [0, 1, 2, 5])
@pytest.mark.parametrize("flags", [1, 0, 3, 5])
def test_gemm_groupnorm_apply_behind_the_gemm(hip, splitk, flags):
    """mvd_gemm_desc.gna_out_sp: GroupNorm (+ SiLU) of the output applied behind the GEMM -- ONE reduce + apply kernel per (image, group)
    when the GEMM splits K and the group's slab fits the LDS (splitk_gn_kernel), otherwise producer statistics + the apply kernel launched
    by the library.  Every path must reproduce F.group_norm of the reference output, leave the fp32 output and the statistics slot of
    the output valid (another consumer may normalise it), and be repeatable bit for bit.  Shapes: a 3x3 convolution with a per-image
    bias (ResBlock conv1 + time embedding, openaimodel.py:262-270), dense GEMMs with a residual, groups of 10 / 20 / 40 / 2 channels,
    one slab (4096 x 10 floats = 160 KB) that does not fit -> fallback."""
    ws = torch.empty(32 * 1024 * 1024, device="cuda")
    cases = [("conv", 2, 16, 64, 320), ("conv", 3, 64, 96, 640), ("dense", 2, 256, 320, 1280), ("dense", 1, 1024, 96, 64), ("dense", 1, 4096, 64, 320),
             ("dense", 4, 16, 160, 64)]
    for kind, B, HW, K, N in cases:
        M = B * HW
        gm, bt = torch.randn(N, generator=g(64)) + 1.0, torch.randn(N, generator=g(65))
        bb = torch.randn(N, generator=g(66))
        if kind == "conv":
            H = int(math.isqrt(HW))
            x = torch.randn(B, K, H, H, generator=g(60)) + 0.2
            w = torch.randn(N, K, 3, 3, generator=g(61)) / math.sqrt(9 * K)
            ref = F.conv2d(x, w, None, padding=1).permute(0, 2, 3, 1).reshape(M, N) + bb
            Wp = hip.pack_conv3x3(w.cuda(), None)
            ap = hip.split_planes(x.permute(0, 2, 3, 1).reshape(M, K).contiguous().cuda())
            kw = dict(conv=dict(B=B, Hin=H, Win=H, Cin=K, Hout=H, Wout=H, stride=1, upsample=0), bias=False, bias_b=bb.cuda(), rows_per_batch=M)
        else:
            a = torch.randn(M, K, generator=g(60)) + 0.1
            w = torch.randn(N, K, generator=g(61)) / math.sqrt(K)
            b = torch.randn(N, generator=g(62))
            r = torch.randn(M, N, generator=g(63))
            ref = a @ w.t() + b + r
            Wp = hip.pack_linear(w.cuda(), b.cuda())
            ap = hip.split_planes(a.cuda())
            kw = dict(res=r.cuda())
        want = F.group_norm(ref.view(B, HW, N).permute(0, 2, 1), 32, gm, bt, eps=1e-5).permute(0, 2, 1).reshape(M, N)
        if flags & 2:
            want = want.half().float()
        if flags & 1:
            want = F.silu(want)
        gc, bc = gm.cuda(), bt.cuda()
        runs = []
        for rep in range(2):
            out = torch.full((M, N), float("nan"), device="cuda")
            y = hip.planes_like(M, N, "cuda")
            y.fill_(0x7e00)
            st = torch.zeros(B, 32, 2, dtype=torch.int64, device="cuda")
            hip.gemm(ap, Wp, out, prec=4, workspace=ws, splitk=splitk, gn_stats=st, gn_hw=HW, gn_apply=(gc, bc, 1e-5, flags, y), **kw)
            runs.append((planes_to_float(y).cpu(), st.cpu(), out.cpu()))
        assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
        tol = (2e-3 if flags & 2 else 0.0) + PL + 4e-6        # (fp16 rounding of the normalised value: a value next to a rounding boundary may flip)
        assert rel_err(runs[0][0], want) < tol, (kind, B, HW, K, N)
        if not flags & 4:                                      # the fp32 output is part of the contract unless marked unused
            assert rel_err(runs[0][2], ref) < TOL[4], (kind, B, HW, K, N)
            # ... and its statistics slot serves any other GroupNorm over the same tensor
            y2 = hip.planes_like(M, N, "cuda")
            hip.groupnorm_from_stats(out, y2, gc, bc, st, B, HW, N, 1e-5, flags & 3)
            assert rel_err(planes_to_float(y2).cpu(), want) < tol, (kind, B, HW, K, N)


@pytest.mark.parametrize("splitk", [0, 1, 3])
def test_gemm_concat_groupnorm_behind_the_gemm(hip, splitk):
    """mvd_gemm_desc.cat_b: the GEMM's output goes straight into torch.cat([out, skip], 1) -> GroupNorm32 -> SiLU (unet.py:550 -> the next
    ResBlock's in_layers): the split-K reduce kernel (or one launch behind an unsplit GEMM) writes the normalised planes and the raw planes
    of the concatenation; neither the GEMM output nor the concatenation needs to exist in fp32.  3x3 convolution with residual (a decoder
    ResBlock's conv2) and a dense GEMM; groups straddling the boundary (640 | 320: 30 channels per group)."""
    ws = torch.empty(32 * 1024 * 1024, device="cuda")
    for kind, B, HW, K, N, cb in [("conv", 2, 64, 64, 640, 320), ("conv", 2, 16, 96, 1280, 1280), ("dense", 2, 256, 320, 640, 640),
                                  ("dense", 1, 1024, 128, 320, 320)]:
        M, C = B * HW, N + cb
        gm, bt = torch.randn(C, generator=g(74)) + 1.0, torch.randn(C, generator=g(75))
        sk = torch.randn(M, cb, generator=g(76)) * 1.5 + 0.3
        r = torch.randn(M, N, generator=g(73))
        if kind == "conv":
            H = int(math.isqrt(HW))
            x = torch.randn(B, K, H, H, generator=g(70)) + 0.2
            w = torch.randn(N, K, 3, 3, generator=g(71)) / math.sqrt(9 * K)
            bias = torch.randn(N, generator=g(72))
            ref = F.conv2d(x, w, bias, padding=1).permute(0, 2, 3, 1).reshape(M, N) + r
            Wp = hip.pack_conv3x3(w.cuda(), bias.cuda())
            ap = hip.split_planes(x.permute(0, 2, 3, 1).reshape(M, K).contiguous().cuda())
            kw = dict(conv=dict(B=B, Hin=H, Win=H, Cin=K, Hout=H, Wout=H, stride=1, upsample=0), res=r.cuda())
        else:
            a = torch.randn(M, K, generator=g(70)) + 0.1
            w = torch.randn(N, K, generator=g(71)) / math.sqrt(K)
            bias = torch.randn(N, generator=g(72))
            ref = a @ w.t() + bias + r
            Wp = hip.pack_linear(w.cuda(), bias.cuda())
            ap = hip.split_planes(a.cuda())
            kw = dict(res=r.cuda())
        cat = torch.cat([ref, sk], 1)
        want = F.silu(F.group_norm(cat.view(B, HW, C).permute(0, 2, 1), 32, gm, bt, eps=1e-5).permute(0, 2, 1).reshape(M, C))
        gc, bc, skc = gm.cuda(), bt.cuda(), sk.cuda()
        runs = []
        for rep in range(2):
            out = torch.full((M, N), float("nan"), device="cuda")
            y, raw = hip.planes_like(M, C, "cuda"), hip.planes_like(M, C, "cuda")
            st = torch.zeros(B, 32, 2, dtype=torch.int64, device="cuda")
            hip.gemm(ap, Wp, out, prec=4, workspace=ws, splitk=splitk, gn_stats=st, gn_hw=HW,
                     gn_apply=(gc, bc, 1e-5, hip.GNA_SILU | hip.GNA_OUT_UNUSED, y), cat=(skc, raw), **kw)
            runs.append((planes_to_float(y).cpu(), planes_to_float(raw).cpu(), st.cpu()))
        assert all(torch.equal(runs[0][i], runs[1][i]) for i in range(3))
        assert rel_err(runs[0][0], want) < PL + 4e-6, (kind, B, HW, K, N, cb)
        assert rel_err(runs[0][1], cat) < TOL[4] + PL, (kind, B, HW, K, N, cb)
        # the statistics slot is the concatenation's
        cg = C // 32
        xg = cat.double().view(B, HW, 32, cg)
        sums = torch.stack([xg.sum(dim=(1, 3)), (xg * xg).sum(dim=(1, 3))], -1)
        assert rel_err(runs[0][2].double() / 2.0 ** 24, sums) < 1e-5


@pytest.mark.parametrize("silu", [1, 0])
def test_concat_groupnorm_one_launch(hip, silu):
    """mvd_concat_groupnorm: torch.cat([h, skip], 1) -> GroupNorm32 -> SiLU (unet.py:550, openaimodel.py:201-204) in one launch: normalised
    planes, raw planes (the 1x1 skip convolution's operand), optional fp32 concatenation and statistics slot; groups of 80 / 60 / 30 / 20
    channels that straddle the boundary between the two sources (640 | 320 with 30-channel groups), the largest slab of a step
    (1024 rows x 30 channels = 120 KB of LDS), and a shape the kernel does not serve (odd group width)."""
    L = hip.lib()
    assert not L.mvd_concat_groupnorm_fits(64, 32, 16, 32) and not L.mvd_concat_groupnorm_fits(320, 320, 4096, 32)
    for B, HW, ca, cb in [(2, 64, 1280, 1280), (2, 256, 1280, 640), (4, 1024, 640, 320), (3, 1024, 320, 320), (1, 16, 1280, 1280), (2, 256, 640, 640)]:
        assert L.mvd_concat_groupnorm_fits(ca, cb, HW, 32)
        M, C = B * HW, ca + cb
        a, b = torch.randn(M, ca, generator=g(56)) * 2, torch.randn(M, cb, generator=g(57)) + 0.5
        gm, bt = torch.randn(C, generator=g(58)), torch.randn(C, generator=g(59))
        ad, bd, gc, bc = a.cuda(), b.cuda(), gm.cuda(), bt.cuda()
        cat = torch.cat([a, b], 1)
        ref = F.group_norm(cat.view(B, HW, C).permute(0, 2, 1), 32, gm, bt, eps=1e-5).permute(0, 2, 1).reshape(M, C)
        if silu:
            ref = F.silu(ref)
        outs = []
        for rep, with_out in enumerate((True, False, True)):
            out = torch.full((M, C), float("nan"), device="cuda")
            raw, y = hip.planes_like(M, C, "cuda"), hip.planes_like(M, C, "cuda")
            st = torch.zeros(B, 32, 2, dtype=torch.int64, device="cuda")
            hip.check(L.mvd_concat_groupnorm(hip.ptr(ad), ca, hip.ptr(bd), cb, hip.ptr(out) if with_out else None, hip.ptr(raw), hip.ptr(y),
                                             hip.ptr(gc), hip.ptr(bc), hip.ptr(st), B, HW, 32, 1e-5, silu, hip.stream()))
            outs.append((planes_to_float(y).cpu(), planes_to_float(raw).cpu(), st.cpu()))
            if with_out:
                assert torch.equal(out.cpu(), cat)
            else:
                assert bool(torch.isnan(out).all())
        assert all(torch.equal(outs[0][i], outs[2][i]) and torch.equal(outs[0][i], outs[1][i]) for i in range(3))
        assert rel_err(outs[0][0], ref) < PL + 4e-6, (B, HW, ca, cb)
        assert rel_err(outs[0][1], cat) < PL
        # the statistics slot serves mvd_groupnorm_from_stats on the fp32 concatenation
        y2 = hip.planes_like(M, C, "cuda")
        hip.groupnorm_from_stats(cat.cuda(), y2, gc, bc, outs[0][2].cuda(), B, HW, C, 1e-5, silu)
        assert rel_err(planes_to_float(y2).cpu(), ref) < PL + 4e-6


def test_concat_groupnorm_statistics(hip):
    for B, HW, ca, cb in [(2, 64, 1280, 1280), (4, 1024, 320, 320), (3, 16, 64, 32)]:
        M = B * HW
        a, b = torch.randn(M, ca, generator=g(56)) * 2, torch.randn(M, cb, generator=g(57)) + 0.5
        C = ca + cb
        gm, bt = torch.randn(C, generator=g(58)), torch.randn(C, generator=g(59))
        ad, bd, gc, bc = a.cuda(), b.cuda(), gm.cuda(), bt.cuda()
        out = torch.empty(M, C, device="cuda")
        outp = hip.planes_like(M, C, "cuda")
        st = torch.zeros(B, 32, 2, dtype=torch.int64, device="cuda")
        hip.check(hip.lib().mvd_concat_channels(hip.ptr(ad), ca, hip.ptr(bd), cb, hip.ptr(out), hip.ptr(outp), M, hip.ptr(st), HW, 32,
                                                hip.stream()))
        cat = torch.cat([a, b], 1)
        assert torch.equal(out.cpu(), cat)
        assert rel_err(planes_to_float(outp), cat) < PL
        y = hip.planes_like(M, C, "cuda")
        hip.groupnorm_from_stats(out, y, gc, bc, st, B, HW, C, 1e-5, False)
        ref = F.group_norm(cat.view(B, HW, C).permute(0, 2, 1), 32, gm, bt, eps=1e-5).permute(0, 2, 1)
        assert rel_err(planes_to_float(y).view(B, HW, C), ref) < PL + 3e-6


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("B,HW,C,silu,eps", [(4, 1024, 320, True, 1e-5), (2, 256, 1920, True, 1e-5), (3, 64, 1280, False, 1e-6),
                                             (2, 16, 2560, True, 1e-5), (2, 1024, 32, False, 1e-6)])
def test_groupnorm(hip, B, HW, C, silu, eps):
    x = torch.randn(B, HW, C, generator=g(30)) * 2 + 0.5
    gm, bt = torch.randn(C, generator=g(31)), torch.randn(C, generator=g(32))
    ref = F.group_norm(x.permute(0, 2, 1), 32, gm, bt, eps=eps).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    y = hip.planes_like(B * HW, C, "cuda")
    ws = torch.empty(64 * 64 * 32 * 2, dtype=torch.float64, device="cuda")
    xc, gc, bc = x.cuda(), gm.cuda(), bt.cuda()
    hip.groupnorm(xc, y, gc, bc, B, HW, C, eps, silu, ws)
    assert rel_err(planes_to_float(y).view(B, HW, C), ref) < PL + 3e-6


def test_groupnorm_fp16_kept_output(hip):
    """silu flag bit 1: the GroupNorm result is rounded to fp16 before the swish (VAE decoder tail, model.py:564-570)."""
    B, HW, C = 2, 4096, 128
    x = torch.randn(B, HW, C, generator=g(33)) * 3 + 0.2
    gm, bt = torch.randn(C, generator=g(34)), torch.randn(C, generator=g(35))
    y = F.group_norm(x.permute(0, 2, 1), 32, gm, bt, eps=1e-6).permute(0, 2, 1)
    ref = y.half().float()
    ref = ref * torch.sigmoid(ref)
    yp = hip.planes_like(B * HW, C, "cuda")
    ws = torch.empty(B * hip.lib().mvd_groupnorm_chunks(HW) * 32 * 2, dtype=torch.float64, device="cuda")
    xd, gd_, bd = x.cuda(), gm.cuda(), bt.cuda()           # keep the device tensors alive across the async launch
    hip.groupnorm(xd, yp, gd_, bd, B, HW, C, 1e-6, 3, ws)
    got = planes_to_float(yp).view(B, HW, C)
    # a value within an fp32 ulp of an fp16 rounding boundary may round the other way: allow 2^-11 relative on <0.1 % of them
    d = (got - ref).abs()
    assert float((d > (2e-6 + 2 * PL) * (1 + ref.abs())).float().mean()) < 1e-3        # PL: split-plane representation error
    assert float((d / (1e-3 + ref.abs())).max()) < 2.5e-3      # one flipped fp16 ulp (2^-10 at the bottom of a binade) through the swish


@pytest.mark.parametrize("rows,cols,scale", [(1024, 1024, 512 ** -0.5), (64, 64, 1.0), (100, 4096, 0.05), (7, 32, 3.0)])
def test_softmax_rows(hip, rows, cols, scale):
    x = torch.randn(rows, cols, generator=g(36)) * 8
    ref = torch.softmax(x * scale, dim=1)
    yp = hip.planes_like(rows, cols, "cuda")
    xd = x.cuda()
    hip.softmax_rows(xd, yp, scale, out_scale=1024.0)
    assert rel_err(planes_to_float(yp) / 1024.0, ref) < 2e-6 + PL


@pytest.mark.parametrize("rows,C", [(1000, 320), (64, 1280), (4096, 256), (37, 640), (16, 32)])
def test_layernorm(hip, rows, C):
    x = torch.randn(rows, C, generator=g(33)) * 3 + 1
    w, b = torch.randn(C, generator=g(34)), torch.randn(C, generator=g(35))
    y = hip.planes_like(rows, C, "cuda")
    xc, wc, bc = x.cuda(), w.cuda(), b.cuda()
    hip.layernorm(xc, y, wc, bc, rows, C, eps=1e-5)
    assert rel_err(planes_to_float(y), F.layer_norm(x, (C,), w, b, eps=1e-5)) < PL + 3e-6
    hip.layernorm(xc, y, wc, bc, rows, C, eps=1e-6, w_plus_one=True)   # adaLN modulate
    assert rel_err(planes_to_float(y), F.layer_norm(x, (C,), eps=1e-6) * (1 + w) + b) < PL + 3e-6


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("prec", [4, 3, 1])
@pytest.mark.parametrize("B,H,L,d", [(2, 8, 1024, 40), (2, 8, 256, 80), (3, 8, 64, 160), (2, 8, 16, 160), (2, 8, 1024, 4),
                                     (1, 8, 256, 8), (1, 8, 64, 16), (1, 8, 256, 32)])
def test_qkv_gemm_and_attention(hip, prec, B, H, L, d):
    """QKV GEMM with the routing epilogue + flash attention == CrossAttention(context=None) core (attention.py:170-193)."""
    C = H * d
    x = torch.randn(B * L, C, generator=g(40))
    wq, wk, wv = (torch.randn(C, C, generator=g(41 + i)) / math.sqrt(C) * 1.5 for i in range(3))
    q, k, v = (F.linear(x, w).view(B, L, H, d).permute(0, 2, 1, 3) for w in (wq, wk, wv))
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * d ** -0.5
    ref = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), v).permute(0, 2, 1, 3).reshape(B * L, C)
    Wp = hip.pack_linear_cat([wq.cuda(), wk.cuda(), wv.cuda()])
    planes = hip.alloc_attn_planes(B, H, L, d, "cuda")
    ws = torch.empty(16 * 1024 * 1024, device="cuda")
    xp = hip.split_planes(x.cuda())
    hip.gemm(xp, Wp, None, prec=prec, epi=hip.EPI_QKV, qkv=dict(planes=planes, heads=H, dhead=d, L=L), workspace=ws)
    out = hip.planes_like(B * L, C, "cuda")
    hip.attention(planes, out, B, H, L, d, prec=prec)
    assert rel_err(planes_to_float(out), ref) < 2 * TOL[prec] + PL
    if prec == 4 and L <= 256:      # every kernel configuration that serves the QKV routing epilogue writes the same operand planes
        want = None                 # (the call above may have split K: compare the unsplit runs with each other)
        for cfg in hip.gemm_configs(hip.EPI_QKV)[::2]:
            for t in planes:
                t.zero_()
            hip.gemm(xp, Wp, None, prec=prec, epi=hip.EPI_QKV, qkv=dict(planes=planes, heads=H, dhead=d, L=L), workspace=ws, cfg=cfg, splitk=1)
            if want is None:
                want = [t.clone() for t in planes]
            assert all(torch.equal(a, b) for a, b in zip(planes, want)), cfg
        hip.attention(planes, out, B, H, L, d, prec=prec)
        assert rel_err(planes_to_float(out), ref) < 2 * TOL[prec] + PL


@pytest.mark.parametrize("prec", [4, 3, 1])
@pytest.mark.parametrize("B,H,L,Lk,d", [(8, 8, 1024, 0, 40), (8, 8, 1024, 1000, 40), (16, 8, 640, 0, 32), (8, 8, 1024, 999, 4), (8, 8, 1024, 0, 80)])
def test_attention_phased_equals_single_tile(hip, prec, B, H, L, Lk, d):
    """Long sequences run two query tiles per wavefront (K / V^T fragments read once for both; with -DMVD_ATTN_PHASED in phases
    S(q0) | S(q1) || softmax(q0) | PV(q0) || softmax(q1) | PV(q1)) with the per-accumulator MFMA order of the one-tile kernel: the
    outputs must be BIT-IDENTICAL to it (MVD_ATTN_QT1=1 forces the one-tile kernel), including the ragged last key tile (Lk keys < L
    rows) and the head dims without a full 32-channel step (d = 4).  This is the only op-level test that reaches the two-tile kernel
    (it needs >= 512 workgroups: B * heads * L / 128)."""
    import os
    C = H * d
    x = torch.randn(B * L, C, generator=g(140)) * 1.3
    w = torch.randn(3 * C, C, generator=g(141)) / math.sqrt(C) * 1.5
    Wp = hip.pack_linear(w.cuda())
    planes = hip.alloc_attn_planes(B, H, L, d, "cuda")
    ws = torch.empty(16 * 1024 * 1024, device="cuda")
    hip.gemm(hip.split_planes(x.cuda()), Wp, None, prec=4, epi=hip.EPI_QKV, qkv=dict(planes=planes, heads=H, dhead=d, L=L), workspace=ws)
    outs = []
    for force in (False, True):
        if force:
            os.environ["MVD_ATTN_QT1"] = "1"
        try:
            out = hip.planes_like(B * L, C, "cuda")
            out.zero_()
            hip.attention(planes, out, B, H, L, d, prec=prec, Lkeys=Lk)
            torch.cuda.synchronize()
            outs.append(out.cpu())
        finally:
            os.environ.pop("MVD_ATTN_QT1", None)
    if prec >= 3:
        assert torch.equal(outs[0], outs[1])
    else:      # the one-product debugging mode: the two kernels agree to its own 11-bit (fp16) / 8-bit (bf16) operand tolerance only
        assert rel_err(planes_to_float(outs[0]), planes_to_float(outs[1])) < TOL[1]
    q, k, v = (F.linear(x, w[i * C:(i + 1) * C]).view(B, L, H, d).permute(0, 2, 1, 3) for i in range(3))
    kk = Lk or L
    sim = torch.einsum("bhid,bhjd->bhij", q, k[:, :, :kk]) * d ** -0.5
    ref = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), v[:, :, :kk]).permute(0, 2, 1, 3).reshape(B * L, C)
    assert rel_err(planes_to_float(outs[0]), ref) < 2 * TOL[prec] + PL


@pytest.mark.parametrize("pcfg,psplit", [(0, 1), (_hip.make_cfg(2, 1), 1), (_hip.make_cfg(1, 0), 1), (_hip.make_cfg(3, 4), 1), (0, 3)])
@pytest.mark.parametrize("B,L,H,d,offs", [(2, 256, 8, 40, 0.0), (1, 64, 8, 160, 3.0), (3, 32, 4, 24, -1.5)])
def test_gemm_layernorm_fold(hip, pcfg, psplit, B, L, H, d, offs):
    """LayerNorm folded into the consumer GEMM (mvd_gemm_desc.rs_out / ln_stats): the producer GEMM (tile epilogues with 32- and
    80-column wave tiles, and the split-K reduce) emits per-row {sum, sum of squares} slots and the row's split planes; the QKV and
    GEGLU consumers run on the RAW rows with W' = W diag(gamma) and finish  rstd (acc - mean colsum) + W beta  in their epilogues.
    Rows with a large mean (offs) exercise the cancellation in  x.W' - mean sum(W')."""
    import torch.nn as nn
    M, C = B * L, H * d
    x = torch.randn(M, C, generator=g(80))
    Win = torch.randn(C, C, generator=g(81)) / math.sqrt(C)
    bin_ = torch.randn(C, generator=g(82)) + offs
    res = torch.randn(M, C, generator=g(83))
    t = F.linear(x, Win, bin_) + res
    norm = nn.LayerNorm(C)
    with torch.no_grad():
        norm.weight.copy_(1.0 + 0.3 * torch.randn(C, generator=g(84)))
        norm.bias.copy_(0.2 * torch.randn(C, generator=g(85)))
    ln = F.layer_norm(t, (C,), norm.weight, norm.bias, norm.eps)
    ws = torch.empty(16 * 1024 * 1024, device="cuda")
    normc = norm.cuda()
    # producer: t (fp32), its planes and the row statistics
    tp = hip.planes_like(M, C, "cuda")
    tt = torch.empty(M, C, device="cuda")
    rs = hip.RowStats(M, C, "cuda")
    hip.gemm(hip.split_planes(x.cuda()), hip.pack_linear(Win.cuda(), bin_.cuda()), tt, res=res.cuda(), out_planes=tp, row_stats=rs,
             workspace=ws, cfg=pcfg or None, splitk=psplit)
    assert rel_err(tt, t) < TOL[4]
    cnt = int(rs.count.item())
    got = rs.slots[:, :cnt].sum(1).cpu().double()
    assert 1 <= cnt <= rs.ld
    assert float((got[:, 0] - t.double().sum(1)).abs().max()) < 1e-3 and rel_err(got[:, 1], (t.double() ** 2).sum(1)) < 1e-5
    # consumer 1: QKV -> attention
    wq, wk, wv = (torch.randn(C, C, generator=g(86 + i)) / math.sqrt(C) * 1.5 for i in range(3))
    q, k, v = (F.linear(ln, w).view(B, L, H, d).permute(0, 2, 1, 3) for w in (wq, wk, wv))
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * d ** -0.5
    ref = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), v).permute(0, 2, 1, 3).reshape(M, C)
    fold = hip.LnFold(torch.cat([wq, wk, wv], 0).cuda(), None, normc)
    planes = hip.alloc_attn_planes(B, H, L, d, "cuda")
    out = hip.planes_like(M, C, "cuda")
    for ccfg in (None, hip.make_cfg(1, hip.WS_LOOP)):      # built-in choice; the wave-specialised kernel (two 32-column blocks per consumer tile)
        for t_ in planes:
            t_.zero_()
        hip.gemm(tp, fold.w, None, epi=hip.EPI_QKV, qkv=dict(planes=planes, heads=H, dhead=d, L=L), workspace=ws, ln=(rs, fold), cfg=ccfg,
                 splitk=1 if ccfg else 0)
        hip.attention(planes, out, B, H, L, d)
        assert rel_err(planes_to_float(out), ref) < 4 * TOL[4] + PL, ccfg
    # consumer 2: GEGLU, reading the rows as a column block of a wider operand buffer and writing its first columns
    Wg = torch.randn(8 * C, C, generator=g(90)) / math.sqrt(C)
    bg = torch.randn(8 * C, generator=g(91))
    a, gt = F.linear(ln, Wg, bg).chunk(2, dim=-1)
    refg = a * F.gelu(gt)
    foldg = hip.LnFold(Wg.cuda(), bg.cuda(), normc, geglu=True)
    cat5 = hip.planes_like(M, 5 * C, "cuda")
    cat5[:, 2 * 4 * C:] = tp
    for ccfg in (None, hip.make_cfg(1, hip.WS_LOOP)):
        cat5[:, :2 * 4 * C].zero_()
        hip.gemm(cat5[:, 2 * 4 * C:], foldg.w, None, M=M, lda=5 * C, epi=hip.EPI_GEGLU, out_planes=cat5, workspace=ws, ln=(rs, foldg), cfg=ccfg,
                 splitk=1 if ccfg else 0)
        gotg = planes_to_float(cat5)
        assert rel_err(gotg[:, :4 * C], refg) < 4 * TOL[4] + PL, ccfg
        assert torch.equal(cat5[:, 2 * 4 * C:].cpu(), tp.cpu())


def test_attention_forced_rescale(hip):
    """A key whose score dominates late in the sequence forces the online-softmax rescale path."""
    B, H, L, d = 1, 8, 256, 40
    C = H * d
    x = torch.randn(B * L, C, generator=g(50))
    x[200] *= 12.0
    wq = torch.eye(C)
    q, k, v = (x.view(B, L, H, d).permute(0, 2, 1, 3) for _ in range(3))
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * d ** -0.5
    ref = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), v).permute(0, 2, 1, 3).reshape(B * L, C)
    Wp = hip.pack_linear_cat([wq.cuda(), wq.cuda(), wq.cuda()])
    planes = hip.alloc_attn_planes(B, H, L, d, "cuda")
    xp = hip.split_planes(x.cuda())
    hip.gemm(xp, Wp, None, epi=hip.EPI_QKV, qkv=dict(planes=planes, heads=H, dhead=d, L=L))
    out = hip.planes_like(B * L, C, "cuda")
    hip.attention(planes, out, B, H, L, d)
    assert rel_err(planes_to_float(out), ref) < 2 * TOL[4] + PL


@pytest.mark.parametrize("D", [1, 3])
def test_pixel_cross_attn(hip, D):
    P, H, d = 300, 8, 40
    C = H * d
    q = torch.randn(P, C, generator=g(60))
    k = torch.randn(P * D, C, generator=g(61))
    v = torch.randn(P * D, C, generator=g(62))
    qq = q.view(P, 1, H, d).permute(0, 2, 1, 3)
    kk, vv = (t.view(P, D, H, d).permute(0, 2, 1, 3) for t in (k, v))
    sim = torch.einsum("phid,phjd->phij", qq, kk) * d ** -0.5
    ref = torch.einsum("phij,phjd->phid", sim.softmax(-1), vv).permute(0, 2, 1, 3).reshape(P, C)
    out = hip.planes_like(P, C, "cuda")
    qc, kc, vc = q.cuda(), k.cuda(), v.cuda()      # keep the device tensors alive across the raw-pointer call
    hip.check(hip.lib().mvd_pixel_cross_attn(hip.ptr(qc), hip.ptr(kc), hip.ptr(vc), hip.ptr(out), P, D, H, d, hip.stream()))
    assert rel_err(planes_to_float(out), ref) < PL + 2e-6


@pytest.mark.parametrize("V", [4, 8, 3, 2, 5, 6, 7, 15, 16])      # 2/3/4/8: templated kernel; the rest: generic kernel
def test_view_mha_and_pool(hip, V):
    N, H, d = 500, 8, 32
    C = H * d
    qkv = torch.randn(N * V, 3 * C, generator=g(70))
    t = qkv.view(N, V, 3, H, d).permute(2, 0, 3, 1, 4)
    q, k, v = t.unbind(0)
    ref = (((q * d ** -0.5) @ k.transpose(-2, -1)).softmax(-1) @ v).transpose(1, 2).reshape(N * V, C)
    out = hip.planes_like(N * V, C, "cuda")
    qkvc = qkv.cuda()
    hip.check(hip.lib().mvd_view_mha(hip.ptr(qkvc), hip.ptr(out), N, V, H, d, hip.stream()))
    assert rel_err(planes_to_float(out), ref) < PL + 2e-6
    x = torch.randn(N, V, C, generator=g(71))
    w, b = torch.randn(1, C, generator=g(72)) * 0.2, torch.randn(1, generator=g(73))
    wt = F.linear(x, w, b).softmax(dim=-2)
    refp = (x * wt).sum(-2)
    outp = hip.planes_like(N, C, "cuda")
    xc, wc, bc = x.cuda(), w.cuda(), b.cuda()
    hip.check(hip.lib().mvd_view_pool(hip.ptr(xc), hip.ptr(wc), hip.ptr(bc), hip.ptr(outp), N, V, C, hip.stream()))
    assert rel_err(planes_to_float(outp), refp) < PL + 2e-6


# ------------------------------------------------------------------------------------------------ small kernels
@pytest.mark.parametrize("M,N,K", [(1, 1280, 320), (8, 768, 796), (16, 1280, 1280), (3, 50, 7)])  # fp32 GEMV
def test_gemv(hip, M, N, K):
    W, b, x = torch.randn(N, K, generator=g(80)), torch.randn(N, generator=g(81)), torch.randn(M, K, generator=g(82))
    y = torch.empty(M, N, device="cuda")
    hip.gemv(W.cuda(), b.cuda(), x.cuda(), y, act_in=hip.ACT_SILU, act_out=hip.ACT_SILU)
    assert rel_err(y, F.silu(F.linear(F.silu(x), W, b))) < 3e-6


def test_area_pool_concat_input(hip):
    B, S, D, C = 2, 32, 3, 768
    vol = torch.randn(B, S, S, D, C, generator=g(90))
    v = vol.permute(0, 3, 4, 1, 2).reshape(B * D, C, S, S)
    for f in (2, 4, 8):
        ref = F.interpolate(v, scale_factor=1.0 / f, mode="area").reshape(B, D, C, S // f, S // f).permute(0, 3, 4, 1, 2)
        out = hip.planes_like(B * (S // f) * (S // f) * D, C, "cuda")
        volc = vol.cuda()
        hip.check(hip.lib().mvd_area_pool(hip.ptr(volc), hip.ptr(out), B, S, D, C, f, 0, hip.stream()))
        assert rel_err(planes_to_float(out).view(B, S // f, S // f, D, C), ref) < PL
        # into columns [64, 64 + C) of a wider operand buffer (ldp = C + 96); the other columns are left alone
        rows, ld = B * (S // f) * (S // f) * D, C + 96
        wide = hip.split_planes(torch.full((rows, ld), 7.0, device="cuda"))
        hip.check(hip.lib().mvd_area_pool(hip.ptr(volc), hip.c_void_p(wide.data_ptr() + 4 * 64), B, S, D, C, f, ld, hip.stream()))
        got = planes_to_float(wide)
        assert rel_err(got[:, 64:64 + C].reshape(B, S // f, S // f, D, C), ref) < PL
        assert bool((got[:, :64] == 7.0).all()) and bool((got[:, 64 + C:] == 7.0).all())
    a, b = torch.randn(100, 320, generator=g(91)), torch.randn(100, 640, generator=g(92))
    out = torch.empty(100, 960, device="cuda")
    ac, bc = a.cuda(), b.cuda()
    outp = hip.planes_like(100, 960, "cuda")
    hip.check(hip.lib().mvd_concat_channels(hip.ptr(ac), 320, hip.ptr(bc), 640, hip.ptr(out), hip.ptr(outp), 100, None, 0, 0, hip.stream()))
    assert torch.equal(out.cpu(), torch.cat([a, b], 1))
    assert rel_err(planes_to_float(outp), torch.cat([a, b], 1)) < PL
    V, S = 3, 32
    x, il = torch.randn(V, 5, S, S, generator=g(93)), torch.randn(1, 5, S, S, generator=g(94))
    xip = hip.planes_like(2 * V * S * S, 32, "cuda")
    xc, ilc = x.cuda(), il.cuda()
    hip.check(hip.lib().mvd_unet_input(hip.ptr(xc), hip.ptr(ilc), hip.ptr(xip), V, S, 32, 1, hip.stream()))
    xi = planes_to_float(xip).view(2 * V, S, S, 32)
    xc = il.expand(V, -1, -1, -1).clone()
    xc[:, :4] = xc[:, :4] / 0.18215
    ref = torch.zeros(2 * V, 32, S, S)
    ref[:V, :10] = torch.cat([x, xc], 1)
    ref[V:, :5] = x
    assert rel_err(xi.permute(0, 3, 1, 2), ref) < PL


def test_cfg_ddim_update_golden(hip):
    """CFG combine + DDIM update against the reference's denoise_apply_impl outputs (tests/golden/schedule.npz)."""
    from mvdfusion_amd.engine import ddim_step_table
    from mvdfusion_amd.scheduler import make_tables
    gd = load_golden("schedule")
    tab = make_tables()
    dd = {"timesteps": gd["ddim_timesteps"], "alphas": gd["ddim_alphas"], "alphas_prev": gd["ddim_alphas_prev"],
          "sigmas": gd["ddim_sigmas"], "sqrt_one_minus_alphas": gd["ddim_sqrt_one_minus_alphas"]}
    V, S = 2, 8
    for index in (49, 25, 1, 0):
        steps = ddim_step_table(tab, dd, [index]).cuda()
        it = torch.zeros(1, dtype=torch.int32, device="cuda")
        x = gd["x"].clone().cuda()
        x0 = torch.empty_like(x)
        eps = gd["eps"]
        # fake "UNet head" layout: (2V, S, S, 8) with cond == uncond == eps  => guided eps == eps for any scale
        e = torch.zeros(2 * V, S, S, 8)
        e[:V, ..., :5] = eps.permute(0, 2, 3, 1)
        e[V:, ..., :5] = eps.permute(0, 2, 3, 1)
        noise = gd["noise"].reshape(1, V, 5, S, S).cuda()
        ec = e.cuda()
        hip.check(hip.lib().mvd_cfg_ddim_update(hip.ptr(ec), 8, hip.ptr(x), hip.ptr(x0), None, hip.ptr(noise),
                                                V * 5 * S * S, hip.ptr(steps), hip.ptr(it), V, S, 1, 2.5, 1, hip.stream()))
        assert rel_err(x, gd[f"x_prev_{index}"]) < 2e-6
        assert rel_err(x0, gd[f"x0_{index}"]) < 2e-6


def test_weight_prefetch_requests_change_nothing():
    """mvd_gemm_desc.pf_items (the role-split kernel's consumer wavefronts request the weights of later launches) only READS: outputs
    bit-identical with and without, the weights untouched."""
    from mvdfusion_amd import hip
    M, N, K = 1024, 320, 640
    A = hip.split_planes(torch.randn(M, K, generator=g(1)).cuda())
    Ws = [hip.pack_linear((torch.randn(N, K, generator=g(10 + i)) / math.sqrt(K)).cuda(), torch.zeros(N).cuda()) for i in range(4)]
    keep = [w.data.clone() for w in Ws]
    ws = torch.empty(8 * 1024 * 1024, device="cuda")
    cfgs = [hip.make_cfg(2, hip.WS_LOOP), 0, hip.make_cfg(0, 0), hip.make_cfg(2, hip.WS_LOOP)]
    outs = [torch.empty(M, N, device="cuda") for _ in range(4)]

    def run():
        for w, c, o in zip(Ws, cfgs, outs):
            hip.gemm(A, w, o, prec=3, workspace=ws, splitk=1, cfg=c)
        torch.cuda.synchronize()
        return [o.clone() for o in outs]

    ref = run()
    pf = hip.WeightPrefetcher(A.device)
    with pf.following(record=True):
        run()
    assert len(pf.seq) == 4 and [e[2] for e in pf.seq] == [True, False, False, True]
    # launch 0 hosts the weights of launches 1 .. 3, launch 3 (the last) nothing
    assert pf.shares == {0: (0, 3)} and [it[3] for it in pf.items] == [1, 2, 3]
    with pf.following():
        got = run()
    assert all(torch.equal(a, b) for a, b in zip(got, ref))
    assert all(torch.equal(w.data, k) for w, k in zip(Ws, keep))
