"""ctypes binding of the C ABI in include/mvd_hip.h (mvdfusion_amd/csrc/libmvd_hip.so).

There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised (the product path must
never silently run on something other than the HIP kernels).  Everything is enqueued on torch's current stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# two flavours of the same sources: fp16 MFMA operands (default) and bf16 (-DMVD_OPERAND_BF16)
LIB_PATHS = {"f16": os.path.join(_HERE, "csrc", "libmvd_hip.so"), "bf16": os.path.join(_HERE, "csrc", "libmvd_hip_bf16.so")}
if os.environ.get("MVD_HIP_LIB"):        # A/B measurements: another build of the same ABI (tools/probes/ab_build.sh)
    LIB_PATHS["f16"] = os.environ["MVD_HIP_LIB"]
LIB_PATH = LIB_PATHS["f16"]
OPERAND_FORMAT = os.environ.get("MVD_OPERAND_FORMAT", "f16")   # chosen before the first call, fixed per process

PREC_X1, PREC_X3, PREC_X4 = 1, 3, 4
# Layer classes a precision policy may address (Ctx.gemm(kind=...), Ctx.prec_of): the number of partial products of the
# (hi + lo)(hi + lo) operand split is chosen per class, not per model.
PREC_KINDS = ("conv", "skip", "proj", "qkv", "attn", "out", "geglu", "ffproj", "xattn", "ga")


def parse_precision(name):
    """'f16x4' / 'f16x3' / 'bf16x3' / 'f16' / 'bf16' -> (operand format, default products, {}); a policy string
    'f16x4:conv=3,geglu=3' (default x4, the named layer classes of PREC_KINDS at x3) -> (format, 4, {'conv': 3, 'geglu': 3})."""
    base, _, over = name.partition(":")
    names = {"f16": ("f16", PREC_X1), "bf16": ("bf16", PREC_X1), "f16x3": ("f16", PREC_X3), "f16x4": ("f16", PREC_X4),
             "bf16x3": ("bf16", PREC_X3), "bf16x4": ("bf16", PREC_X4)}
    if base not in names:          # (a typo such as 'f16x33' used to select the one-product mode silently)
        raise ValueError(f"precision '{name}': base must be one of {sorted(names)}")
    fmt, default = names[base]
    policy = {}
    for item in filter(None, over.split(",")):
        k, _, v = item.partition("=")
        k = k.strip()
        if k not in PREC_KINDS or int(v) not in (PREC_X1, PREC_X3, PREC_X4):
            raise ValueError(f"precision policy '{name}': '{item}' is not <kind>=<1|3|4> with kind in {PREC_KINDS}")
        policy[k] = int(v)
    return fmt, default, policy
A_DENSE, A_CONV3X3 = 0, 1
EPI_STORE, EPI_GEGLU, EPI_QKV = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_SILU, ACT_QUICKGELU = 0, 1, 2, 3
STEP_STRIDE = 8
CAM_RECORD = 20
TOKEN_DIM, TOKEN_LD = 723, 736

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
c_void_p = C.c_void_p


class GemmDesc(C.Structure):
    """struct mvd_gemm_desc (field order must match include/mvd_hip.h)."""
    _fields_ = [
        ("M", _i), ("N", _i), ("K", _i),
        ("A", _vp), ("lda", _i), ("a_mode", _i),
        ("B", _i), ("Hin", _i), ("Win", _i), ("Cin", _i), ("Hout", _i), ("Wout", _i), ("stride", _i), ("upsample", _i), ("no_pad_tl", _i),
        ("Wp", _vp), ("b_mode", _i), ("ldb", _i), ("acc_scale", _f), ("prec", _i),
        ("epi", _i), ("act", _i), ("out", _vp), ("ldo", _i), ("out_sp", _vp), ("ldp", _i),
        ("n_store", _i),
        ("bias", _vp), ("bias_b", _vp), ("rows_per_batch", _i), ("ldbb", _i), ("colscale", _vp), ("res", _vp), ("ldr", _i),
        ("q_hi", _vp), ("q_lo", _vp), ("k_hi", _vp), ("k_lo", _vp), ("vt_hi", _vp), ("vt_lo", _vp),
        ("heads", _i), ("dhead", _i), ("L", _i), ("Lpad", _i), ("qscale", _f),
        ("splitk", _i), ("workspace", _vp), ("workspace_elems", _sz), ("cfg", _i),
        ("gn_stats", _vp), ("gn_hw", _i), ("gn_groups", _i),
        ("rs_out", _vp), ("rs_count", _vp), ("rs_ld", _i),
        ("ln_stats", _vp), ("ln_count", _vp), ("ln_colsum", _vp), ("ln_ld", _i), ("ln_dim", _i), ("ln_eps", _f),
        ("gna_out_sp", _vp), ("gna_gamma", _vp), ("gna_beta", _vp), ("gna_eps", _f), ("gna_flags", _i),
        ("cat_b", _vp), ("cat_cb", _i), ("cat_raw_sp", _vp),
        ("acc_scale_dev", _vp), ("pf_items", _vp), ("pf_n", _i),
    ]


class PrefetchItem(C.Structure):
    """ctypes mirror of mvd_prefetch_item (include/mvd_hip.h)."""
    _fields_ = [("ptr", _vp), ("bytes", C.c_ulonglong), ("start_after", _i), ("consumer", _i)]


# name -> (restype, argtypes): every symbol declared in include/mvd_hip.h
SIGNATURES = {
    "mvd_version": (_i, []),
    "mvd_last_error": (C.c_char_p, []),
    "mvd_packed_weight_bytes": (_sz, [_i, _i]),
    "mvd_operand_format": (_i, []),
    "mvd_pack_linear_weight": (_i, [_vp, _i, _i, _i, _i, _f, _vp, _vp]),
    "mvd_pack_linear_weight_t": (_i, [_vp, _i, _i, _i, _f, _vp, _vp]),
    "mvd_pack_conv3x3_weight": (_i, [_vp, _i, _i, _i, _f, _vp, _vp]),
    "mvd_gemm": (_i, [C.POINTER(GemmDesc), _vp]),
    "mvd_gemm_cfg_supported": (_i, [C.POINTER(GemmDesc), _i]),
    "mvd_split_planes": (_i, [_vp, _vp, _sz, _i, _i, _i, _vp]),
    "mvd_gemv": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "mvd_groupnorm_chunks": (_i, [_i]),
    "mvd_groupnorm_nhwc": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp]),
    "mvd_layernorm": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp]),
    "mvd_softmax_rows": (_i, [_vp, _vp, _i, _i, _i, _f, _f, _vp]),
    "mvd_attn_qk_plane_elems": (_sz, [_i, _i, _i, _i]),
    "mvd_attn_vt_plane_elems": (_sz, [_i, _i, _i, _i]),
    "mvd_attn_lpad": (_i, [_i]),
    "mvd_attention": (_i, [_vp] * 7 + [_i, _i, _i, _i, _i, _i, _i, _vp]),
    "mvd_pixel_cross_attn": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mvd_unet_input": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mvd_concat_channels": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _vp]),
    "mvd_concat_groupnorm_fits": (_i, [_i, _i, _i, _i]),
    "mvd_concat_groupnorm": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp]),
    "mvd_transpose_planes": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "mvd_im2col3x3_t_planes": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "mvd_col_sum_workspace_doubles": (_sz, [_i, _i]),
    "mvd_col_sum": (_i, [_vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "mvd_col_sum_pow2": (_i, [_vp, _i, _i, _i, _vp, _vp, _sz, _vp, _vp, _vp]),
    "mvd_gridattn_tokens_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _f, _i, _i, _i, _i, _i, _f, _f, _vp]),
    "mvd_layernorm_backward": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp]),
    "mvd_geglu_backward": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "mvd_act_planes": (_i, [_vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _vp]),
    "mvd_act_backward": (_i, [_vp, _vp, _vp, _sz, _i, _vp]),
    "mvd_attention_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mvd_pixel_cross_attn_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "mvd_groupnorm_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mvd_groupnorm_from_stats": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "mvd_area_pool": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "mvd_fill_zero": (_i, [_vp, _sz, _vp]),
    "mvd_pow2_scale": (_i, [_vp, _sz, _vp, _vp, _vp]),
    "mvd_split_planes_scaled": (_i, [_vp, _vp, _sz, _i, _i, _i, _vp, _vp]),
    "mvd_transpose_planes_scaled": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp]),
    "mvd_adamw_multi": (_i, [_vp, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _vp]),
    "mvd_timestep_embedding": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "mvd_advance_iter": (_i, [_vp, _vp]),
    "mvd_zembed": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "mvd_gridattn_tokens": (_i, [_vp] * 10 + [_i, _i, _i, _i, _i, _f, _f, _vp]),
    "mvd_gridattn_fused_slots": (_i, []),
    "mvd_gridattn_fused_stream_bytes": (_sz, []),
    "mvd_gridattn_fused_vec_floats": (_sz, []),
    "mvd_gridattn_fused": (_i, [_vp] * 12 + [_i, _i, _i, _i, _i, _f, _f, _i, _vp]),
    "mvd_view_mha": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "mvd_view_pool": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "mvd_cfg_ddim_update": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _i, _i, _i, _f, _i, _vp]),
    "mvd_graph_begin": (_i, [_vp]),
    "mvd_graph_end": (_i, [_vp, C.POINTER(_vp)]),
    "mvd_graph_launch": (_i, [_vp, _vp]),
    "mvd_graph_destroy": (_i, [_vp]),
    "mvd_event_create": (_i, [C.POINTER(_vp)]),
    "mvd_event_record": (_i, [_vp, _vp]),
    "mvd_event_elapsed_ms": (_i, [_vp, _vp, C.POINTER(_f)]),
    "mvd_event_destroy": (_i, [_vp]),
}

_lib = None


def lib():
    """Load libmvd_hip.so (once).  Raises if it has not been built -- there is no CPU fallback."""
    global _lib
    if _lib is None:
        path = LIB_PATHS[OPERAND_FORMAT]
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} not found: build it with `python -m mvdfusion_amd.csrc.build` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). mvdfusion_amd has no CPU fallback.")
        l = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        assert l.mvd_operand_format() == {"f16": 0xf16, "bf16": 0xbf16}[OPERAND_FORMAT]
        _lib = l
        load_default_tuned()
    return _lib


def set_operand_format(fmt):
    """'f16' (default) or 'bf16'; must be called before the first kernel call of the process."""
    global OPERAND_FORMAT
    assert fmt in LIB_PATHS
    if _lib is not None and fmt != OPERAND_FORMAT:
        raise RuntimeError(f"operand format already fixed to {OPERAND_FORMAT} for this process")
    OPERAND_FORMAT = fmt


def check(rc):
    if rc != 0:
        raise RuntimeError(f"mvd_hip error {rc}: {lib().mvd_last_error().decode()}")


try:
    _raw_stream = torch._C._cuda_getCurrentRawStream      # one C call (torch.cuda.current_stream() builds a Stream object through five Python
    _cur_device = torch._C._cuda_getDevice                 # frames: 17 000 calls = ~10 ms of host time per training step)
except AttributeError:
    _raw_stream = None


def stream():
    """torch's current stream of the current device as the library's mvd_stream_t."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(_cur_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def _req(t, dtype=torch.float32):
    assert t.is_cuda and t.dtype == dtype and t.is_contiguous(), (t.device, t.dtype, t.shape, t.stride())
    return t


# ---------------------------------------------------------------------------------------------
# weights
# ---------------------------------------------------------------------------------------------
class PackedWeight:
    """A weight in the MFMA operand image ([K/32][N/16][hi image | lo image], include/mvd_hip.h) plus its fp32 bias."""

    __slots__ = ("data", "N", "K", "n_real", "bias", "geglu", "conv_cin", "acc_scale")

    def __init__(self, data, N, K, n_real, bias, geglu=False, conv_cin=0, acc_scale=1.0):
        self.data, self.N, self.K, self.n_real, self.bias, self.geglu, self.conv_cin = data, N, K, n_real, bias, geglu, conv_cin
        self.acc_scale = acc_scale


class RowStats:
    """Per-row {sum, sum of squares} slots a GEMM emits for the LayerNorm folded into its consumer (mvd_gemm_desc.rs_out)."""

    __slots__ = ("slots", "count", "ld")

    def __init__(self, rows, n_max, device):
        self.ld = (n_max + 31) // 32
        self.slots = torch.zeros(rows, self.ld, 2, dtype=torch.float32, device=device)
        self.count = torch.zeros(1, dtype=torch.int32, device=device)


class LnFold:
    """A Linear behind a LayerNorm as ONE packed weight: W' = W diag(gamma) (PackedWeight .w, bias = W beta + b), the column sums of
    W' and the LayerNorm's width / eps (mvd_gemm_desc.ln_*; exact algebra, the composition is formed in fp64)."""

    __slots__ = ("w", "colsum", "dim", "eps")

    def __init__(self, weight, bias, norm, geglu=False):
        Wd = weight.detach().double()
        g, b = norm.weight.detach().double(), norm.bias.detach().double()
        Wf = Wd * g[None, :]
        bf = Wd @ b + (bias.detach().double() if bias is not None else 0.0)
        self.w = pack_linear(Wf.float().contiguous(), bf.float().contiguous(), geglu=geglu)
        Np = self.w.N if not geglu else Wd.shape[0]
        cs = torch.zeros(max(Np, Wd.shape[0]), dtype=torch.float32, device=weight.device)
        cs[:Wd.shape[0]] = Wf.sum(dim=1).float()
        self.colsum, self.dim, self.eps = cs, int(Wd.shape[1]), float(norm.eps)


class PlanesOperand:
    """An activation matrix (N rows, K columns, split planes with `ld` elements per row) in the B-operand role of mvd_gemm:
    out = A @ B^T between two activations (MVD_B_PLANES).  Duck-types PackedWeight for hip.gemm."""

    __slots__ = ("data", "N", "K", "n_real", "bias", "geglu", "conv_cin", "acc_scale", "ld")

    def __init__(self, planes, N=None, K=None, bias=None, acc_scale=1.0, ld=None):
        assert planes.dtype == torch.int16 and planes.dim() == 2
        self.data, self.ld = planes, int(ld if ld is not None else planes.shape[-1] // 2)   # ld: row stride when `planes` is a column view
        self.N = int(N if N is not None else planes.shape[0])
        self.K = int(K if K is not None else self.ld)
        assert self.N % 16 == 0 and self.K % 32 == 0 and self.K <= self.ld, (self.N, self.K, self.ld)
        self.n_real, self.bias, self.geglu, self.conv_cin, self.acc_scale = self.N, bias, False, 0, acc_scale


# max|w| of the model's parameters, measured by ONE batched reduction and ONE host read (register_param_maxima): storage pointer ->
# (weakref to the parameter, numel, max|w|).  Reading max|w| per packed weight is a host synchronisation each -- ~900 per training step
# (every weight is re-packed after an optimizer step, forward and transposed for the dgrad): 0.15 s of the 0.42 s step (cProfile, round 4).
_PARAM_MAX = {}


def register_param_maxima(params):
    """Measure max|w| of every fp32 GPU parameter in `params` with one reduction pass and one host read; pack_* calls on those
    parameters (or on tensors derived from them with the same maximum: transposes, flips -- `like=`) then need no synchronisation.
    The owner re-registers after every in-place update (ViewFusion.engine)."""
    import weakref
    ps = [p for p in params if p.dtype == torch.float32 and p.numel() > 0]
    _PARAM_MAX.clear()
    if not ps:
        return 0
    with torch.no_grad():
        try:
            norms = torch._foreach_norm([p.detach().reshape(-1) for p in ps], float("inf"))
        except Exception:
            norms = [torch.linalg.vector_norm(p.detach().reshape(-1), ord=float("inf")) for p in ps]
        vals = torch.stack([n.reshape(()) for n in norms]).tolist()
    for p, v in zip(ps, vals):
        _PARAM_MAX[p.data_ptr()] = (weakref.ref(p), p.numel(), float(v), p._version)
    return len(ps)


def forget_param_maxima():
    """Drop the registry (the owner calls this when parameters may have been replaced wholesale: load_state_dict, .cuda())."""
    _PARAM_MAX.clear()


def _known_max(t):
    """max|t| from the registry if `t` is (a view of the whole of) a registered, still-alive, UNMODIFIED parameter; else None."""
    ent = _PARAM_MAX.get(t.data_ptr())
    if ent is None:
        return None
    ref, numel, mx, version = ent
    p = ref()
    # (an in-place update -- optimizer step, load_state_dict into a submodule -- bumps the parameter's version counter: the
    #  recorded maximum is stale then and the caller falls back to the one-reduction path)
    if p is None or p.data_ptr() != t.data_ptr() or numel != t.numel() or p._version != version:
        return None
    return mx


def _pack_scale(w, like=None):
    """Power of two that brings max|w| to [1024, 2048): the low half of the fp16 split then stays a normal number for
    every weight within 2^-13 of the largest one (exact to undo: the GEMM multiplies its accumulator by 1/scale).
    like: parameter(s) whose elements are exactly those of `w` (a transpose / flip / concatenation of them): their registered maxima
    are used instead of reducing `w` (no host synchronisation)."""
    import math
    mx = None
    srcs = [w] if like is None else (list(like) if isinstance(like, (list, tuple)) else [like])
    known = [_known_max(s.detach()) for s in srcs]
    if all(k is not None for k in known):
        mx = max(known)
    if mx is None:
        mx = float(torch.linalg.vector_norm(w.detach().reshape(-1), ord=float("inf")))      # max|w| in ONE reduction (abs().max() is two kernels and a temporary)
    if not (mx > 0.0) or not math.isfinite(mx):
        return 1.0
    return 2.0 ** (10 - math.floor(math.log2(mx)))


def pack_linear(weight, bias=None, geglu=False, like=None):
    """weight (N, K) fp32 on the GPU (nn.Linear / 1x1 nn.Conv2d weight).  like: see _pack_scale."""
    w = weight.detach().reshape(weight.shape[0], -1).contiguous().float()
    N, K = w.shape
    Np, Kp = (N + 15) // 16 * 16, (K + 31) // 32 * 32
    data = torch.empty(lib().mvd_packed_weight_bytes(N, K), dtype=torch.uint8, device=w.device)
    scale = _pack_scale(w, like if like is not None else weight)
    check(lib().mvd_pack_linear_weight(ptr(w), N, K, K, int(geglu), scale, ptr(data), stream()))
    b = None
    if bias is not None:
        b = torch.zeros(Np, dtype=torch.float32, device=w.device)
        b[:N] = bias.detach().float()
    # (w may be a temporary: the pack kernel runs on torch's current stream, and the caching allocator re-uses a freed block only
    #  for work enqueued later on that stream -- no host synchronisation needed)
    return PackedWeight(data, Np, Kp, N, b, geglu, acc_scale=1.0 / scale)


def pack_linear_t(weight, like=None):
    """The packed image of weight^T -- weight (K, N) fp32 row-major, image (N, K): the dgrad operand dX = dY W of a Linear, packed straight
    from the parameter (no transposed copy; round 6)."""
    w = weight.detach()
    assert w.dim() == 2 and w.dtype == torch.float32 and w.is_contiguous()
    K, N = w.shape
    Np, Kp = (N + 15) // 16 * 16, (K + 31) // 32 * 32
    data = torch.empty(lib().mvd_packed_weight_bytes(N, K), dtype=torch.uint8, device=w.device)
    scale = _pack_scale(w, like if like is not None else weight)
    check(lib().mvd_pack_linear_weight_t(ptr(w), N, K, N, scale, ptr(data), stream()))
    return PackedWeight(data, Np, Kp, N, None, False, acc_scale=1.0 / scale)


def pack_linear_cat(weights):
    """Row-concatenate several (N_i, K) weights (e.g. to_q/to_k/to_v -> one QKV GEMM)."""
    return pack_linear(torch.cat([w.detach().float() for w in weights], dim=0), like=list(weights))


def pack_conv3x3(weight, bias=None, like=None):
    w = weight.detach().contiguous().float()
    Cout, Cin = w.shape[0], w.shape[1]
    cin_pad = (Cin + 31) // 32 * 32
    Np = (Cout + 15) // 16 * 16
    data = torch.empty(Np * 9 * cin_pad * 4, dtype=torch.uint8, device=w.device)
    scale = _pack_scale(w, like if like is not None else weight)
    check(lib().mvd_pack_conv3x3_weight(ptr(w), Cout, Cin, cin_pad, scale, ptr(data), stream()))
    b = None
    if bias is not None:
        b = torch.zeros(Np, dtype=torch.float32, device=w.device)
        b[:Cout] = bias.detach().float()
    return PackedWeight(data, Np, 9 * cin_pad, Cout, b, conv_cin=cin_pad, acc_scale=1.0 / scale)


# ---------------------------------------------------------------------------------------------
# packed-weight cache invalidation
# ---------------------------------------------------------------------------------------------
_CACHE_ATTRS = ("_p", "_pg", "_temb", "_xattn", "_head", "_pq", "_q", "_fused", "_lnf")


def params_signature(module):
    """Fingerprint of a module's parameters: one (storage pointer, autograd version) pair PER tensor, hashed as a tuple -- no
    sum / xor that two changes could cancel.  It changes when any parameter is updated in place through autograd-visible ops
    (load_state_dict, optimizer steps, fill_) or re-allocated (.cuda() / .to()).  Writes through ``p.data`` (``p.data.copy_``,
    EMA swaps via ``.data``, ``torch._foreach_*`` on ``.data``) bump no version and keep the pointer: they are invisible here by
    construction, and ``ViewFusion.invalidate_packed()`` is MANDATORY after them (ViewFusion.load_state_dict / _apply call it
    themselves).  The packed MFMA operand images are derived from the fp32 parameters; their owners compare this before reuse."""
    # (the walk over the module tree -- named_modules / named_parameters, ~600 k Python calls for the full model -- is done ONCE: the slots
    #  (owner's _parameters / _buffers dict, name) are cached on the module and read through on every call, so a Parameter OBJECT replaced
    #  under an existing name is still seen; adding / removing submodules after the first call needs invalidate_packed(), like .data writes)
    slots = module.__dict__.get("_mvd_sig_slots")
    if slots is None:
        ps, bs, seen = [], [], set()
        for m in module.modules():
            for n, p in m._parameters.items():
                if p is not None and id(p) not in seen:
                    seen.add(id(p))
                    ps.append((m._parameters, n))
            for n, b in m._buffers.items():
                if b is not None and id(b) not in seen:
                    seen.add(id(b))
                    bs.append((m._buffers, n))
        slots = module.__dict__["_mvd_sig_slots"] = (ps, bs)
    ps, bs = slots
    sig = []
    for d_, n in ps:
        p = d_.get(n)
        sig.append((p.data_ptr(), p._version) if p is not None else None)
    for d_, n in bs:
        b = d_.get(n)
        sig.append(b.data_ptr() if b is not None else None)
    return hash(tuple(sig))


def drop_packed_caches(module):
    """Forget every lazily packed weight image under `module` (they are re-packed from the live parameters on next use)."""
    module.__dict__.pop("_mvd_sig_slots", None)          # (params_signature's cached slots: re-walk the module tree next time)
    for m in module.modules():
        for name in _CACHE_ATTRS:
            if name in m.__dict__:
                cur = m.__dict__[name]
                if isinstance(cur, dict):
                    m.__dict__[name] = {}
                elif cur is not None:
                    m.__dict__[name] = None


def check_finite(t, what):
    """The 16-bit MFMA operand split has fp16's range (include/mvd_hip.h, "Operand range"): an activation beyond
    +-65504 becomes inf and surfaces as non-finite outputs.  Called once per sample / decode on the final tensor."""
    if not bool(torch.isfinite(t).all()):
        raise FloatingPointError(
            f"{what}: non-finite values. mvd_hip splits GEMM operands into fp16 hi + lo (range +-65504); an activation or "
            "weight outside that range overflowed. Use precision='bf16x3' (bf16 operands, fp32 range) for such weights.")
    return t


# ---------------------------------------------------------------------------------------------
# ops (thin wrappers; all outputs are caller-provided tensors)
# ---------------------------------------------------------------------------------------------
def planes_like(rows, cols, device):
    """Split-planes buffer for a (rows, cols) activation, cols % 32 == 0: int16 tensor (rows, 2*cols); per row and
    32-element block [32 hi | 32 lo] bf16 (csrc/common.hpp)."""
    assert cols % 32 == 0, cols
    return torch.empty(rows, 2 * cols, dtype=torch.int16, device=device)


def sp_cols(p):
    return p.shape[-1] // 2


def split_planes(x, out=None, ldp=None, scale=None):
    """fp32 (rows, cols) -> split planes (rows, ldp) with zero-padded columns (mvd_split_planes); scale: device scalar the values are
    multiplied with on the way (mvd_split_planes_scaled: the power-of-two gradient scale of the backward)."""
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    ldp = (cols + 31) // 32 * 32 if ldp is None else ldp
    if out is None:
        out = planes_like(rows, ldp, x.device)
    if scale is None:
        check(lib().mvd_split_planes(ptr(x), ptr(out), rows, cols, cols, ldp, stream()))
    else:
        check(lib().mvd_split_planes_scaled(ptr(x), ptr(out), rows, cols, cols, ldp, ptr(scale), stream()))
    return out


def gemm(A, W, out=None, *, prec=PREC_X4, M=None, lda=None, bias=True, act=ACT_NONE, res=None, colscale=None,
         bias_b=None, rows_per_batch=0, epi=EPI_STORE, conv=None, qkv=None, workspace=None, splitk=0, ldo=None,
         out_planes=None, out_planes_col=0, cfg=None, gn_stats=None, gn_hw=0, gn_groups=32, row_stats=None, ln=None, gn_apply=None, cat=None,
         acc_scale_dev=None):
    """out = epilogue(A @ W^T).  A: split planes (M, 2*K) int16 (dense) or the NHWC image rows (B*H*W, 2*C) with
    conv=dict(B, Hin, Win, Cin, Hout, Wout, stride, upsample).  qkv = dict(planes=(qh,ql,kh,kl,vh,vl), heads, dhead, L).
    out: fp32 tensor or None; out_planes: split-planes tensor or None (feeds the next GEMM); out_planes_col: first column
    (multiple of 32) of a WIDER planes buffer that receives the output -- the GEMM then fills columns
    [out_planes_col, out_planes_col + N) of every row and leaves the others alone (operand concatenation along K).
    gn_stats: zeroed int64 (M / gn_hw, gn_groups, 2) tensor that receives the GroupNorm statistics of the output (consumed by
    groupnorm_from_stats instead of a statistics kernel).
    row_stats = RowStats: the GEMM also emits per-row {sum, sum of squares} slots of its output (for a LayerNorm folded into the
    consumer); ln = (RowStats of the producer of A, LnFold of this weight): LayerNorm(A rows) folded into this QKV / GEGLU GEMM.
    gn_apply = (gamma, beta, eps, flags, planes): the GroupNorm consuming `out` is applied behind the GEMM (needs gn_stats): `planes`
    receives act(GroupNorm(out)) -- fused with the split-K reduce when the GEMM splits (mvd_gemm_desc.gna_out_sp; flags GNA_*).
    cat = (skip (M, cb) fp32, raw planes (M, 2 * (N + cb)) or None): the GroupNorm of gn_apply runs over [out | skip] (mvd_gemm_desc.cat_b).
    acc_scale_dev: device scalar multiplied into the accumulator scale (mvd_gemm_desc.acc_scale_dev: the backward's 1 / gradient scale).
    """
    assert A.dtype == torch.int16, "A must be in split-planes format (see hip.split_planes)"
    d = GemmDesc()
    d.N, d.K = W.N, W.K
    d.A = A.data_ptr()
    d.Wp = W.data.data_ptr()
    if isinstance(W, PlanesOperand):
        d.b_mode, d.ldb = 1, W.ld
    d.acc_scale = W.acc_scale
    if acc_scale_dev is not None:
        d.acc_scale_dev = acc_scale_dev.data_ptr()
    d.prec = prec
    if conv is not None:
        d.a_mode = A_CONV3X3
        for k in ("B", "Hin", "Win", "Cin", "Hout", "Wout", "stride", "upsample"):
            setattr(d, k, int(conv[k]))
        d.no_pad_tl = int(conv.get("no_pad_tl", 0))
        d.M = conv["B"] * conv["Hout"] * conv["Wout"]
        assert conv["Cin"] * 9 == W.K, (conv["Cin"], W.K)
    else:
        d.a_mode = A_DENSE
        d.M = int(M if M is not None else A.numel() // A.shape[-1])
        d.lda = int(lda if lda is not None else A.shape[-1] // 2)
        assert d.lda >= W.K, f"A has {d.lda} columns, packed K is {W.K} (pad A)"
    d.epi, d.act = epi, act
    if out is not None:
        d.out = out.data_ptr()
        d.ldo = int(ldo if ldo is not None else out.shape[-1])
    if out_planes is not None:
        assert out_planes_col % 32 == 0 and out_planes.dtype == torch.int16
        d.out_sp = out_planes.data_ptr() + 4 * int(out_planes_col)     # element (row, k) sits (k >> 5) * 128 bytes into its row
        d.ldp = int(out_planes.shape[-1] // 2)
    d.n_store = W.n_real
    if bias and W.bias is not None:
        d.bias = W.bias.data_ptr()
    if bias_b is not None:
        d.bias_b = bias_b.data_ptr()
        d.rows_per_batch = int(rows_per_batch)
        d.ldbb = bias_b.stride(0) if bias_b.dim() == 2 else 0
    if colscale is not None:
        d.colscale = colscale.data_ptr()
    if res is not None:
        d.res = res.data_ptr()
        d.ldr = int(res.shape[-1])
    if qkv is not None:
        qh, ql, kh, kl, vh, vl = qkv["planes"]
        d.q_hi, d.q_lo, d.k_hi, d.k_lo, d.vt_hi, d.vt_lo = (t.data_ptr() for t in (qh, ql, kh, kl, vh, vl))
        d.heads, d.dhead, d.L = qkv["heads"], qkv["dhead"], qkv["L"]
        d.Lpad = lib().mvd_attn_lpad(qkv["L"])
        d.qscale = float(qkv["dhead"]) ** -0.5 * 1.4426950408889634      # * log2(e): mvd_attention works in base 2
    if gn_stats is not None:
        d.gn_stats, d.gn_hw, d.gn_groups = gn_stats.data_ptr(), int(gn_hw), int(gn_groups)
    if gn_apply is not None:
        gamma, beta, eps, flags, planes = gn_apply
        ct = d.N + (cat[0].shape[-1] if cat is not None else 0)
        assert gn_stats is not None and out is not None and planes.dtype == torch.int16 and planes.shape[-1] == 2 * ct
        if cat is not None:
            skip, raw = cat
            assert skip.dtype == torch.float32 and skip.is_contiguous() and skip.shape[0] == d.M and (raw is None or raw.shape[-1] == 2 * ct)
            d.cat_b, d.cat_cb, d.cat_raw_sp = skip.data_ptr(), int(skip.shape[-1]), (raw.data_ptr() if raw is not None else None)
        d.gna_gamma, d.gna_beta, d.gna_eps, d.gna_flags, d.gna_out_sp = gamma.data_ptr(), beta.data_ptr(), float(eps), int(flags), planes.data_ptr()
    if row_stats is not None:
        assert row_stats.slots.shape[0] >= d.M and row_stats.ld >= (d.N + 31) // 32
        d.rs_out, d.rs_count, d.rs_ld = row_stats.slots.data_ptr(), row_stats.count.data_ptr(), row_stats.ld
    if ln is not None:
        rs, fold = ln
        d.ln_stats, d.ln_count, d.ln_ld = rs.slots.data_ptr(), rs.count.data_ptr(), rs.ld
        d.ln_colsum, d.ln_dim, d.ln_eps = fold.colsum.data_ptr(), fold.dim, fold.eps
        splitk = 1
    d.splitk = splitk
    if workspace is not None:
        d.workspace = workspace.data_ptr()
        d.workspace_elems = workspace.numel()
    key = (d.M, d.N, d.K, d.a_mode, d.Cin, d.stride, d.upsample, d.no_pad_tl, d.epi, d.prec, res is not None, out is not None,
           out_planes is not None, splitk, d.b_mode, row_stats is not None, ln is not None, gn_apply is not None, cat[0].shape[-1] if cat is not None else 0)
    if cfg is None:
        tuned = _TUNED.get(key)
        if tuned is None and AUTOTUNE and 2.0 * d.M * d.N * d.K >= AUTOTUNE_MIN_FLOPS:
            tuned = _autotune(d, A if A.is_contiguous() else None, W_data=W.data if not isinstance(W, PlanesOperand) else None)
            _TUNED[key] = tuned
            global TUNED_IN_RUN
            TUNED_IN_RUN += 1
        if tuned is not None:
            cfg, d.splitk = tuned
    d.cfg = cfg or 0
    global LAST_CFG
    LAST_CFG = d.cfg
    if GEMM_SEQUENCE is not None:            # a step engine follows its GEMM launches (weight prefetch, see WeightPrefetcher)
        GEMM_SEQUENCE.note(d, W)
    check(lib().mvd_gemm(C.byref(d), stream()))
    return out


GEMM_SEQUENCE = None      # set by a step engine around ITS launches (WeightPrefetcher.following())
SELF_PREFETCH_MIN = int(os.environ.get("MVD_SELF_PREFETCH_MIN", "0"))      # bytes; see WeightPrefetcher (0 = off)


class WeightPrefetcher:
    """Host side of the weight prefetch of ONE graph-captured step (include/mvd_hip.h: mvd_gemm_desc.pf_items).  While
    `following(record=True)` is active (a second eager pass behind the tuned warm-up step) every hip.gemm launch appends its packed weight
    and kernel kind to the launch-order list; `following()` during capture then hands every host launch its share: every role-split launch
    (gemm_ws_kernel, the long convolutions: idle consumer wavefronts, idle HBM) requests the weights of the launches that follow it, up to
    and including the next role-split launch and `window` bytes (24 MB: the sweep of profiles/r05_prefetch_ab.log -- 4 ... 32 MB all gain
    1.2 - 1.8 % of the step, 64 MB less, 128 MB loses: the big low-resolution weights push the step's activations out of the Infinity Cache).
    (Round 5 also measured a stand-alone prefetch kernel on a parallel graph branch: 12 % SLOWER than no prefetch, DESIGN.md section 6.00;
    removed from the library in round 6, source under tools/probes/prefetch.hip.)
    """

    def __init__(self, device, window=24 << 20, max_items=24, gn_hosts=0, self_min=None):
        self.device = device
        self.window, self.max_items = int(window), int(max_items)
        # SELF prefetch (round 6): a role-split launch whose own packed weight is at least `self_min` bytes also requests ITS OWN weight at
        # kernel start (first item of its share, outside the window: those bytes pass through the caches in this launch anyway) -- the
        # consumers' requests run ahead of the loaders' k-tile-by-k-tile DMAs, which then meet their lines in L2 / Infinity Cache instead
        # of paying an HBM round trip per k-tile (the "cold weight" penalty of the low-resolution levels).  0 = off.
        self.self_min = int(SELF_PREFETCH_MIN if self_min is None else self_min)
        # gn_hosts: the fused reduce + GroupNorm kernels of split GEMMs host prefetch shares too.  Measured (profiles/r05_prefetch_ab.log):
        # 8.25 - 8.27 ms against 8.21 - 8.24 with the role-split hosts alone (8.33 - 8.38 without any prefetch): the requests lengthen the tail
        # of a 10 us kernel by about what they save the next one.  Off by default.
        self.gn_hosts = bool(gn_hosts)
        self.seq, self.recording, self.table, self.n, self.launch_idx, self.shares = [], False, None, 0, 0, {}

    class _Following:
        def __init__(self, pf, record):
            self.pf, self.record = pf, record

        def __enter__(self):
            global GEMM_SEQUENCE
            self.prev, GEMM_SEQUENCE = GEMM_SEQUENCE, self.pf
            self.pf.recording = self.record
            self.pf.launch_idx = 0
            if self.record:
                self.pf.seq = []
            return self.pf

        def __exit__(self, *exc):
            global GEMM_SEQUENCE
            GEMM_SEQUENCE = self.prev
            if self.record and exc[0] is None:
                self.pf._build()
            self.pf.recording = False
            return False

    def following(self, record=False):
        return WeightPrefetcher._Following(self, record)

    def note(self, d, W):
        """Called by hip.gemm right before the launch (d.cfg is final)."""
        j = self.launch_idx
        self.launch_idx += 1
        # hosts of the in-kernel prefetch: 1 = the role-split kernel (certain); 2 = a GEMM whose GroupNorm is applied behind it and that MAY
        # split K -- then its fused reduce + GroupNorm kernel hosts (the library decides the split; an unsplit launch ignores its share)
        is_ws = bool(d.cfg) and _cfg_parts(d.cfg)[1] == WS_LOOP
        host = 1 if is_ws else (2 if (d.gna_out_sp and d.splitk != 1 and self.gn_hosts) else 0)
        if self.recording:
            packed = not isinstance(W, PlanesOperand)
            self.seq.append((W.data.data_ptr(), W.data.numel() * W.data.element_size(), host) if packed else (0, 0, host))
            return
        if host and j in self.shares:
            first, n = self.shares[j]
            d.pf_items, d.pf_n = self.table.data_ptr() + first * C.sizeof(PrefetchItem), n

    def _build(self):
        items = []
        self.shares = {}
        # host j takes the weights of launches j + 1 .. (next CERTAIN host), first come first served inside `window`; the shares of
        # "maybe" hosts overlap those of the certain host before them (a second request of a resident line is cheap)
        hosts = [j for j, e in enumerate(self.seq) if e[2]]
        sure = [j for j, e in enumerate(self.seq) if e[2] == 1]
        for j in hosts:
            nxt = [k for k in sure if k > j]
            end = nxt[0] if nxt else len(self.seq) - 1
            first, acc, seen = len(items), 0, set()
            if self.self_min > 0 and self.seq[j][2] == 1 and self.seq[j][1] >= self.self_min:
                items.append((self.seq[j][0], self.seq[j][1], j, j))
                seen.add(self.seq[j][0])
            for k in range(j + 1, end + 1):
                ptr_, nbytes, _ = self.seq[k]
                if not nbytes or ptr_ in seen or acc + nbytes > self.window or len(items) - first >= self.max_items:
                    continue
                seen.add(ptr_)
                acc += nbytes
                items.append((ptr_, nbytes, j, k))
            if len(items) > first:
                self.shares[j] = (first, len(items) - first)
        self.n = len(items)
        arr = (PrefetchItem * max(self.n, 1))()
        for i, (ptr_, nbytes, sa, cons) in enumerate(items):
            arr[i].ptr, arr[i].bytes, arr[i].start_after, arr[i].consumer = ptr_, nbytes, sa, cons
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone().to(self.device)
        self.items = items


# mvd_gemm_desc.cfg = 1 + CFG_STRIDE * tile + 2 * loop + order (include/mvd_hip.h: MVD_GEMM_CFG_STRIDE)
CFG_STRIDE = 32
GNA_SILU, GNA_ROUND_F16, GNA_OUT_UNUSED = 1, 2, 4      # mvd_gemm_desc.gna_flags
TUNE_CACHE_VERSION = 11            # bump when the cfg encoding or the tuner's problem key changes (save_tuned / load_tuned)
GEMM_TILES = ((64, 64, 2, 2), (128, 128, 2, 4), (128, 80, 4, 1), (64, 80, 4, 1), (128, 160, 4, 2))     # BM, BN, WM, WN
GEMM_LOOPS = (2, 3, 4, None, 6, 7, "patch", "ws")  # template STAGES of gemm_kernel: 2 = plain, 3 = register-pipelined, 4 = staggered wave
                                 # groups (3 LDS buffers), 6 / 7 = register-pipelined over a ring of <= 4 / <= 8 LDS buffers; "patch" = conv_patch_kernel
                                 # (stride-1 3x3 convolutions: the input patch is staged once per channel block); None = loop variants removed in
                                 # round 5 (never selected by the tuner; the numbering of the others is unchanged -- include/mvd_hip.h: cfg)
REMOVED_LOOPS = tuple(i for i, l in enumerate(GEMM_LOOPS) if l is None)
PATCH_LOOP = 6
WS_LOOP = 7                      # gemm_ws_kernel: consumer / loader wavefront roles (tiles 1, 2, 4; EPI_STORE)


def _cfg_parts(cfg):
    return (cfg - 1) // CFG_STRIDE, ((cfg - 1) % CFG_STRIDE) >> 1, (cfg - 1) & 1          # tile, loop, order


def make_cfg(tile, loop, order=0):
    return 1 + CFG_STRIDE * tile + 2 * loop + order


def _cfg_valid(cfg, epi, b_mode=0):
    tile, loop, _ = _cfg_parts(cfg)
    if loop >= len(GEMM_LOOPS):
        return False
    bm, bn, wm, wn = GEMM_TILES[tile]
    waves = wm * wn
    return loop not in REMOVED_LOOPS and (loop != 2 or waves == 8) and (loop != 5 or waves == 4) and \
        (loop != PATCH_LOOP or tile in (1, 2, 4)) and (tile < 2 or epi == EPI_STORE) and \
        (loop != WS_LOOP or tile == 1 or (tile in (2, 4) and epi == EPI_STORE))


_ALL_CONFIGS = tuple(c for c in range(1, CFG_STRIDE * len(GEMM_TILES) + 1) if _cfg_valid(c, EPI_STORE))
GEMM_CONFIGS = tuple(c for c in _ALL_CONFIGS if _cfg_parts(c)[1] != PATCH_LOOP)       # serve every problem kind (dense and conv)
PATCH_CONFIGS = tuple(c for c in _ALL_CONFIGS if _cfg_parts(c)[1] == PATCH_LOOP)      # stride-1 padded 3x3 convolutions only
GEMM_CONFIGS_CONV = GEMM_CONFIGS + PATCH_CONFIGS


def gemm_configs(epi=EPI_STORE, b_mode=0, conv=False):
    """Kernel configurations valid for an epilogue / B operand kind (the 80-column tiles serve EPI_STORE only; the staggered loop
    needs an 8-wave tile; every configuration serves both B operand kinds); conv: + the input-patch kernel (whether it takes a
    given convolution: cfg_supported)."""
    return tuple(c for c in (GEMM_CONFIGS_CONV if conv else GEMM_CONFIGS) if _cfg_valid(c, epi, b_mode))


def cfg_supported(d, cfg):
    """Whether kernel configuration `cfg` serves the problem of the GemmDesc `d` (mvd_gemm_cfg_supported: e.g. the input-patch loop
    only takes stride-1 padded 3x3 convolutions whose tiles are whole image rows / whole images)."""
    return bool(lib().mvd_gemm_cfg_supported(C.byref(d), cfg))


def kernel_symbol(cfg, prec, conv):
    """The gemm_kernel<BM, BN, WM, WN, NS, AMODE, LOOP> instantiation (as rocprofv3 prints it) that `cfg` selects."""
    if not cfg:
        return f"gemm_kernel<auto, {prec}, {1 if conv else 0}>"
    tile, loop, _ = _cfg_parts(cfg)
    bm, bn, wm, wn = GEMM_TILES[tile]
    if loop == PATCH_LOOP:
        return f"conv_patch_kernel<{bm}, {bn}, {wm}, {wn}, {prec}>"
    if loop == WS_LOOP:
        cm, cn = {1: (2, 2), 2: (4, 1), 4: (2, 2)}[tile]
        return f"gemm_ws_kernel<{bm}, {bn}, {cm}, {cn}, {prec}, {1 if conv else 0}>"
    return f"gemm_kernel<{bm}, {bn}, {wm}, {wn}, {prec}, {1 if conv else 0}, {GEMM_LOOPS[loop]}>"


def gemm_fingerprint():
    """sha256 over the sources that decide how fast each GEMM configuration runs (the GEMM kernels and their shared device code; the cfg
    ENCODING is guarded separately by TUNE_CACHE_VERSION / CFG_STRIDE): a tuner cache records it, and a cache measured on other kernels is
    not used (the problems are re-tuned in the run)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(_HERE)
    pats = ("mvdfusion_amd/csrc/gemm*.hip", "mvdfusion_amd/csrc/gemm*.hpp", "mvdfusion_amd/csrc/common.hpp")
    for f in sorted(f for pat in pats for f in glob.glob(os.path.join(root, pat))):
        h.update(os.path.relpath(f, root).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def default_tuned_path(fmt=None):
    """The COMMITTED tuner choices of the BASELINE workloads (tools/tune_all.py writes them on a GPU box): loaded when the library is, so
    that the driver's bench run, the step traces and the counter passes of profiles/ all launch ONE kernel mix."""
    return os.path.join(_HERE, "tuned", f"gemm_{fmt or OPERAND_FORMAT}.json")


TUNED_IN_RUN = 0          # problems the tuner timed in this process (0 = every problem came from the cache)
TUNED_SOURCE = None       # what load_default_tuned() did (bench.py prints it: config.gemm_tuning)


def load_default_tuned():
    """MVD_TUNE_CACHE=0: none (every problem is tuned in the run); MVD_TUNE_CACHE=<file>: that file; default: default_tuned_path()."""
    global TUNED_SOURCE
    env = os.environ.get("MVD_TUNE_CACHE", "")
    if env == "0":
        TUNED_SOURCE = "none (MVD_TUNE_CACHE=0): tuned in this run"
        return 0
    path = env or default_tuned_path()
    if not os.path.exists(path):
        TUNED_SOURCE = f"none ({os.path.relpath(path, os.path.dirname(_HERE))} missing): tuned in this run"
        return 0
    # (MVD_TUNE_CACHE_ANY=1: same-box A/B runs of an edited kernel against the committed choices -- both legs launch one kernel mix)
    n = load_tuned(path, require_fingerprint=os.environ.get("MVD_TUNE_CACHE_ANY") != "1")
    rel = os.path.relpath(path, os.path.dirname(_HERE))
    TUNED_SOURCE = f"{rel}: {n} problems" if n else f"none ({rel} was tuned on other GEMM sources or another cfg encoding): tuned in this run"
    return n


def save_tuned(path, merge=False):
    """Persist the autotuner's choices (problem key -> (cfg, splitk)) so that separate processes -- the bench run and the rocprofv3
    counter passes of one profiling session -- launch identical kernels.  merge: keep the entries of an existing file of the same
    encoding AND the same GEMM sources for problems this process did not meet."""
    import json
    entries = {}
    if merge and os.path.exists(path):
        doc = json.load(open(path))
        if _tuned_doc_ok(doc) and doc.get("gemm_fingerprint") == gemm_fingerprint():
            entries = {tuple(k): tuple(v) for k, v in doc["entries"]}
    entries.update(_TUNED)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        json.dump({"version": TUNE_CACHE_VERSION, "cfg_stride": CFG_STRIDE, "operand_format": OPERAND_FORMAT,
                   "gemm_fingerprint": gemm_fingerprint(),
                   "entries": sorted([list(k), list(v)] for k, v in entries.items())}, f, indent=0)


def _tuned_doc_ok(doc):
    return isinstance(doc, dict) and doc.get("version") == TUNE_CACHE_VERSION and doc.get("cfg_stride") == CFG_STRIDE and \
        doc.get("operand_format") == OPERAND_FORMAT


def load_tuned(path, require_fingerprint=False):
    """Load a cache written by save_tuned.  A file of another encoding (version / cfg stride / operand format) is REJECTED as a whole,
    and entries whose cfg is not a valid configuration of this build are dropped (an old cache must never launch the wrong
    kernels silently).  require_fingerprint: also reject a file measured on other GEMM sources (gemm_fingerprint).  Returns the number of
    entries taken."""
    import json
    doc = json.load(open(path))
    if not _tuned_doc_ok(doc) or (require_fingerprint and doc.get("gemm_fingerprint") != gemm_fingerprint()):
        return 0
    n = 0
    for k, v in doc["entries"]:
        cfg = int(v[0])
        if cfg != 0 and not (1 <= cfg <= CFG_STRIDE * len(GEMM_TILES) and _cfg_parts(cfg)[1] < len(GEMM_LOOPS) and
                             _cfg_parts(cfg)[1] not in REMOVED_LOOPS):
            continue
        _TUNED[tuple(k)] = tuple(v)
        n += 1
    return n


LAST_CFG = 0
AUTOTUNE = False          # set by the step engine around its eager warm-up step (never during graph capture)
AUTOTUNE_MIN_FLOPS = 0.0  # problems below this many FLOPs keep the library's heuristic (the training step tunes its big GEMMs only)
_TUNED = {}


_TRASH = None
TUNE_SPLITS = os.environ.get("MVD_TUNE_SPLITS") == "1"      # the tuner also times explicit split-K counts (see _autotune)
TUNE_COLD = True          # time the candidates with COLD weights (see _autotune)


def _flush_caches():
    """Evict L2 and the 256 MB Infinity Cache: stream a 640 MB buffer (read + write)."""
    global _TRASH
    if _TRASH is None:
        _TRASH = torch.zeros(160 * 1024 * 1024, dtype=torch.float32, device="cuda")
    _TRASH.add_(1.0)


def release_tuning_buffers():
    """Free the 640 MB cache-flush buffer of the cold-weight tuner (the callers that open a tuning window -- the step engine's eager
    warm-up, the VAE's first encode / decode of a shape -- call this when they close it; the next window re-allocates it)."""
    global _TRASH
    _TRASH = None


_TUNE_STATS = {}


def _autotune(d, A=None, reps=4, trials=int(os.environ.get("MVD_TUNE_TRIALS", "3")), W_data=None):
    """Time the kernel configurations (tile x loop variant x tile order; split-K follows from the library's model)
    on the actual operands -- the op is idempotent -- and return the fastest.

    In a denoising step every GEMM meets its weights COLD (the 4 GB weight set cycles through a 256 MB Infinity Cache) while
    its activations were just written by the previous kernel.  Back-to-back timing of one problem keeps the weights cached and
    ranks the configurations differently (the deeper-prefetch loops only pay off on cold weights), so by default each timed
    launch is preceded by a cache flush and a re-read of the A operand; min over `trials` single launches."""
    best, best_ms = (0, d.splitk), float("inf")
    e0, e1 = Event(), Event()
    stats_ptr, d.gn_stats = d.gn_stats, None          # the statistics atomics must run exactly once: only in the real launch
    if d.gna_out_sp:                                   # GroupNorm apply behind the GEMM: needs statistics -- a scratch slot while timing (the
        n = (d.M // d.gn_hw) * d.gn_groups * 2         # planes it writes are rewritten by the real launch)
        n = (n, str(A.device) if A is not None else "cuda")          # (the scratch slot lives on the device of the operands: ADVICE r04)
        scratch = _TUNE_STATS.get(n)
        if scratch is None:
            scratch = _TUNE_STATS[n] = torch.zeros(n[0], dtype=torch.int64, device=A.device if A is not None else "cuda")
        d.gn_stats = scratch.data_ptr()
    # split-K: the library's model (0 = auto) or none (1); the timed region includes the reduce kernel of a split GEMM
    # Loops the tuner does not time unless asked (MVD_TUNE_INCLUDE_LOOPS=8,9): the two register-staged delivery paths of round 4 were
    # candidates for a whole session and were selected for NO shape of any workload (profiles/r04_ws_variants.json);
    # they stay built, tested (test_gemm_configurations_agree) and selectable by cfg.  MVD_TUNE_EXCLUDE_LOOPS: A/B measurements.
    skip = set()
    skip |= {int(t) for t in os.environ.get("MVD_TUNE_EXCLUDE_LOOPS", "").split(",") if t}      # (e.g. "7" = no role-split kernel)
    fixed_split = d.splitk
    cands = [(c, sk) for c in gemm_configs(d.epi, d.b_mode, conv=d.a_mode == A_CONV3X3) if cfg_supported(d, c) and _cfg_parts(c)[1] not in skip
             for sk in ((0, 1) if d.splitk == 0 else (d.splitk,))]
    timed, second_pass = [], False
    # three cold copies of the packed weight, launched back to back between one pair of events: the eager launch latency (a few us
    # of jitter, the size of the differences being ranked) is paid once per three kernels and hides behind the first one
    wp0 = d.Wp
    rot = [wp0]
    if TUNE_COLD and A is not None and d.b_mode == 0:
        if W_data is not None and W_data.data_ptr() == wp0:
            rot += [W_data.clone(), W_data.clone()]
    ptrs = [wp0] + [t.data_ptr() for t in rot[1:]]
    for cfg, sk in cands:
        d.cfg, d.splitk = cfg, sk
        check(lib().mvd_gemm(C.byref(d), stream()))
        ms = float("inf")
        if TUNE_COLD and A is not None:
            for _ in range(trials + 1):
                _flush_caches()
                A.view(-1)[:A.numel() // 2 * 2].view(torch.int32).sum()        # the producer of A just ran: A is cache-warm
                e0.record()
                for wp in ptrs:
                    d.Wp = wp
                    check(lib().mvd_gemm(C.byref(d), stream()))
                e1.record()
                ms = min(ms, e0.elapsed_ms(e1) / len(ptrs))
            d.Wp = wp0
        else:
            for _ in range(trials):
                e0.record()
                for _ in range(reps):
                    check(lib().mvd_gemm(C.byref(d), stream()))
                e1.record()
                ms = min(ms, e0.elapsed_ms(e1))
        # a split GEMM drags a second launch (~1.5 us of boundary inside the captured graph that event timing of eager
        # launches does not see): it has to win by that margin
        if sk != 1:
            ms += 0.0015
        timed.append((ms, cfg, sk))
        if ms < best_ms * 0.99:
            best, best_ms = (cfg, sk), ms
        if len(timed) == len(cands) and TUNE_SPLITS and fixed_split == 0 and not second_pass:
            second_pass = True
            # second pass (tools/tune_all.py: the committed cache): explicit split-K counts around the library's model for the three fastest
            # configurations -- the model is analytic (csrc/gemm.hip: choose_splits) and the low-resolution levels live on it
            nk = d.K // 32
            extra = [s_ for s_ in (2, 3, 4, 6, 8, 12, 16, 24, 32) if s_ <= max(1, nk // 2)]
            top = sorted(timed)[:3]
            cands.extend((c, s_) for _, c, _ in top for s_ in extra if (c, s_) not in cands)
    d.gn_stats = stats_ptr
    return best


def gemv(W, bias, x, y, act_in=ACT_NONE, act_out=ACT_NONE):
    """y (M,N) = act_out(act_in(x) (M,K) @ W^T (N,K) + bias), fp32 exact, M <= 16."""
    M, K = x.shape
    N = W.shape[0]
    check(lib().mvd_gemv(ptr(W), ptr(bias), ptr(x), ptr(y), M, N, K, x.stride(0), y.stride(0), act_in, act_out, stream()))
    return y


def groupnorm(x, y, gamma, beta, B, HW, Cc, eps, silu, ws):
    """y: split planes (B*HW, 2*C)."""
    check(lib().mvd_groupnorm_nhwc(ptr(x), ptr(y), ptr(gamma), ptr(beta), B, HW, Cc, 32, eps, int(silu), ptr(ws), ws.numel(),
                                   stream()))
    return y


def groupnorm_from_stats(x, y, gamma, beta, stats, B, HW, Cc, eps, silu, groups=32):
    """GroupNorm apply with producer-emitted statistics (see gemm(gn_stats=...)); y: split planes (B*HW, 2*C)."""
    check(lib().mvd_groupnorm_from_stats(ptr(x), ptr(y), ptr(gamma), ptr(beta), ptr(stats), B, HW, Cc, groups, eps, int(silu), stream()))
    return y


def layernorm(x, y, w, b, rows, Cc, eps=1e-5, w_plus_one=False, y_f32=None):
    """y: split planes (rows, 2*C) (or None); y_f32: optional fp32 (rows, C) copy of the result."""
    check(lib().mvd_layernorm(ptr(x), ptr(y), ptr(y_f32), ptr(w), ptr(b), rows, Cc, eps, int(w_plus_one), stream()))
    return y


def softmax_rows(x, y, scale=1.0, out_scale=1.0):
    """y (rows, 2*cols) split planes = out_scale * softmax(scale * x) along the last dim of the fp32 matrix x (rows, cols)."""
    rows, cols = x.shape
    check(lib().mvd_softmax_rows(ptr(x), ptr(y), rows, cols, x.stride(0), float(scale), float(out_scale), stream()))
    return y


def attention(planes, out, B, heads, L, dhead, prec=PREC_X4, Lkeys=0):
    """out: split planes (B*L, 2*heads*dhead).  Lkeys (0 = L): leading tokens of each sequence that act as keys."""
    qh, ql, kh, kl, vh, vl = planes
    check(lib().mvd_attention(ptr(qh), ptr(ql), ptr(kh), ptr(kl), ptr(vh), ptr(vl), ptr(out), out.shape[-1] // 2, B, heads,
                              L, Lkeys, dhead, prec, stream()))
    return out


def alloc_attn_planes(B, heads, L, dhead, device):
    nqk = lib().mvd_attn_qk_plane_elems(B, heads, L, dhead)
    nvt = lib().mvd_attn_vt_plane_elems(B, heads, L, dhead)
    mk = lambda n: torch.zeros(n, dtype=torch.int16, device=device)  # zero padding is part of the contract
    return (mk(nqk), mk(nqk), mk(nqk), mk(nqk), mk(nvt), mk(nvt))


class Graph:
    """A captured hipGraph of everything enqueued on the current stream inside the `with` block."""

    def __init__(self):
        self.exec = C.c_void_p()

    def __enter__(self):
        self._side = torch.cuda.Stream()
        self._side.wait_stream(torch.cuda.current_stream())
        self._ctx = torch.cuda.stream(self._side)
        self._ctx.__enter__()
        check(lib().mvd_graph_begin(stream()))
        return self

    def __exit__(self, et, ev, tb):
        rc = lib().mvd_graph_end(stream(), C.byref(self.exec))
        self._ctx.__exit__(et, ev, tb)
        torch.cuda.current_stream().wait_stream(self._side)
        if et is None:
            check(rc)
        return False

    def launch(self):
        check(lib().mvd_graph_launch(self.exec, stream()))

    def __del__(self):
        try:
            if self.exec:
                lib().mvd_graph_destroy(self.exec)
        except Exception:
            pass


class Event:
    def __init__(self):
        self.h = C.c_void_p()
        check(lib().mvd_event_create(C.byref(self.h)))

    def record(self):
        check(lib().mvd_event_record(self.h, stream()))

    def elapsed_ms(self, stop):
        ms = C.c_float()
        check(lib().mvd_event_elapsed_ms(self.h, stop.h, C.byref(ms)))
        return ms.value
