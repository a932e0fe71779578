"""Execution context shared by the HIP-backed module mirrors: static workspace arena, per-step device tables.

Design (MI355X-first): every intermediate of a denoising step lives in a buffer that is allocated ONCE (first eager
run) and re-used with a fixed address afterwards, so a whole step -- GridAttn + the classifier-free-guidance pair of
UNet passes (batched as 2V views: one sweep of the 4 GB of weights) + the DDIM update -- is allocation-free and can be
captured in a single hipGraph and replayed 50 times with no host work in between.  Scratch buffers are shared between
layers of the same shape (keeps the working set inside the 256 MB Infinity Cache); 288 GB of HBM makes the arena size
a non-issue.
"""
import os

import torch

from . import hip


GN_SLOTS = 256                  # GroupNorm'd tensors per step (the UNet has 61 + 16 + 16)
GN_SLOT_ELEMS = 64 * 32 * 2     # up to 64 images per batch, 32 groups, {sum, sum of squares}


class Workspace:
    """Named static buffers.  get(tag, shape) returns the same tensor (same address) on every call."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.bufs = {}
        self.frozen = False         # True while a hipGraph is being captured

    def get(self, tag, shape, dtype=torch.float32, zero=False):
        key = (tag, tuple(int(s) for s in shape), dtype)
        t = self.bufs.get(key)
        if t is None:
            assert not self.frozen, f"workspace buffer {key} requested during graph capture (the eager warm-up step must " \
                                    "allocate every buffer the captured step uses)"
            t = (torch.zeros if zero else torch.empty)(key[1], dtype=dtype, device=self.device)
            self.bufs[key] = t
        return t

    def planes(self, tag, rows, cols):
        """Static split-planes buffer (rows, 2*cols) int16 (the A-operand format of mvd_gemm, csrc/common.hpp)."""
        assert cols % 32 == 0, (tag, cols)
        return self.get(tag, (rows, 2 * cols), torch.int16)

    def attn_planes(self, B, heads, L, dhead):
        key = ("attn_planes", B, heads, L, dhead)
        t = self.bufs.get(key)
        if t is None:
            assert not self.frozen, f"attention planes {key} requested during graph capture"
            t = hip.alloc_attn_planes(B, heads, L, dhead, self.device)
            self.bufs[key] = t
        return t

    def nbytes(self):
        n = 0
        for v in self.bufs.values():
            for t in (v if isinstance(v, tuple) else (v,)):
                n += t.numel() * t.element_size()
        return n


class Ctx:
    """Per-model execution context handed down the module tree."""
    gn_from_producer = True         # False: every GroupNorm runs its own statistics kernel (A/B measurement, bench.py --gn-two-pass)
    ln_fold = True                  # False: LayerNorm kernels instead of the fold into the QKV / GEGLU GEMMs (A/B: bench.py --ln-kernels)
    gn_fuse = True                  # False: GroupNorm apply always as its own launch (A/B: bench.py --gn-apply-kernels)
    # keep_fp32 (instance attribute, see __init__): True during training forwards -- the backward reads the blocks' fp32 inputs from the
    # workspace, so fused producers write every fp32 tensor they would otherwise skip (the decoder's concatenations)

    def __init__(self, device, prec=hip.PREC_X4, policy=None):
        self.device = torch.device(device)
        self.ws = Workspace(device)
        self.prec = prec            # default number of partial products (hip.PREC_*)
        self.policy = dict(policy or {})      # layer class (hip.PREC_KINDS) -> products, where it differs from the default
        self.B = 0                  # views in the UNet batch (2V with classifier-free guidance)
        self.D = 1                  # depth samples per ray
        self.context = None         # (B, 768) projected CLIP context
        self.vol_levels = None      # list of (B*h*w*D, 768)
        self.emb_bias = None        # dict: ResBlock -> (Cout,) slice of the per-step time-embedding biases
        self._rot = {}
        self.capturing = False      # set by StepEngine around graph capture: no new workspace buffer may appear then
        self.gemm_ws = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=self.device)  # 256 MB split-K slabs
        # GroupNorm statistics emitted by the producers of the normalised tensors (GEMM epilogue / split-K reduce / concat):
        # one (B, 32, 2) int64 slot per produced tensor per step, zeroed in one launch at the start of the step.
        self.gn_arena = torch.zeros(GN_SLOTS * GN_SLOT_ELEMS, dtype=torch.int64, device=self.device)
        self._gn_next = 0
        self._gn = {}               # data_ptr of a produced tensor -> (stats slot, B, HW, C)
        self._rs = {}               # (tag, rows, width) -> hip.RowStats (static: part of the captured graph)
        # GroupNorm of the NEXT layer, applied by the GEMM that produces this layer's output (TimestepEmbedSequential sets the hint:
        # (norm module, name of the consumer's planes buffer, silu)); _gn_done: (tensor, norm) -> planes already holding the result
        self.next_gn = None
        self._gn_done = {}
        self._unwritten = set()     # data_ptr of fp32 tensors a fused producer did NOT write (their normalised planes exist instead)
        # the decoder's concat + GroupNorm, done by the GEMM that produces the block output (UNetModel.run sets the hint for a block's last
        # layer): (skip tensor, cat buffer, raw planes buffer, norm module, planes buffer name, silu); _cat_done: cat buffers so produced
        self.next_cat = None
        self._cat_done = set()
        self.keep_fp32 = False      # instance attribute: a training forward sets it on ITS engine's context only

    def begin_step(self):
        """Reset the rotation of the layer-output buffers: the eager warm-up step and the captured step then walk the
        SAME act{i} slots in the same order (capture never meets a slot the warm-up did not allocate)."""
        self._rot.clear()
        self._gn.clear()
        self._gn_done.clear()
        self._unwritten.clear()
        self._cat_done.clear()
        self.next_gn = None
        self.next_cat = None
        self._gn_next = 0
        if self.gn_from_producer:          # (one launch of the library's own fill kernel: the step contains no framework kernels)
            hip.check(hip.lib().mvd_fill_zero(hip.ptr(self.gn_arena), self.gn_arena.numel() * 2, hip.stream()))

    def gn_slot(self, out, B, HW, C):
        """Reserve the statistics slot of `out` (a (B*HW, C) tensor about to be produced) and remember it for ctx.groupnorm."""
        n = B * 32 * 2
        if n > GN_SLOT_ELEMS or self._gn_next >= GN_SLOTS:
            return None                 # more images / normalised tensors than the arena holds: the consumer runs the two-pass kernels
        st = self.gn_arena[self._gn_next * GN_SLOT_ELEMS:self._gn_next * GN_SLOT_ELEMS + n]
        self._gn_next += 1
        self._gn[out.data_ptr()] = (st, B, HW, C)
        return st

    def _forget(self, out):
        """`out` is about to be overwritten: its producer statistics, its 'fp32 not written' mark and every 'GroupNorm already applied'
        entry keyed on its address are stale (a stale (ptr, norm) match is impossible by construction then)."""
        ptr = out.data_ptr()
        self._gn.pop(ptr, None)
        self._unwritten.discard(ptr)
        if self._gn_done:
            for k in [k for k in self._gn_done if k[0] == ptr]:
                del self._gn_done[k]

    def row_stats(self, tag, rows, width):
        """Static per-row statistics slots for a (rows, width) tensor whose LayerNorm is folded into the consumer GEMM."""
        key = (tag, int(rows), int(width))
        rs = self._rs.get(key)
        if rs is None:
            assert not self.ws.frozen, f"row statistics {key} requested during graph capture"
            rs = hip.RowStats(rows, width, self.device)
            self._rs[key] = rs
        return rs

    def act(self, shape):
        """Rotating layer-output buffers (3 per shape): a layer's input stays valid while it writes its output."""
        key = tuple(int(s) for s in shape)
        i = self._rot.get(key, 0)
        self._rot[key] = (i + 1) % 3
        return self.ws.get(f"act{i}", key)

    # -- op helpers bound to this context
    def prec_of(self, kind):
        return self.policy.get(kind, self.prec)

    def ln_fold_enabled(self):
        """LayerNorm folded into the QKV / GEGLU GEMMs (hip.LnFold): the epilogue's `rstd (x.W' - mean colsum)` cancels exactly only as
        far as the MFMA operands carry W' and x -- 2^-22 with fp16 hi + lo, but 2^-12 / 2^-9 with one product and 2^-17 with bf16 hi + lo,
        amplified by |mean| / std of the row.  Those modes run the LayerNorm kernels instead."""
        return (self.ln_fold and hip.OPERAND_FORMAT == "f16" and self.prec_of("qkv") >= hip.PREC_X3 and
                self.prec_of("geglu") >= hip.PREC_X3 and self.prec_of("proj") >= hip.PREC_X3)

    def gemm(self, A, W, out, gn=None, kind=None, gn_apply=None, **kw):
        """gn=(B, HW): `out` feeds a GroupNorm over (B, HW, N) -- the GEMM emits its statistics (see gn_slot).
        gn_apply=(norm, planes, silu, out_unused): that GroupNorm (+ SiLU) is applied right behind the GEMM into `planes` -- inside the
        split-K reduce when the GEMM splits (mvd_gemm_desc.gna_out_sp), else by the apply kernel; out_unused: nothing else reads `out`.
        kind: the layer class of the precision policy (hip.PREC_KINDS)."""
        kw.setdefault("prec", self.prec_of(kind))
        kw.setdefault("workspace", self.gemm_ws)
        applied = False
        if out is not None:
            self._forget(out)                           # whatever statistics / normalised planes the buffer had are stale now
            handled = False
            if gn is not None and gn_apply is None and self.next_cat is not None and self.gn_fuse and self.gn_from_producer and \
                    gn[1] % 16 == 0 and out.is_contiguous():
                # the block output goes straight into torch.cat([h, skip]) -> GroupNorm -> SiLU of the next decoder block: this GEMM's
                # reduce (or one launch behind it) writes the normalised planes and the raw planes of the concatenation
                sk, cat, catp, norm, pname, silu = self.next_cat
                N, cb = out.shape[-1], sk.shape[-1]
                if norm.num_groups == 32 and norm.num_channels == N + cb and sk.shape[0] == out.shape[0] == gn[0] * gn[1] and \
                        sk.is_contiguous() and hip.lib().mvd_concat_groupnorm_fits(N, cb, gn[1], 32):
                    st = self.gn_slot(cat, gn[0], gn[1], N + cb)
                    if st is not None:
                        planes = self.ws.planes(pname, out.shape[0], N + cb)
                        kw.update(gn_stats=st, gn_hw=gn[1], gn_groups=32, cat=(sk, catp),
                                  gn_apply=(norm.weight, norm.bias, norm.eps, (hip.GNA_SILU if silu else 0) | hip.GNA_OUT_UNUSED, planes))
                        self._gn.pop(cat.data_ptr(), None)          # (the fp32 concatenation is not written: nothing to re-normalise)
                        self._gn_done[(cat.data_ptr(), id(norm))] = (planes, bool(silu))
                        self._unwritten.add(cat.data_ptr())
                        self._unwritten.add(out.data_ptr())
                        self._cat_done.add(cat.data_ptr())
                        handled = True
            if not handled and gn is not None and self.gn_from_producer and gn[1] % 16 == 0 and out.shape[-1] % 32 == 0:
                st = self.gn_slot(out, gn[0], gn[1], out.shape[-1])
                if st is not None:
                    kw.update(gn_stats=st, gn_hw=gn[1], gn_groups=32)
                    if gn_apply is not None and self.gn_fuse and gn_apply[0].num_groups == 32 and out.is_contiguous():
                        norm, planes, silu, unused = gn_apply
                        kw["gn_apply"] = (norm.weight, norm.bias, norm.eps, (hip.GNA_SILU if silu else 0) | (hip.GNA_OUT_UNUSED if unused else 0),
                                          planes)
                        applied = True
                    elif gn_apply is None and self.next_gn is not None and self.gn_fuse and out.is_contiguous():
                        # the layer's output GEMM: the next layer's GroupNorm rides along (ctx.groupnorm finds the planes in _gn_done)
                        norm, pname, silu = self.next_gn
                        if norm.num_groups == 32 and norm.num_channels == out.shape[-1] and out.shape[0] == gn[0] * gn[1]:
                            planes = self.ws.planes(pname, out.shape[0], out.shape[-1])
                            kw["gn_apply"] = (norm.weight, norm.bias, norm.eps, hip.GNA_SILU if silu else 0, planes)
                            self._gn_done[(out.data_ptr(), id(norm))] = (planes, bool(silu))
        if gn is not None:
            if os.environ.get("MVD_GN_DEBUG") and not self.capturing:
                print(f"[gn] kind={kind} M={out.shape[0]} N={out.shape[-1]} explicit={gn_apply is not None} hint={self.next_gn is not None} "
                      f"cat={self.next_cat is not None} -> fused_apply={'gn_apply' in kw} cat={'cat' in kw}", flush=True)
            if gn_apply is None:        # (the hints are for the layer's OUTPUT GEMM; a ResBlock's conv1 -- explicit gn_apply -- leaves them)
                self.next_gn = None
                self.next_cat = None
        r = hip.gemm(A, W, out, **kw)
        if gn_apply is not None and not applied:
            self.groupnorm(out, gn_apply[1], gn_apply[0], gn[0], gn[1], out.shape[-1], gn_apply[2])
        return r

    def groupnorm(self, x, y, norm, B, HW, C, silu):
        done = self._gn_done.pop((x.data_ptr(), id(norm)), None)
        if done is not None and done[0].data_ptr() == y.data_ptr() and done[0].shape == y.shape and done[1] == bool(silu) and silu in (True, False):
            return y                    # applied behind the GEMM that produced x
        if x.data_ptr() in self._unwritten:
            raise RuntimeError("GroupNorm of a tensor whose fused producer left the fp32 data unwritten (Ctx.concat(need_out=False))")
        ent = self._gn.get(x.data_ptr())
        if ent is not None and ent[1:] == (B, HW, C) and norm.num_groups == 32:
            return hip.groupnorm_from_stats(x, y, norm.weight, norm.bias, ent[0], B, HW, C, norm.eps, silu)
        # partial-sum workspace sized for THIS call (B * chunks(HW) * groups * 2 doubles); the ABI checks the size
        ws = self.ws.get("gn_ws", (B * hip.lib().mvd_groupnorm_chunks(HW) * 32 * 2,), torch.float64)
        return hip.groupnorm(x, y, norm.weight, norm.bias, B, HW, C, norm.eps, silu, ws)

    def concat(self, a, ca, b, cb, out, out_planes, B, HW, gn_apply=None, need_out=True):
        """out = [a | b] along the channels (+ its split planes), with the GroupNorm statistics of the result.
        gn_apply = (norm, planes buffer name, silu): the GroupNorm that consumes the result is applied in the same launch
        (mvd_concat_groupnorm) when a group fits the LDS; need_out=False: nobody reads the fp32 concatenation itself."""
        self._forget(out)
        st = None
        if gn_apply is not None and self.gn_fuse and self.gn_from_producer and HW % 16 == 0 and (ca + cb) % 32 == 0 and \
                gn_apply[0].num_groups == 32 and gn_apply[0].num_channels == ca + cb and \
                hip.lib().mvd_concat_groupnorm_fits(ca, cb, HW, 32):
            norm, pname, silu = gn_apply
            st = self.gn_slot(out, B, HW, ca + cb)
            planes = self.ws.planes(pname, B * HW, ca + cb)
            hip.check(hip.lib().mvd_concat_groupnorm(hip.ptr(a), ca, hip.ptr(b), cb, hip.ptr(out) if need_out else None, hip.ptr(out_planes),
                                                     hip.ptr(planes), hip.ptr(norm.weight), hip.ptr(norm.bias),
                                                     hip.ptr(st) if st is not None else None, B, HW, 32, norm.eps, 1 if silu else 0,
                                                     hip.stream()))
            if not need_out:
                self._gn.pop(out.data_ptr(), None)      # (no fp32 data behind the statistics slot: nobody may normalise `out` again)
                self._unwritten.add(out.data_ptr())
            self._gn_done[(out.data_ptr(), id(norm))] = (planes, bool(silu))
            return out
        if self.gn_from_producer and HW % 16 == 0 and (ca + cb) % 32 == 0:
            st = self.gn_slot(out, B, HW, ca + cb)
        hip.check(hip.lib().mvd_concat_channels(hip.ptr(a), ca, hip.ptr(b), cb, hip.ptr(out), hip.ptr(out_planes), B * HW,
                                                hip.ptr(st) if st is not None else None, HW, 32, hip.stream()))
        return out

    def layernorm(self, x, y, norm, rows, C):
        return hip.layernorm(x, y, norm.weight, norm.bias, rows, C, norm.eps)

    def gemv_rows(self, W, bias, x, y, act_in=hip.ACT_NONE, act_out=hip.ACT_NONE):
        """gemv for any number of rows (the kernel takes <= 16 at a time)."""
        for r in range(0, x.shape[0], 16):
            hip.gemv(W, bias, x[r:r + 16], y[r:r + 16], act_in, act_out)
        return y


def ddim_step_table(scheduler_tables, ddim, iters):
    """(len(iters), 8) fp32 table for the device kernels (include/mvd_hip.h, MVD_STEP_STRIDE).

    ``iters`` is the list of DDIM indices in execution order (49, 48, ..., 0 for a full sample).
    """
    rows = []
    for index in iters:
        t = int(ddim["timesteps"][index])
        sac = float(scheduler_tables["sqrt_alphas_cumprod"][t])
        s1m = float(scheduler_tables["sqrt_one_minus_alphas_cumprod"][t])
        dstd = (scheduler_tables["sqrt_one_minus_alphas_cumprod"][t] / scheduler_tables["sqrt_alphas_cumprod"][t] / 10.0)
        rows.append([float(t), sac, float(dstd), float(ddim["alphas"][index]), float(ddim["alphas_prev"][index]),
                     float(ddim["sigmas"][index]), float(ddim["sqrt_one_minus_alphas"][index]),
                     1.0 if index > 0 else 0.0])
        del s1m
    return torch.tensor(rows, dtype=torch.float32)
