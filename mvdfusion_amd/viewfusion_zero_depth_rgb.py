"""ViewFusion facade on the MI355X-native hot path -- mirror of
``mvdfusion.viewfusion_zero_depth_rgb.ViewFusion`` (mvdfusion/viewfusion_zero_depth_rgb.py:19-417).

Drop-in surface kept: constructor kwargs (unknown keys ignored, :41), ``apply_model`` (:282), ``sample`` (:348),
``embed_time`` (:276), ``prepare_batch`` (:165), ``encode``/``decode`` (:157-163), ``forward``/``p_losses`` (:362-397
-- the training loss with a hand-written HIP backward behind torch autograd), ``configure_optimizers`` (:399), ``_print_parameter_count`` (:134),
``.ddim`` (DDIMSampler), ``.scheduler``; state_dict keys ``view_attn.*``, ``unet_model.unet_model.*``,
``cc_projection.{0,2,4}``, ``time_embed.{0,2}``, ``scheduler.*`` (+ ``vae.*`` / ``clip_image_encoder.*`` when those
host-side PyTorch modules are plugged in -- they are outside the hot path).

The per-step work is done by :class:`StepEngine`: static buffers + device-resident step tables, one hipGraph per
(V, S, D, cfg) signature, replayed once per DDIM step.
"""
import math

import torch
import torch.nn as nn

from . import hip
from .cameras import Cameras, get_camera_slice, get_relative_camera, pack_cameras
from .engine import Ctx, ddim_step_table
from .sampler import DDIMSampler
from .scheduler import DDPMScheduler
from .unet import UNetWrapper
from .view_attn_efficient2 import GridAttn

import os

# Weight prefetch inside the graph-replayed step (include/mvd_hip.h: mvd_gemm_desc.pf_items).  MVD_PREFETCH=0 turns it
# off (A/B runs); MVD_PREFETCH_OPTS="window=100663296,max_items=24" overrides hip.WeightPrefetcher's parameters.
PREFETCH_WEIGHTS = os.environ.get("MVD_PREFETCH", "ws")          # "ws" (in-kernel, default) | "0"
PREFETCH_OPTIONS = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("MVD_PREFETCH_OPTS", "").split(",") if kv)}


def _sinusoid_freqs(dim, max_period=10000):
    """exp(-ln(max_period) * i / half), computed on the host exactly as the reference does
    (diffusionmodules/util.py:161-164; mvdfusion/embedder.py:124-127)."""
    half = dim // 2
    return torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)


class StepEngine:
    """One denoising iteration (GridAttn -> CFG-batched UNet -> CFG combine [+ DDIM update]) on static buffers."""

    def __init__(self, model, V, S, D, cfg, device, prec, q0=0, Vq=None, policy=None):
        self.m, self.V, self.S, self.D, self.cfg = model, V, S, D, bool(cfg)
        self.q0, self.Vq = q0, (V if Vq is None else Vq)   # query views owned by this rank (view-parallel sharding)
        self.ctx = Ctx(device, prec, policy)
        dev = self.ctx.device
        B = 2 * self.Vq if cfg else self.Vq
        self.B = B
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        self.x = z(V, 5, S, S)                 # current latents (NCHW like the reference), updated in place
        self.x0 = z(V, 5, S, S)
        self.eps = z(V, 5, S, S)
        self.input_latents = z(1, 5, S, S)
        self.clip_v_embed = z(V, 796)
        self.cams = z(V, hip.CAM_RECORD)
        self.in_cam = z(1, hip.CAM_RECORD)
        self.context = z(B, 768)               # rows [V,2V) stay zero: the null branch (unet.py:173)
        self.vol = z(B * S * S * D, 768)       # rows of the null branch stay zero (unet.py:190)
        # the same as split planes, inside the level-0 operand buffer of the UNet's view-aligned transformers (zero-initialised)
        self.vol_planes, self.vol_col = model.unet_model.level0_operand(self.ctx, B, S, D)
        self.x_in = torch.zeros(B * S * S, 2 * 32, dtype=torch.int16, device=dev)              # UNet input (split planes)
        self.iter = torch.zeros(1, dtype=torch.int32, device=dev)
        self.steps = z(1, hip.STEP_STRIDE)
        self.steps_nodiv = z(1, hip.STEP_STRIDE)   # the same table with sqrt(alpha_bar) = 1: GridAttn's depth source is used as is
        self.prev = z(V, 5, S, S)              # channel 4: an explicit overwrite_attn_depth (apply_model(prev_depth=...))
        self.depth_mode = 0                    # GridAttn samples depth around 0: x[:,4] / sqrt(alpha_bar) (the x0-style estimate);
                                               # 1: the previous step's x0 estimate self.x0[:,4] (DDIMSampler feed_prev_depth);
                                               # 2: self.prev[:,4] (view_attn_efficient2.py:418-426)
        self.depth_noise = z(1, V, D, S, S)
        self.ddim_noise = z(1, V, 5, S, S)
        self.f256 = _sinusoid_freqs(256).to(dev)
        self.funet = _sinusoid_freqs(model.unet_model.unet_model.model_channels).to(dev)
        self.graphs = {}
        self.prefetchers = {}      # per captured graph: its hip.WeightPrefetcher (None when PREFETCH_WEIGHTS is off)
        self.drop_masks = None     # training forward: (clip_mask, volume_mask, concat_mask), each (Vq,) in {0, 1} (unet.py:140-151)
        self.n_rows = 1            # rows of the device step table; the kernels index steps[iter], noise[iter] unchecked
        self.done = 0              # host mirror of the device iteration counter

    # -- host-side inputs ----------------------------------------------------------------------------
    def set_conditioning(self, batch_cameras, input_latents, input_cameras, clip_v_embed):
        self.cams.copy_(pack_cameras(batch_cameras).to(self.cams.device))
        self.in_cam.copy_(pack_cameras(input_cameras).to(self.cams.device))
        self.input_latents.copy_(input_latents.reshape(1, 5, self.S, self.S))
        self.clip_v_embed.copy_(clip_v_embed.reshape(self.V, 796))

    def set_schedule(self, steps_table, depth_noise, ddim_noise):
        dev = self.ctx.device
        if self.steps.shape != steps_table.shape:
            self.graphs.clear()                # table buffers are re-allocated: captured pointers go stale
            self.prefetchers.clear()
            self.steps = steps_table.to(dev).contiguous()
            self.steps_nodiv = self.steps.clone()
            self.depth_noise = depth_noise.to(dev).contiguous()
            self.ddim_noise = ddim_noise.to(dev).contiguous()
        else:
            self.steps.copy_(steps_table)
            self.depth_noise.copy_(depth_noise)
            self.ddim_noise.copy_(ddim_noise)
        self.steps_nodiv.copy_(self.steps)
        self.steps_nodiv[:, 1] = 1.0
        self.n_rows = int(steps_table.shape[0])
        self.rewind()

    def rewind(self, it=0):
        """Reset the device iteration counter (and its host mirror) to row `it` of the step table."""
        assert 0 <= it < self.n_rows
        self.iter.fill_(it)
        self.done = it

    def depth_geo(self):
        """(depth source (V,5,S,S), step table) GridAttn's kernels read the depth channel and its std from (see depth_mode)."""
        if self.depth_mode == 0:
            return self.x, self.steps
        return (self.x0 if self.depth_mode == 1 else self.prev), self.steps_nodiv

    # -- one iteration -------------------------------------------------------------------------------
    def enqueue(self, cfg_scale, do_update, prefetcher=None):
        m, ctx, L = self.m, self.ctx, hip.lib()
        V, S, D, B, q0, Vq = self.V, self.S, self.D, self.B, self.q0, self.Vq
        ctx.B, ctx.D = B, D
        ctx.begin_step()           # eager warm-up and graph capture walk the same rotating buffers
        st = hip.stream
        # embed_time (:276-279): sinusoid(256) -> Linear -> SiLU -> Linear; only row 0 is used downstream (t[:1])
        ts = ctx.ws.get("vf.tsin", (1, 256))
        hip.check(L.mvd_timestep_embedding(hip.ptr(self.steps), hip.ptr(self.iter), hip.ptr(self.f256), hip.ptr(ts), 256, st()))
        te1 = ctx.ws.get("vf.te1", (1, 256))
        hip.gemv(m.time_embed[0].weight, m.time_embed[0].bias, ts, te1, act_out=hip.ACT_SILU)
        c = ctx.ws.get("vf.c", (1, 256))
        hip.gemv(m.time_embed[2].weight, m.time_embed[2].bias, te1, c)
        # view-aligned features (:303-313)
        dsrc, dsteps = self.depth_geo()
        m.view_attn.run(ctx, self.x, self.depth_noise, self.steps, self.iter, self.cams, self.in_cam,
                        self.input_latents, c, self.vol, V, S, D, q0=q0, Vq=Vq, vol_planes=self.vol_planes,
                        vol_planes_col=self.vol_col, depth_src=None if self.depth_mode == 0 else dsrc,
                        depth_steps=None if self.depth_mode == 0 else dsteps)
        # cc_projection (:322)
        p = m.cc_projection
        c1 = ctx.ws.get("vf.cc1", (Vq, 768))
        c2 = ctx.ws.get("vf.cc2", (Vq, 768))
        ctx.gemv_rows(p[0].weight, p[0].bias, self.clip_v_embed[q0:q0 + Vq], c1, act_out=hip.ACT_SILU)
        ctx.gemv_rows(p[2].weight, p[2].bias, c1, c2, act_out=hip.ACT_SILU)
        ctx.gemv_rows(p[4].weight, p[4].bias, c2, self.context[:Vq])
        if self.drop_masks is not None:        # UNetWrapper.forward(is_train=True) condition dropout (eager only, never captured)
            clip_m, vol_m, cat_m = self.drop_masks
            self.context[:Vq] *= clip_m[:, None]
            self.vol.view(B, -1)[:Vq] *= vol_m[:, None]
            vp = self.vol_planes.view(B, S * S * D, -1)
            vp[:Vq, :, 2 * self.vol_col:] *= vol_m.to(torch.int16)[:, None, None]        # x * {0, 1} keeps / zeroes the planes
        ctx.context = self.context
        # UNet on the CFG batch (unet.py:167-196)
        xq, x0q, epsq = self.x[q0:q0 + Vq], self.x0[q0:q0 + Vq], self.eps[q0:q0 + Vq]
        hip.check(L.mvd_unet_input(hip.ptr(xq), hip.ptr(self.input_latents), hip.ptr(self.x_in), Vq, S, 32,
                                   int(self.cfg), st()))
        if self.drop_masks is not None:        # x_concat channels 5..9 of the (rows, [32 hi | 32 lo]) input planes
            xin = self.x_in.view(B, S * S, 64)
            cm = self.drop_masks[2].to(torch.int16)[:, None, None]
            xin[:Vq, :, 5:10] *= cm
            xin[:Vq, :, 37:42] *= cm
        unet = m.unet_model.unet_model
        ctx.vol_levels = m.unet_model.volume_pyramid(ctx, self.vol.view(B, S, S, D, 768), B, S, D)
        tsu = ctx.ws.get("vf.tsin_unet", (1, unet.model_channels))
        hip.check(L.mvd_timestep_embedding(hip.ptr(self.steps), hip.ptr(self.iter), hip.ptr(self.funet), hip.ptr(tsu),
                                           unet.model_channels, st()))
        y = unet.run(ctx, self.x_in, tsu, S)
        hip.check(L.mvd_cfg_ddim_update(hip.ptr(y), 8, hip.ptr(xq), hip.ptr(x0q), hip.ptr(epsq),
                                        hip.ptr(self.ddim_noise[:, q0:q0 + Vq]), V * 5 * S * S, hip.ptr(self.steps),
                                        hip.ptr(self.iter), Vq, S, int(self.cfg), float(cfg_scale), int(do_update), st()))
        if do_update:
            hip.check(L.mvd_advance_iter(hip.ptr(self.iter), st()))

    def step(self, cfg_scale, do_update, use_graph=True):
        key = (float(cfg_scale), bool(do_update), int(self.depth_mode))
        if self.done >= self.n_rows:
            raise IndexError(f"StepEngine.step: iteration {self.done} is past the {self.n_rows}-row step table "
                             "(set_schedule() / rewind() before stepping again)")
        if do_update:
            self.done += 1
        if not use_graph:
            return self.enqueue(cfg_scale, do_update)
        g = self.graphs.get(key)
        if g is None:
            # first call runs eagerly (allocates every workspace buffer, packs weights), then capture
            it0 = self.iter.clone()
            x_keep, x0_keep = self.x.clone(), self.x0.clone()      # (x0: the previous step's estimate is an INPUT under feed_prev_depth)
            hip.AUTOTUNE = True            # pick the GEMM kernel configuration per problem shape (cached)
            try:
                self.enqueue(cfg_scale, do_update)
            finally:
                hip.AUTOTUNE = False
                hip.release_tuning_buffers()
            torch.cuda.synchronize()
            # weight prefetch (include/mvd_hip.h: mvd_gemm_desc.pf_items): a second eager pass with the tuned configurations records the
            # step's GEMM launch order; at capture every role-split launch is handed the weights of the launches behind it
            pf = None
            if PREFETCH_WEIGHTS == "ws":
                pf = hip.WeightPrefetcher(self.x.device, **PREFETCH_OPTIONS)
                self.iter.copy_(it0)
                self.x.copy_(x_keep)
                self.x0.copy_(x0_keep)
                with pf.following(record=True):
                    self.enqueue(cfg_scale, do_update)
                torch.cuda.synchronize()
            self.iter.copy_(it0)
            self.x.copy_(x_keep)
            self.x0.copy_(x0_keep)
            g = hip.Graph()
            self.ctx.ws.frozen = True      # capture must not allocate: every buffer exists after the eager step
            try:
                with g:
                    if pf is None:
                        self.enqueue(cfg_scale, do_update)
                    else:
                        with pf.following():
                            self.enqueue(cfg_scale, do_update, prefetcher=pf)
            finally:
                self.ctx.ws.frozen = False
            self.graphs[key] = g
            self.prefetchers[key] = pf
        g.launch()


class _HipTrainingLoss(torch.autograd.Function):
    """Bridges ViewFusion.gradients (forward + hand-written backward kernels) into torch autograd: forward computes the loss AND the
    parameter gradients (the HIP buffers of the step are only valid until the next step), backward scales and returns them."""

    @staticmethod
    def forward(ctx, model, batch, trainer_config, names, *params):
        loss, grads = model.gradients(batch, trainer_config, noise_source=getattr(model, "_noise_source", None), only_trainable=True)
        ctx.grads = [grads.get(n) for n in names]
        ctx.shapes = [p.shape for p in params]
        return loss.clone()

    @staticmethod
    def backward(ctx, grad_out):
        # the gradients are this step's own buffers: scale them in place with multi-tensor launches (994 separate multiplies were 15 ms).
        #  in place is only safe ONCE and only over disjoint buffers: a second backward over the same node (retain_graph=True)
        # must not scale again, and two gradients that overlap without being the same view take the out-of-place path.
        if getattr(ctx, "_scaled", False):
            raise RuntimeError("_HipTrainingLoss.backward ran twice on one node: the step's gradient buffers were already scaled in place "
                               "(call model(batch, cfg) again instead of retain_graph=True)")
        ctx._scaled = True
        live, seen, spans = [], set(), []
        for g in ctx.grads:
            if g is None:
                continue
            key = (g.untyped_storage().data_ptr(), g.storage_offset(), tuple(g.shape), tuple(g.stride()))
            if key in seen:                                      # (a buffer shared by two names is scaled once)
                continue
            seen.add(key)
            live.append(g)
            spans.append((g.data_ptr(), g.data_ptr() + g.numel() * g.element_size()) if g.is_contiguous() else None)
        disjoint = all(sp is not None for sp in spans)
        if disjoint:
            srt = sorted(spans)
            disjoint = all(srt[i][1] <= srt[i + 1][0] for i in range(len(srt) - 1))
        if live and disjoint:
            torch._foreach_mul_(live, grad_out.to(live[0].dtype))
            pick = lambda g: g
        else:                                                     # overlapping / strided views: the safe out-of-place multiply per name
            pick = lambda g: g * grad_out.to(g.dtype)
        outs = [torch.zeros(s, dtype=grad_out.dtype, device=grad_out.device) if g is None else pick(g).reshape(s)
                for g, s in zip(ctx.grads, ctx.shapes)]
        return (None, None, None, None, *outs)


class ViewFusion(nn.Module):
    def __init__(self, view_attn_config, unet_config, ddpm_config, vae_config=None, unet_path="", vae_path="",
                 clip_path="", unet_cc_path="", z_scale_factor=0.18215, vae_max_batch=8, objective="noise",
                 loss_type="l2", embed_camera_pose=True, finetune_projection=False, finetune_unet=False,
                 finetune_cross_attn=True, finetune_view_attn=True, feed_prev_depth=False, drop_conditions=False,
                 vae=None, clip_image_encoder=None, precision=None, reference_eval_dropout=False, **kwargs):
        super().__init__()
        assert embed_camera_pose, "this build implements the embed_camera_pose=True configuration of configs/*.yaml"
        self.finetune_projection, self.finetune_unet, self.z_scale_factor = finetune_projection, finetune_unet, z_scale_factor
        self.vae_max_batch, self.objective, self.loss_type = vae_max_batch, objective, loss_type
        self.embed_camera_pose, self.finetune_cross_attn, self.finetune_view_attn = \
            embed_camera_pose, finetune_cross_attn, finetune_view_attn
        self.feed_prev_depth, self.drop_conditions = feed_prev_depth, drop_conditions
        # The reference applies the condition dropout whenever cfg_scale == 1 (is_train=True is hard-wired at :322-330 / unet.py:140),
        # i.e. also under model.eval() (validation losses, cfg = 1 sampling).  Default here: only in training mode; True reproduces the
        # reference's eval-mode behaviour bit for bit in distribution (validation curves comparable 1:1).
        self.reference_eval_dropout = bool(reference_eval_dropout)
        # precision = MFMA operand type x number of partial products of the (hi+lo)(hi+lo) operand split:
        # "f16x3" (default, configs.DEFAULT_PRECISION: fp16 hi + lo, lo*lo dropped, ~2^-22), "f16x4" (all 4 products; +5 - 7 % time, no
        # measurable accuracy difference: DESIGN.md section 4), "bf16x3" (~2^-16), "f16" / "bf16" (one product, hi only).  The operand type selects the library
        # flavour and is fixed per process.
        # A policy string "f16x4:conv=3,geglu=3" sets the products per layer class (hip.PREC_KINDS).
        from .configs import DEFAULT_PRECISION
        precision = precision or DEFAULT_PRECISION
        fmt, self.precision, self.precision_policy = hip.parse_precision(precision)
        hip.set_operand_format(fmt)
        self.precision_name = precision

        def params(cfg):
            return dict(cfg.get("params", cfg)) if hasattr(cfg, "get") else dict(cfg)

        self.view_attn = GridAttn(**params(view_attn_config))
        self.unet_model = UNetWrapper(unet_config, unet_path=unet_path or None, drop_conditions=drop_conditions,
                                      drop_scheme="default", finetune_unet=finetune_unet,
                                      finetune_cross_attn=finetune_cross_attn, finetune_view_attn=finetune_view_attn,
                                      use_zero_123=True,
                                      remove_keys=["input_blocks.0.0.weight", "out.2.weight", "out.2.bias"])
        self.scheduler = DDPMScheduler(**params(ddpm_config))
        # VAE: `vae_config` (configs/*.yaml: external.sd1.ldm.models.autoencoder.AutoencoderKL) builds the HIP-backed decode
        # mirror (mvdfusion_amd/autoencoder.py; its encode needs an injected module); an injected `vae` takes precedence.
        if vae is not None:
            self.vae = vae
        elif vae_config is not None:
            from .load_model import instantiate_from_config
            self.vae = instantiate_from_config(vae_config)
            if vae_path:
                sd = torch.load(vae_path, map_location="cpu")
                sd = sd.get("state_dict", sd)
                self.vae.load_state_dict({k.replace("first_stage_model.", ""): v for k, v in sd.items()}, strict=False)
        # CLIP image encoder (viewfusion_zero_depth_rgb.py:103-105: FrozenCLIPImageEmbedder(model=clip_path), frozen): an injected
        # module takes precedence; otherwise `clip_path` (the CLIP model name of configs/*.yaml, e.g. "ViT-L/14") builds the
        # HIP-backed mirror, whose weights come with the checkpoint's clip_image_encoder.* keys.
        if clip_image_encoder is not None:
            self.clip_image_encoder = clip_image_encoder
        elif clip_path:
            from .encoders import FrozenCLIPImageEmbedder
            self.clip_image_encoder = FrozenCLIPImageEmbedder(model=clip_path, precision=precision)
            for prm in self.clip_image_encoder.parameters():
                prm.requires_grad_(False)
        self.cc_projection = nn.Sequential(nn.Linear(768 + 14 * 2, 768), nn.SiLU(True), nn.Linear(768, 768),
                                           nn.SiLU(True), nn.Linear(768, 768))
        nn.init.eye_(list(self.cc_projection.parameters())[0][:768, :768])
        nn.init.zeros_(list(self.cc_projection.parameters())[1])
        self.time_embed_dim = 256
        self.time_embed = nn.Sequential(nn.Linear(256, 256), nn.SiLU(True), nn.Linear(256, 256))
        self.register_buffer("_device", torch.tensor([0.0]), persistent=False)
        self.latent_size = int(self.view_attn.input_size)
        self.ddim = DDIMSampler(self, ddim_num_steps=50, ddim_discretize="uniform", ddim_eta=1.0,
                                latent_size=self.latent_size, z_dim=4, feed_prev_depth=feed_prev_depth)
        self._engines = {}
        self._packed_sig = None
        assert self.finetune_view_attn is True, "must finetune new view attention layers"

    # ------------------------------------------------------------------------------------------------
    train_autotune_min_flops = 2.0e9      # training step: GEMMs at least this large are autotuned on first sight (None: heuristic tiles only)

    def invalidate_packed(self, keep_engines=False):
        """Drop every packed weight image, captured graph and engine: they are rebuilt from the live fp32 parameters.
        keep_engines: the parameters changed IN PLACE (an optimizer step): the engines' static arenas (activations, split-K slabs,
        statistics slots -- GBs of buffers) stay, only the packed images and the graphs that captured their addresses go."""
        hip.drop_packed_caches(self)
        hip.forget_param_maxima()          # (engine() measures them again before anything is packed for a step)
        if keep_engines:
            for e in self._engines.values():
                e.graphs.clear()
                e.prefetchers.clear()
        else:
            self._engines.clear()

    def load_state_dict(self, state_dict, strict=True, **kw):
        """nn.Module.load_state_dict, then drop every derived packed weight / engine / graph (demo.py:165, train.py:144-153)."""
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self.invalidate_packed()
        self._packed_sig = None
        return r

    def _apply(self, fn, *a, **kw):
        """.cuda() / .to() / .float(): parameters move, so every derived packed image is stale."""
        r = super()._apply(fn, *a, **kw)
        if "_engines" in self.__dict__:
            self.invalidate_packed()
            self._packed_sig = None
        return r

    def engine(self, V, S, D, cfg, q0=0, Vq=None):
        sig = hip.params_signature(self)
        if sig != self._packed_sig:      # an in-place update (optimizer step, fill_) since the weights were packed; load_state_dict /
            if self._packed_sig is not None:      # .cuda() invalidate fully themselves (the device may have changed)
                dev_now = self._device.device
                same_dev = all(e.ctx.device == dev_now for e in self._engines.values())
                self.invalidate_packed(keep_engines=same_dev)
            self._packed_sig = sig
            hip.register_param_maxima(self.parameters())      # one reduction + one host read instead of one per packed weight
        if Vq is not None and Vq <= 0:
            raise ValueError(f"engine(V={V}, q0={q0}, Vq={Vq}): a rank must own at least one query view")
        key = (V, S, D, bool(cfg), q0, Vq)
        e = self._engines.get(key)
        if e is None:
            dev = self._device.device
            if dev.type != "cuda":
                raise RuntimeError("mvdfusion_amd.ViewFusion runs on the GPU only (model.cuda() first); "
                                   "there is no CPU path in the product")
            e = StepEngine(self, V, S, D, cfg, dev, self.precision, q0=q0, Vq=Vq, policy=self.precision_policy)
            self._engines[key] = e
        return e

    def _print_parameter_count(self):
        va = sum(p.numel() for p in self.view_attn.parameters())
        un = sum(p.numel() for p in self.unet_model.get_trainable_parameters())
        uf = sum(p.numel() for p in self.unet_model.parameters())
        tp = sum(p.numel() for p in self.time_embed.parameters())
        pp = sum(p.numel() for p in self.cc_projection.parameters()) if self.finetune_projection else 0
        print(f"view_attn {va * 1e-6:.2f}M | unet trainable {un * 1e-6:.2f}M / full {uf * 1e-6:.2f}M | "
              f"total trainable {(va + un + tp + pp) * 1e-6:.2f}M / full {(va + uf + tp + pp) * 1e-6:.2f}M")

    @torch.no_grad()
    def encode_clip(self, x):
        return self.clip_image_encoder.encode(x)

    @torch.no_grad()
    def encode(self, x):
        return self.vae.encode(torch.clip(x * 2 - 1.0, -1.0, 1.0)).mode() * self.z_scale_factor

    @torch.no_grad()
    def decode(self, z):
        return torch.clip((self.vae.decode(z * 1 / self.z_scale_factor) + 1.0) / 2.0, 0.0, 1.0).clip(0.0, 1.0)

    def prepare_batch(self, batch, trainer_config, generator=None):
        """viewfusion_zero_depth_rgb.py:165-273 (host-side; needs the plugged-in VAE / CLIP encoders)."""
        images = batch["images"]
        dev = images.device
        Bn, _, H, W = images.shape
        n_in, n_tr = trainer_config["input_batch_size"], trainer_config["train_batch_size"]
        if trainer_config["random_views"]:
            rand = torch.randperm(Bn, generator=generator) if generator is not None else torch.randperm(Bn)
        else:
            rand = torch.linspace(0, Bn - 1, n_in + n_tr).long()
        in_idx, b_idx = rand[:n_in], rand[n_in:n_in + n_tr]
        input_latents = self.encode(images[in_idx])
        batch_latents = self.encode(images[b_idx])
        area = lambda d: torch.nn.functional.interpolate(d, scale_factor=0.125, mode="area")
        in_depth = torch.zeros_like(area(torch.zeros((n_in, 1, H, W), device=dev)))      # input depth forced to 0 (:215)
        input_latents = torch.cat((input_latents, in_depth), dim=1)
        if "depths" in batch:
            b_depth = torch.clip(batch["depths"][b_idx].to(dev) * 2 - 1.0, -1.0, 1.0)
        else:
            b_depth = torch.zeros((n_tr, 1, H, W), device=dev)
        batch_latents = torch.cat((batch_latents, area(b_depth)), dim=1)
        cams = get_relative_camera(Cameras(batch["R"].float().cpu(), batch["T"].float().cpu(),
                                           batch["f"].float().cpu(), batch["c"].float().cpu()), in_idx)
        input_cameras, batch_cameras = get_camera_slice(cams, in_idx), get_camera_slice(cams, b_idx)
        clip_embed = self.encode_clip(images[in_idx]).expand(n_tr, -1, -1)
        from .synthetic import cam_embed
        clip_v_embed = torch.cat((clip_embed, cam_embed(input_cameras, batch_cameras).to(dev)), dim=-1)
        return batch_latents, batch_cameras, input_latents, input_cameras, clip_v_embed

    def embed_time(self, t):
        """(B,) int timesteps -> (B, 256).  Runs the same GEMV kernels as the step engine."""
        dev = self._device.device
        e = torch.cat([torch.cos(t[:, None].float() * self._f256(dev)), torch.sin(t[:, None].float() * self._f256(dev))], -1)
        h = torch.empty(t.shape[0], 256, device=dev)
        out = torch.empty(t.shape[0], 256, device=dev)
        for r in range(0, t.shape[0], 16):
            hip.gemv(self.time_embed[0].weight, self.time_embed[0].bias, e[r:r + 16].contiguous(), h[r:r + 16], act_out=hip.ACT_SILU)
            hip.gemv(self.time_embed[2].weight, self.time_embed[2].bias, h[r:r + 16], out[r:r + 16])
        return out

    def _f256(self, dev):
        if not hasattr(self, "_f256_cache") or self._f256_cache.device != dev:
            self._f256_cache = _sinusoid_freqs(256).to(dev)
        return self._f256_cache[None]

    @torch.no_grad()
    def apply_model(self, noisy_latents, batch_cameras, input_latents, input_cameras, clip_v_embed, t, prev_depth=None,
                    cfg_scale=1.0, depth_noise=None, drop_rand=None):
        """viewfusion_zero_depth_rgb.py:282-345.  ``depth_noise`` (V,D,S,S) optionally injects the N(0,1) draw that
        the reference takes inside GridAttn (view_attn_efficient2.py:431); default: torch's device generator.
        With cfg_scale == 1 the reference calls UNetWrapper.forward(is_train=True): when the model was built with
        drop_conditions=True and is in training mode, the per-view condition dropout of unet.py:109-151 is applied
        (``drop_rand`` (V,) optionally injects its torch.rand draw)."""
        V, _, S, _ = noisy_latents.shape
        D = self.view_attn.n_pts_per_ray
        cfg = cfg_scale != 1.0
        eng = self.engine(V, S, D, cfg)
        eng.set_conditioning(batch_cameras, input_latents, input_cameras, clip_v_embed)
        tv = int(t[0])
        sac = self.scheduler.sqrt_alphas_cumprod[tv]
        dstd = self.scheduler.sqrt_one_minus_alphas_cumprod[tv] / sac / 10.0
        table = torch.tensor([[float(tv), float(sac), float(dstd), 1.0, 1.0, 0.0, 0.0, 0.0]], dtype=torch.float32)
        if depth_noise is None:
            depth_noise = torch.randn(V, D, S, S, device=noisy_latents.device)
        eng.set_schedule(table, depth_noise.reshape(1, V, D, S, S), torch.zeros(1, V, 5, S, S))
        eng.x.copy_(noisy_latents)
        eng.depth_mode = 0
        if prev_depth is not None:             # overwrite_attn_depth (:312; view_attn_efficient2.py:418-426): (V or 1, 1, S, S)
            eng.prev[:, 4:5].copy_(prev_depth.to(eng.prev.device).expand(V, 1, S, S))
            eng.depth_mode = 2
        eng.drop_masks = None
        if not cfg and self.drop_conditions and (self.training or self.reference_eval_dropout):
            r = torch.rand(V, device=noisy_latents.device) if drop_rand is None else drop_rand.to(noisy_latents.device).float()
            drop_clip, drop_vol = (r > 0.15) & (r <= 0.2), (r > 0.1) & (r <= 0.15)          # get_drop_scheme 'default' (unet.py:109-117)
            drop_cat, drop_all = (r > 0.05) & (r <= 0.1), r <= 0.05
            eng.drop_masks = tuple(1.0 - (dm | drop_all).float() for dm in (drop_clip, drop_vol, drop_cat))
        self._last_drop_masks = eng.drop_masks
        keep = eng.ctx.keep_fp32
        eng.ctx.keep_fp32 = keep or bool(getattr(self, "_train_keep_fp32", False))      # this engine only (never a class-wide switch)
        try:
            eng.step(cfg_scale, do_update=False, use_graph=eng.drop_masks is None and not getattr(self, "_force_eager", False))
        finally:
            eng.ctx.keep_fp32 = keep
            eng.drop_masks = None      # (depth_mode stays: the training backward re-derives the geometry from the same depth source)
        return hip.check_finite(eng.eps.clone(), "ViewFusion.apply_model")

    def sample(self, batch, trainer_config, cfg_scale, return_input=False, depth=False, verbose=True):
        """viewfusion_zero_depth_rgb.py:348-359."""
        batch_latents, batch_cameras, input_latents, input_cameras, clip_v_embed = self.prepare_batch(batch, trainer_config)
        res = self.ddim.sample(batch_cameras, input_latents, input_cameras, clip_v_embed, unconditional_scale=cfg_scale,
                               depth=depth, return_intermediates=return_input, verbose=verbose)
        if return_input:
            x_sample, intermediates = res
            return x_sample, batch_latents, input_latents, batch_cameras, intermediates
        return res

    @torch.no_grad()
    def p_losses(self, batch, trainer_config, noise_source=None, _aux=None):
        """viewfusion_zero_depth_rgb.py:362-392 -- the training objective's FORWARD pass on the HIP path: prepare_batch, shared
        random timestep, q_sample, apply_model (cfg 1, condition dropout when self.training), MSE against the noise.
        The value is a plain tensor: there are no backward kernels yet, so it serves validation /
        loss monitoring, not optimisation.  ``noise_source(V, D, S) -> dict(t, noise, depth_noise, drop_rand)`` injects the
        reference's random draws (parity tests); default: torch's device generator in the reference's order."""
        if isinstance(batch, dict) and "_prepared" in batch:      # (batch_latents, batch_cameras, input_latents, input_cameras,
            batch_latents, batch_cameras, input_latents, input_cameras, clip_v_embed = batch["_prepared"]      # clip_v_embed): benches
        else:
            batch_latents, batch_cameras, input_latents, input_cameras, clip_v_embed = self.prepare_batch(batch, trainer_config)
        V, _, S, _ = batch_latents.shape
        D = self.view_attn.n_pts_per_ray
        dev = batch_latents.device
        if noise_source is not None:
            ns = noise_source(V, D, S)
            t, noise = ns["t"].to(dev), ns["noise"].to(dev)
            depth_noise, drop_rand = ns["depth_noise"].to(dev), ns["drop_rand"].to(dev)
        else:
            t = self.scheduler.sample_random_times(V, share_t=True, device=dev)
            noise = torch.randn_like(batch_latents)
            depth_noise, drop_rand = torch.randn(V, D, S, S, device=dev), torch.rand(V, device=dev)
        sac = self.scheduler.sqrt_alphas_cumprod.to(dev)[t].view(V, 1, 1, 1)
        s1m = self.scheduler.sqrt_one_minus_alphas_cumprod.to(dev)[t].view(V, 1, 1, 1)
        noisy = sac * batch_latents + s1m * noise                                          # scheduler.q_sample (:55-64)
        prev_depth = input_latents[:, 4:].clone() if self.feed_prev_depth else None          # (:377-379)
        force = getattr(self, "_force_eager", False)
        self._train_keep_fp32, self._force_eager = True, True      # the backward reads block inputs from the workspace: every fp32 tensor is
        try:                                                      # written (apply_model sets keep_fp32 on ITS engine's context only: ADVICE
            pred = self.apply_model(noisy, batch_cameras, input_latents, input_cameras, clip_v_embed, t, prev_depth=prev_depth,   # r04), and
                                    depth_noise=depth_noise, drop_rand=drop_rand)              # no graph captured in inference mode is replayed
        finally:
            self._train_keep_fp32, self._force_eager = False, force
        if self.objective == "noise":
            target = noise
        elif self.objective == "x_start":
            target = batch_latents
        else:
            raise AssertionError(f"objective {self.objective} not implemented")
        assert self.loss_type == "l2", "loss_type 'l2' is the only one the reference implements (:86-87)"
        loss = torch.nn.functional.mse_loss(target, pred).mean()
        if _aux is not None:
            _aux.update(pred=pred, target=target)
        return loss

    @torch.no_grad()
    def unet_gradients(self, batch, trainer_config, noise_source=None, only_trainable=False):
        """`loss.backward()` (train.py:90-95) through the WHOLE UNet and the per-step vector paths on the HIP backward kernels
        (mvdfusion_amd/backward_unet.py): gradients of every `unet_model.unet_model.*` and `cc_projection.*` parameter, plus the
        gradient w.r.t. the volume features that GridAttn produced (its own backward -- `view_attn.*`, `time_embed.*` -- continues
        from there in `gradients`).  Returns (loss, {state_dict key: gradient}, dvol (V, S, S, D, 768))."""
        from . import backward_blocks as bb
        from . import backward_unet as bu
        unet = self.unet_model.unet_model
        unet._record, self._force_eager = [], True
        try:
            loss, grads, dh = self.head_gradients(batch, trainer_config, noise_source=noise_source)
            record = unet._record
        finally:
            unet._record, self._force_eager = None, False
        V, mc, S, _ = dh.shape
        D = self.view_attn.n_pts_per_ray
        eng = self.engine(V, S, D, False)
        ctx = eng.ctx
        M = V * S * S
        emb = ctx.ws.get("temb.emb", (1, unet.model_channels * 4))
        t_sin = ctx.ws.get("vf.tsin_unet", (1, unet.model_channels))
        tape = bb.Tape(dh.device, prec=ctx.prec, workspace=ctx.gemm_ws, only_trainable=only_trainable)
        dh_rows = dh.permute(0, 2, 3, 1).reshape(M, mc).contiguous()
        context = eng.context[:V].clone()
        g, dcontext, dvol = bu.unet_backward(unet, ctx, tape, record, dh_rows, V, S, D, emb, t_sin, context, eng.vol.view(V, S, S, D, -1))
        grads.update({"unet_model.unet_model." + k: v for k, v in g.items()})
        masks = getattr(self, "_last_drop_masks", None)
        if masks is not None:                      # the conditions were multiplied by the keep masks after their producers
            dcontext = dcontext * masks[0][:, None]
            dvol = dvol * masks[1].view(V, 1, 1, 1, 1)
        grads.update({"cc_projection." + k: v for k, v in bu.cc_projection_backward(self.cc_projection, eng.clip_v_embed[:V], dcontext).items()})
        return loss, grads, dvol

    @torch.no_grad()
    def gradients(self, batch, trainer_config, noise_source=None, only_trainable=False):
        """The complete `loss.backward()` of train.py:90-95 on the HIP path: unet_gradients continued through GridAttn
        (mvdfusion_amd/backward_gridattn.py: final layer, softmax-over-V pooling, 3 DiT blocks, pre layer, grid_sample backward,
        z-embedding) and ViewFusion.time_embed.  Returns (loss, {state_dict key: gradient}) for every parameter the loss depends on;
        with only_trainable the weight gradients of frozen parameters (requires_grad False) are skipped (None) -- their dgrad still runs."""
        from . import backward_blocks as bb
        from . import backward_gridattn as bg
        tune = self.train_autotune_min_flops is not None and not hip.AUTOTUNE
        if tune:          # the forward of the training step runs eagerly (condition dropout): its GEMMs and the backward's are tuned here,
            prev_min = hip.AUTOTUNE_MIN_FLOPS
            hip.AUTOTUNE, hip.AUTOTUNE_MIN_FLOPS = True, float(self.train_autotune_min_flops)      # once per shape (cached), big ones only
        try:
            return self._gradients(batch, trainer_config, noise_source, only_trainable)
        finally:
            if tune:
                hip.AUTOTUNE, hip.AUTOTUNE_MIN_FLOPS = False, prev_min
                hip.release_tuning_buffers()

    def _gradients(self, batch, trainer_config, noise_source, only_trainable):
        from . import backward_blocks as bb
        from . import backward_gridattn as bg
        loss, grads, dvol = self.unet_gradients(batch, trainer_config, noise_source=noise_source, only_trainable=only_trainable)
        V, S, _, D, _ = dvol.shape
        eng = self.engine(V, S, D, False)
        ctx = eng.ctx
        tape = bb.Tape(dvol.device, prec=ctx.prec, workspace=ctx.gemm_ws, only_trainable=only_trainable)
        c = ctx.ws.get("vf.c", (1, 256))
        g, dc = bg.gridattn_backward(self.view_attn, tape, eng, c, dvol.reshape(V * S * S * D, -1).contiguous(), V, S, D)
        grads.update({"view_attn." + k: v for k, v in g.items()})
        t_sin = ctx.ws.get("vf.tsin", (1, 256))
        grads.update({"time_embed." + k: v for k, v in bg.time_embed_backward(self.time_embed, t_sin, dc).items()})
        return loss, grads

    @torch.no_grad()
    def tail_gradients(self, batch, trainer_config, noise_source=None):
        """head_gradients continued through the LAST output block (ResBlock + SpatialTransformer + ViewAlignedFeatureTransformer,
        `output_blocks.11`): every operator kind of the UNet has a backward on the HIP path (mvdfusion_amd/backward_blocks.py).
        Returns (loss, {state_dict key: gradient}, dcat = dL/d(input of that block) (V, 2 mc, S, S))."""
        from . import backward_blocks as bb
        loss, grads, dh = self.head_gradients(batch, trainer_config, noise_source=noise_source)
        unet = self.unet_model.unet_model
        V, mc, S, _ = dh.shape
        D = self.view_attn.n_pts_per_ray
        eng = self.engine(V, S, D, False)
        ctx = eng.ctx
        M = V * S * S
        blk = unet.output_blocks[len(unet.output_blocks) - 1]
        cin = blk[0].channels
        cat, catp = ctx.ws.get("cat", (M, cin)), ctx.ws.planes("catp", M, cin)
        emb = ctx.ws.get("temb.emb", (1, unet.model_channels * 4))
        tape = bb.Tape(dh.device, prec=ctx.prec, workspace=ctx.gemm_ws)
        dh_rows = dh.permute(0, 2, 3, 1).reshape(M, mc).contiguous()
        dcat, g, dcontext, dvol, demb = bb.output_block_backward(tape, ctx, blk, cat, catp, emb, eng.context[:V], eng.vol.view(M * D, -1),
                                                                 dh_rows, V, S, S, D)
        pre = f"unet_model.unet_model.output_blocks.{len(unet.output_blocks) - 1}."
        grads.update({pre + k: v for k, v in g.items()})
        return loss, grads, dcat.view(V, S, S, cin).permute(0, 3, 1, 2).contiguous()

    @torch.no_grad()
    def head_gradients(self, batch, trainer_config, noise_source=None):
        """First slice of train.py:90-95 (`loss.backward()`) on the HIP path: the training forward (p_losses) followed by the
        backward of MSE -> UNet output head (conv3x3 <- SiLU <- GroupNorm32) with the backward kernels of mvdfusion_amd/backward.py.
        Returns (loss, {state_dict key: gradient} for the four head parameters, dL/dh at the head's input (V, mc, S, S) -- where the
        backward currently stops: LayerNorm / attention / GEGLU backward do not exist yet)."""
        from . import backward
        aux = {}
        loss = self.p_losses(batch, trainer_config, noise_source=noise_source, _aux=aux)
        unet = self.unet_model.unet_model
        pred, target = aux["pred"], aux["target"]
        V, C, S, _ = pred.shape
        h, a = unet._head_saved
        rows = lambda t: t.permute(0, 2, 3, 1).reshape(V * S * S, C).contiguous()
        eng = self.engine(V, S, self.view_attn.n_pts_per_ray, False)
        grads, dh = backward.unet_head_backward(unet, h, a, rows(pred), rows(target), V, S, eng.ctx.gemm_ws)
        grads = {"unet_model.unet_model." + k: v for k, v in grads.items()}
        return loss, grads, dh.view(V, S, S, -1).permute(0, 3, 1, 2).contiguous()

    def forward(self, batch, trainer_config):
        """viewfusion_zero_depth_rgb.py:394-397: the training loss.  With autograd enabled and trainable parameters the returned
        scalar carries a graph node whose backward hands the HIP-computed gradients (self.gradients) to the parameters, so the
        reference's loop ``loss = model(batch, cfg); optimizer.zero_grad(); loss.backward(); optimizer.step()`` (train.py:86-95) runs
        unchanged -- under torch's DistributedDataParallel too: the parameters are inputs of that node, so DDP's gradient hooks fire and
        all-reduce over RCCL as usual.  Without autograd (torch.no_grad / eval): the plain loss value."""
        # (view_attn.t_embedder exists in the reference's module tree but its forward never calls it: those parameters stay outside
        #  the graph, exactly like the reference -- DDP(find_unused_parameters=True) treats them as unused)
        params = [(n, p) for n, p in self.named_parameters() if p.requires_grad and not n.startswith("view_attn.t_embedder.")]
        if not torch.is_grad_enabled() or not params:
            return self.p_losses(batch, trainer_config, noise_source=getattr(self, "_noise_source", None))
        return _HipTrainingLoss.apply(self, batch, trainer_config, tuple(n for n, _ in params), *[p for _, p in params])

    def configure_optimizers(self, lr=None, verbose=False):
        lr = self.learning_rate if lr is None else lr
        groups = []
        if self.finetune_projection:
            groups.append({"params": self.cc_projection.parameters(), "lr": lr})
        groups.append({"params": self.unet_model.get_trainable_parameters(), "lr": lr})
        groups.append({"params": self.time_embed.parameters(), "lr": lr})
        groups.append({"params": self.view_attn.parameters(), "lr": lr})
        from .optim import HipAdamW
        return HipAdamW(groups, lr=lr)          # a torch.optim.AdamW whose step() is one HIP launch (mvdfusion_amd/optim.py)
