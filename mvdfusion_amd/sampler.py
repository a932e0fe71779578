"""50-step stochastic DDIM sampler -- mirror of ``mvdfusion.sampler.DDIMSampler`` (mvdfusion/sampler.py:13-147).

``sample`` keeps the reference signature and return values but runs the loop as 50 replays of one captured hipGraph
(GridAttn + CFG-batched UNet + CFG combine + DDIM update + iteration counter all on the device); the per-step scalars
come from a device table built here on the host in float64 -> float32 exactly like sampler.py:25-39.
``denoise_apply`` / ``denoise_apply_impl`` are kept for callers that drive single steps (utils/vis_utils.py:30-35).

Noise: the reference draws on the device generator (torch.normal in GridAttn, randn_like in the update).  Here all noise
of a sample is drawn up-front in the reference's order (depth noise, then update noise, per step) either from torch's
device generator or from ``noise_source`` -- a callable ``(V, S, D, steps) -> (depth_noise, ddim_noise)`` used by the
parity tests to inject host-generated noise (SURVEY.md trap T2).
"""
import numpy as np
import torch

from .engine import ddim_step_table


def make_ddim_timesteps(num_ddim_timesteps, num_ddpm_timesteps):
    """'uniform' discretisation, +1 shift (external/sd1/ldm/modules/diffusionmodules/util.py:46-60)."""
    c = num_ddpm_timesteps // num_ddim_timesteps
    return np.asarray(list(range(0, num_ddpm_timesteps, c))) + 1


class DDIMSampler:
    def __init__(self, model, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0.0, latent_size=32,
                 overwrite_x_noisy=False, z_dim=4, feed_prev_depth=False):
        assert ddim_discretize == "uniform" and not overwrite_x_noisy
        self.model = model
        self.ddpm_num_timesteps = model.scheduler.num_timesteps
        self.latent_size, self.eta, self.z_dim = latent_size, ddim_eta, z_dim
        self.overwrite_x_noisy, self.feed_prev_depth = overwrite_x_noisy, feed_prev_depth
        self.noise_source = None
        self._make_schedule(ddim_num_steps, ddim_eta)

    def _make_schedule(self, ddim_num_steps, ddim_eta):
        self.ddim_timesteps = make_ddim_timesteps(ddim_num_steps, self.ddpm_num_timesteps)
        ts = torch.from_numpy(self.ddim_timesteps.astype(np.int64))
        ac = self.model.scheduler.alphas_cumprod.detach().cpu()
        a = ac[ts].double()
        a_prev = torch.cat([ac[0:1], ac[ts[:-1]]], 0)
        sig = ddim_eta * torch.sqrt((1 - a_prev) / (1 - a) * (1 - a / a_prev))
        self.ddim_alphas_raw = self.model.scheduler.alphas.detach().cpu()[ts].float()
        self.ddim_sigmas = sig.float()
        self.ddim_alphas = a.float()
        self.ddim_alphas_prev = a_prev.float()
        self.ddim_sqrt_one_minus_alphas = torch.sqrt(1.0 - self.ddim_alphas).float()

    def tables(self):
        sch = self.model.scheduler
        st = {k: getattr(sch, k).detach().cpu() for k in ("sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod")}
        dd = {"timesteps": torch.from_numpy(self.ddim_timesteps.astype(np.int64)), "alphas": self.ddim_alphas,
              "alphas_prev": self.ddim_alphas_prev, "sigmas": self.ddim_sigmas,
              "sqrt_one_minus_alphas": self.ddim_sqrt_one_minus_alphas}
        return st, dd

    @torch.no_grad()
    def denoise_apply_impl(self, x_target_noisy, index, noise_pred, is_step0=False):
        dev = x_target_noisy.device
        a_t = self.ddim_alphas[index].to(dev).view(1, 1, 1, 1)
        a_prev = self.ddim_alphas_prev[index].to(dev).view(1, 1, 1, 1)
        s1m = self.ddim_sqrt_one_minus_alphas[index].to(dev).view(1, 1, 1, 1)
        sigma = self.ddim_sigmas[index].to(dev).view(1, 1, 1, 1)
        pred_x0 = (x_target_noisy - s1m * noise_pred) / a_t.sqrt()
        x_prev = a_prev.sqrt() * pred_x0 + torch.clamp(1.0 - a_prev - sigma ** 2, min=1e-7).sqrt() * noise_pred
        if not is_step0:
            x_prev = x_prev + sigma * torch.randn_like(x_target_noisy)
        return x_prev, pred_x0

    @torch.no_grad()
    def denoise_apply(self, x_target_noisy, batch_cameras, input_latents, input_cameras, clip_embed, time_steps, index,
                      is_step0=False, prev_depth=None, cfg_scale=1.0):
        eps = self.model.apply_model(x_target_noisy, batch_cameras, input_latents, input_cameras, clip_embed, time_steps,
                                     prev_depth=prev_depth if self.feed_prev_depth else None, cfg_scale=cfg_scale)      # (:83-86)
        return self.denoise_apply_impl(x_target_noisy, index, eps, is_step0)

    @torch.no_grad()
    def sample(self, batch_cameras, input_latents, input_cameras, clip_embed, unconditional_scale=1.0, depth=False,
               return_intermediates=False, verbose=True, x_T=None, num_steps=None, use_graph=True):
        """Returns x_0 (V, 5, S, S) [and the per-step {'t','xt','x0'} list].  ``x_T``/``num_steps`` are extensions:
        inject the initial noise / run only the first ``num_steps`` iterations (parity tests, bench warm-up)."""
        assert depth, "MVD-Fusion samples RGB-D latents (depth=True at every call site: demo.py:85-90)"
        m = self.model
        dev = m._device.device
        V, S, D = clip_embed.shape[0], self.latent_size, m.view_attn.n_pts_per_ray
        total = self.ddim_timesteps.shape[0]
        n_run = total if num_steps is None else int(num_steps)
        cfg = unconditional_scale != 1.0
        eng = m.engine(V, S, D, cfg)
        eng.set_conditioning(batch_cameras, input_latents.to(dev), input_cameras, clip_embed.to(dev))
        st, dd = self.tables()
        table = ddim_step_table(st, dd, [total - i - 1 for i in range(total)])
        if x_T is None:
            x_T = torch.randn([V, self.z_dim + 1, S, S], device=dev)
        if self.noise_source is not None:
            dn, sn = self.noise_source(V, S, D, total)
        else:
            dn = torch.randn(total, V, D, S, S, device=dev)
            sn = torch.randn(total, V, 5, S, S, device=dev)
        eng.set_schedule(table, dn, sn)
        eng.x.copy_(x_T)
        inter = []
        for i in range(n_run):
            # feed_prev_depth (:135-140): from the second iteration on GridAttn samples depth around the previous step's x0 estimate, which
            # the step engine keeps in eng.x0 (a second captured graph; the first iteration has no estimate yet)
            eng.depth_mode = 1 if (self.feed_prev_depth and i > 0) else 0
            eng.step(unconditional_scale, do_update=True, use_graph=use_graph)
            if return_intermediates:
                inter.append({"t": int(self.ddim_timesteps[total - i - 1]), "xt": eng.x.clone(), "x0": eng.x0.clone()})
        eng.depth_mode = 0
        from . import hip
        out = hip.check_finite(eng.x.clone(), "DDIMSampler.sample")
        return (out, inter) if return_intermediates else out
