"""DDPM noise schedule -- mirror of ``mvdfusion.scheduler.DDPMScheduler`` (mvdfusion/scheduler.py:9-74).

Same constructor, buffer names (state_dict keys ``scheduler.*``) and methods; tables are tiny host-built tensors.
SD "scaled-linear" betas: linspace(sqrt(0.00085), sqrt(0.012), T, fp32)**2, cumprod in fp32 (scheduler.py:15-22).
"""
import torch
import torch.nn as nn


def make_tables(timesteps=1000):
    betas = torch.linspace(0.00085 ** 0.5, 0.0120 ** 0.5, timesteps, dtype=torch.float32) ** 2
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, dim=0)
    ac_prev = torch.cat([torch.ones(1, dtype=torch.float64), ac[:-1]], 0)
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    post_logvar = torch.clamp(torch.log(torch.clamp(post_var, min=1e-20)), min=-10)
    return {
        "betas": betas.float(),
        "alphas": alphas.float(),
        "alphas_cumprod": ac.float(),
        "sqrt_alphas_cumprod": torch.sqrt(ac).float(),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1 - ac).float(),
        "sqrt_recip_alphas_cumprod": torch.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": torch.sqrt(1.0 / ac - 1),
        "posterior_variance": post_var.float(),
        "posterior_log_variance_clipped": post_logvar.float(),
    }


class DDPMScheduler(nn.Module):
    def __init__(self, timesteps):
        super().__init__()
        self.num_timesteps = timesteps
        for k, v in make_tables(timesteps).items():
            self.register_buffer(k, v)
        self.register_buffer("_device", torch.tensor([0.0]), persistent=False)

    def sample_random_times(self, b, share_t=True, device=None):
        device = self._device.device if device is None else device
        t = torch.randint(0, self.num_timesteps, (b,), device=device).long()
        return torch.zeros_like(t) + t[0] if share_t else t

    def _bc(self, table, t, x):
        return table[t].view(x.shape[0], *([1] * (x.dim() - 1)))

    def q_sample(self, x_start, t):
        noise = torch.randn_like(x_start)
        return self._bc(self.sqrt_alphas_cumprod, t, x_start) * x_start + \
            self._bc(self.sqrt_one_minus_alphas_cumprod, t, x_start) * noise, noise

    def predict_start_from_noise(self, x_noisy, eps, t):
        return self._bc(self.sqrt_recip_alphas_cumprod, t, x_noisy) * x_noisy - \
            self._bc(self.sqrt_recipm1_alphas_cumprod, t, x_noisy) * eps
