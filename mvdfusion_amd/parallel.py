"""View-parallel denoising: views shard across GPUs, ONE small all-gather per DDIM step.

The UNet treats the V views as an independent batch and GridAttn needs every view's *latents* (a pure function of x_t)
plus the cameras, so rank r owns the query views [q0, q0+Vq), runs GridAttn for those against all V references and the
CFG-batched UNet on those, updates its rows of x_t, and the ranks exchange only their updated (Vq,5,S,S) latent rows
(20 KB per view at S=32) -- latency-bound over xGMI, no bandwidth concern.  The reference has no such path (its only
multi-GPU mode is scene-parallel DDP replicas, demo.py:63-64); this is the faithful way to shard ONE sample: no
per-UNet-block traffic exists in the algorithm.

Noise: every rank draws the FULL-V noise tensors from the same seed and uses its slice, which keeps the sharded run
bit-compatible with the single-GPU run (trap T2).
"""
import torch
import torch.distributed as dist


def view_range(V, rank, world):
    """Contiguous block partition of V views over `world` ranks (first V % world ranks get one extra)."""
    base, extra = divmod(V, world)
    q0 = rank * base + min(rank, extra)
    return q0, base + (1 if rank < extra else 0)


class ViewExchange:
    """The path's only collective: all-gather of each rank's updated latent rows into the replicated x_t."""

    def __init__(self, V, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.ranges = [view_range(V, r, self.world) for r in range(self.world)]
        self.q0, self.Vq = self.ranges[self.rank]
        if self.Vq == 0:
            raise ValueError(f"view-parallel sharding needs world_size <= V (world {self.world}, V {V}): rank {self.rank} "
                             "would own no view")
        self.uniform = len({n for _, n in self.ranges}) == 1
        self._stage = None          # ragged shards: (world, max Vq, ...) staging buffer of the one padded all-gather

    def gather(self, x_full, force=False):
        """x_full (V, ...) holds this rank's fresh rows at [q0, q0+Vq); on return every rank holds all rows.

        Uniform shards: ONE in-place all-gather -- the send buffer is this rank's own slice of the receive buffer (RCCL's
        in-place form: sendbuff == recvbuff + rank * count), enqueued behind the step's kernels on the compute stream's
        dependency chain; no staging copy, nothing on the host's critical path.  `force` runs the collective even in a
        one-rank group (exercises the RCCL code path on a single GPU)."""
        if self.world == 1 and not (force and dist.is_initialized()):
            return x_full
        assert x_full.is_contiguous()
        if self.uniform:
            dist.all_gather_into_tensor(x_full, x_full[self.q0:self.q0 + self.Vq], group=self.group)
        else:
            # ragged (V = 15 over 8 ranks: 2,2,2,2,2,2,2,1): still ONE collective -- every rank's rows padded to the largest shard in a
            # (world, max Vq, ...) staging buffer, gathered in place, then the peers' rows copied to their places (round 6; it was one
            # broadcast per rank: 8 latency-bound collectives per step)
            vmax = max(n for _, n in self.ranges)
            shape = (self.world, vmax) + tuple(x_full.shape[1:])
            if self._stage is None or self._stage.shape != shape or self._stage.device != x_full.device or self._stage.dtype != x_full.dtype:
                self._stage = torch.zeros(shape, dtype=x_full.dtype, device=x_full.device)
            st = self._stage
            st[self.rank, :self.Vq].copy_(x_full[self.q0:self.q0 + self.Vq])
            dist.all_gather_into_tensor(st.view((self.world * vmax,) + tuple(x_full.shape[1:])), st[self.rank], group=self.group)
            for r, (a, n) in enumerate(self.ranges):
                if r != self.rank:
                    x_full[a:a + n].copy_(st[r, :n])
        return x_full


def run_view_parallel(x_T, n_steps, local_step, exchange, force_collective=False):
    """Generic loop: `local_step(i, x_full) -> None` must update rows [q0, q0+Vq) of x_full in place."""
    x = x_T
    for i in range(n_steps):
        local_step(i, x)
        exchange.gather(x, force=force_collective)
    return x


@torch.no_grad()
def sample_view_parallel(model, batch_cameras, input_latents, input_cameras, clip_embed, cfg_scale, x_T, depth_noise,
                         ddim_noise, num_steps=None, use_graph=True, group=None, force_collective=False):
    """DDIMSampler.sample with the views sharded over the ranks of `group` (one process per GPU, RCCL)."""
    from .engine import ddim_step_table
    samp = model.ddim
    dev = model._device.device
    V, S, D = clip_embed.shape[0], samp.latent_size, model.view_attn.n_pts_per_ray
    ex = ViewExchange(V, group)
    total = samp.ddim_timesteps.shape[0]
    n_run = total if num_steps is None else int(num_steps)
    eng = model.engine(V, S, D, cfg_scale != 1.0, q0=ex.q0, Vq=ex.Vq)
    eng.set_conditioning(batch_cameras, input_latents.to(dev), input_cameras, clip_embed.to(dev))
    st, dd = samp.tables()
    eng.set_schedule(ddim_step_table(st, dd, [total - i - 1 for i in range(total)]), depth_noise, ddim_noise)
    eng.x.copy_(x_T)

    def local_step(i, x):
        # feed_prev_depth (mvdfusion/sampler.py:83-84,135-140): from the second iteration on GridAttn samples the depth of a QUERY view
        # around that view's previous x0 estimate -- a row this rank produced itself (eng.x0[q0:q0+Vq]), so no extra exchange is needed
        eng.depth_mode = 1 if (samp.feed_prev_depth and i > 0) else 0
        eng.step(cfg_scale, do_update=True, use_graph=use_graph)

    try:
        run_view_parallel(eng.x, n_run, local_step, ex, force_collective=force_collective)
    finally:
        eng.depth_mode = 0          # engines are cached per (V, S, D, cfg, shard): never leak a depth mode into the next caller
    return eng.x.clone()
