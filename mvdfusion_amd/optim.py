"""torch.optim.AdamW (train.py:137 -> ViewFusion.configure_optimizers -> torch.optim.AdamW(groups, lr)) with its step as ONE HIP launch.

`HipAdamW` IS a torch.optim.AdamW (same constructor, same `state` layout: per parameter 'step', 'exp_avg', 'exp_avg_sq'), so checkpoints
written by either load into the other (train.py:150,178); only `step()` differs: instead of torch's multi-tensor passes over the 1 G trainable
parameters (~8 kernel kinds, 25 ms) one `mvd_adamw_multi` launch applies the same update (include/mvd_hip.h) -- weight decay, lerp of
exp_avg, exp_avg_sq, bias corrections, addcdiv -- reading parameter, gradient and both moments once and writing three.  Parameters that are
not fp32 CUDA tensors, sparse gradients, amsgrad / maximize / capturable / differentiable / fused / foreach=True settings: the whole step
falls back to torch's.  "The same update" = the same formula; the kernel evaluates lr / bias_correction1 and the decay factor in fp32 where
torch's single-tensor path uses Python doubles, so the two agree to a few ulp (tests/test_gpu_backward.py: 2e-7 relative), not bitwise."""
import ctypes as C
import math

import torch

from . import hip

ADAMW_CHUNK = 4096


class _AdamwTensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("numel", C.c_ulonglong),
                ("first_chunk", C.c_uint), ("pad", C.c_uint)]


class HipAdamW(torch.optim.AdamW):
    def _plain(self, group):
        # (fused=True keeps state['step'] on the device and foreach=True asks for torch's multi-tensor path by name: both are
        #  torch's to run)
        return not (group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("differentiable") or
                    group.get("fused") or group.get("foreach"))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        def plain(group):
            return self._plain(group) and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and not p.grad.is_sparse and
                                              p.grad.dtype == torch.float32 for p in group["params"] if p.grad is not None)

        if not all(plain(g) for g in self.param_groups):
            # anything the one-launch kernel does not cover (CPU tensors, another dtype, amsgrad / maximize / capturable, sparse gradients):
            # torch's own step for the whole optimizer -- same state layout, so the two can alternate
            super().step()
            return loss
        for group in self.param_groups:
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            steps = set()
            for p in params:
                st = self.state[p]
                if len(st) == 0:                       # torch.optim.AdamW._init_group
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                steps.add(float(st["step"]))
            beta1, beta2 = group["betas"]
            for step in sorted(steps):                 # (one launch per distinct step count: parameters that joined later)
                part = [p for p in params if float(self.state[p]["step"]) == step]
                self._launch(part, group, beta1, beta2, step)
        return loss

    def _launch(self, params, group, beta1, beta2, step):
        arr = (_AdamwTensor * len(params))()
        chunk = 0
        keep = []
        for i, p in enumerate(params):
            st = self.state[p]
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            keep.append(g)
            arr[i].p, arr[i].g, arr[i].m, arr[i].v = p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            arr[i].numel, arr[i].first_chunk = p.numel(), chunk
            chunk += (p.numel() + ADAMW_CHUNK - 1) // ADAMW_CHUNK
        dev = params[0].device
        table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev, non_blocking=False)
        with torch.cuda.device(dev):
            hip.check(hip.lib().mvd_adamw_multi(hip.ptr(table), len(params), chunk, float(group["lr"]), float(beta1), float(beta2),
                                                float(group["eps"]), float(group["weight_decay"]), 1.0 - beta1 ** step,
                                                math.sqrt(1.0 - beta2 ** step), 1.0, hip.stream()))
        self._keep = (table, keep)                      # alive until the next step (the launch is asynchronous)
        # the kernel wrote through raw pointers: tell autograd / every cache keyed on tensor versions (hip.params_signature: packed weight
        # images, hip._PARAM_MAX) that parameters and moments changed
        for p in params:
            st = self.state[p]
            torch.autograd.graph.increment_version(p)
            torch.autograd.graph.increment_version(st["exp_avg"])
            torch.autograd.graph.increment_version(st["exp_avg_sq"])
