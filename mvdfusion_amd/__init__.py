"""mvdfusion_amd -- MI355X-native (gfx950) implementation of MVD-Fusion's multi-view denoising hot path.

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all per-step math runs in
hand-written HIP kernels behind the C ABI declared in ``include/mvd_hip.h`` (``mvdfusion_amd/csrc``).
Importing the package never needs a GPU; calling any op without the built library raises loudly.
"""
__version__ = "0.1.0"

from .configs import install_aliases, model_config  # noqa: E402,F401
