"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of MVD-Fusion's per-DDIM-step denoising hot path:
plain fp32 PyTorch-on-CPU, written functionally over a flat ``{state_dict key: tensor}`` dict.

Who may import this package: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` -- and there only as the checker / the timed CPU baseline.  The product package
(``mvdfusion_amd``) never imports it and fails loudly when its HIP library is missing.

Parity pinning: the reference has no tests / golden vectors of its own, so the
restatement in ``oracle/ref_torch.py`` is pinned against the reference *itself*, imported in the build
container from ``/root/reference`` by ``oracle/make_golden.py`` (third-party deps that are absent from
the reference tree -- pytorch3d, timm, omegaconf -- are restated in ``oracle/shims.py``; their algebra
is pinned by the known-answer tests in ``tests/test_cpu_oracle_and_host.py`` (test_kat_*)).  The resulting input /
output vectors are committed under ``tests/golden/`` and are what the GPU box checks against.
"""
