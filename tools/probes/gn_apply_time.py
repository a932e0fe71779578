"""Time mvd_groupnorm_from_stats (gn_apply_stats_kernel) on the step's shapes: HIP events around a graph of 50 launches."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from mvdfusion_amd import hip

for (B, HW, C) in ((8, 1024, 320), (8, 1024, 640), (8, 1024, 960), (8, 256, 640), (8, 256, 1280), (8, 256, 1920), (8, 64, 1280), (8, 64, 2560), (8, 16, 2560),
                   (16, 1024, 320), (16, 1024, 960), (16, 4096, 320)):
    x = torch.randn(B * HW, C, device="cuda")
    y = hip.planes_like(B * HW, C, "cuda")
    gamma, beta = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    stats = torch.zeros(B, 32, 2, dtype=torch.int64, device="cuda")
    xs = x.view(B, HW, 32, C // 32).double()
    stats[:, :, 0] = (xs.sum(dim=(1, 3)) * (1 << 24)).round().long()
    stats[:, :, 1] = ((xs * xs).sum(dim=(1, 3)) * (1 << 24)).round().long()
    hip.groupnorm_from_stats(x, y, gamma, beta, stats, B, HW, C, 1e-5, True)
    torch.cuda.synchronize()
    g = hip.Graph()
    with g:
        for _ in range(50):
            hip.groupnorm_from_stats(x, y, gamma, beta, stats, B, HW, C, 1e-5, True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        g.launch()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
    print(f"gn_apply B={B} HW={HW} C={C}: {best:6.2f} us / launch  ({B * HW * C * 8 / best / 1e6:5.2f} TB/s)  checksum {float(y.view(torch.int16).float().abs().sum()):.6e}", flush=True)
