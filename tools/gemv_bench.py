import sys, os, torch
sys.path.insert(0, "/root/repo")
from mvdfusion_amd import hip
def t(N, K, M):
    W = torch.randn(N, K).cuda(); b = torch.randn(N).cuda(); x = torch.randn(M, K).cuda(); y = torch.empty(M, N).cuda()
    for _ in range(3): hip.gemv(W, b, x, y)
    torch.cuda.synchronize()
    e0, e1 = hip.Event(), hip.Event()
    e0.record()
    for _ in range(20): hip.gemv(W, b, x, y)
    e1.record()
    print(N, K, M, f"{e0.elapsed_ms(e1)/20*1e3:.1f} us", f"{N*K*4/ (e0.elapsed_ms(e1)/20*1e-3)/1e12:.2f} TB/s")
for N, K, M in [(12480, 768, 8), (320, 768, 8), (1280, 768, 8), (1280, 1280, 8), (14720, 1280, 1), (320,320,8)]:
    t(N, K, M)
