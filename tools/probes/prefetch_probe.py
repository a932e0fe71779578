"""What a weight prefetch could buy: mvd_gemm with WARM weights (one packed weight), COLD weights (launches rotate over > 600 MB of copies: every
launch streams its weight from HBM, as in a DDIM step whose 3.4 GB weight set cycles through the 256 MB Infinity Cache) and HEAD-WARM weights
(cold copies whose first HEAD k-tiles were touched by a small kernel one launch earlier -- what a prefetch issued from the previous
kernel's tail would leave in the Infinity Cache / translation caches).  The touch kernel's own time is measured separately and subtracted.
    python tools/probes/prefetch_probe.py"""
import copy, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mvdfusion_amd import hip

SHAPES = [("proj32", 8192, 320, 320), ("ff2_32", 8192, 320, 1280), ("qkv32", 8192, 960, 320), ("proj16", 2048, 640, 640),
          ("ff2_16", 2048, 640, 2560), ("proj8", 512, 1280, 1280), ("qkv8", 512, 3840, 1280), ("ff2_8", 512, 1280, 5120)]
DIST = int(os.environ.get("DIST", "1"))          # how many launches ahead of its consumer a weight is touched
HEAD = int(os.environ.get("HEAD", "4"))          # k-tiles of every 16-column block that are touched (0 = the whole weight)


def bench(fn, reps):
    g = hip.Graph()
    with g:
        for i in range(reps):
            fn(i)
    g.launch()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = hip.Event(), hip.Event()
        e0.record(); g.launch(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_ms(e1) / reps)
    return best * 1e3


def main():
    ws = torch.empty(64 * 1024 * 1024, device="cuda")
    y = torch.zeros(1, 65536, device="cuda")
    x = torch.ones(1, 1024, device="cuda")
    print(f"{'shape':8s} {'copies':>6s} {'warm us':>8s} {'cold us':>8s} {'head-warm us':>12s} {'touch us':>8s}   (head = first {HEAD or 'all'} k-tiles, touched {DIST} launches ahead)")
    for name, M, N, K in SHAPES:
        g = torch.Generator().manual_seed(0)
        A = hip.split_planes(torch.randn(M, K, generator=g).cuda())
        W = hip.pack_linear((torch.randn(N, K, generator=g) / math.sqrt(K)).cuda(), torch.zeros(N).cuda())
        wbytes = W.data.numel()
        ncopy = min(800, (640 << 20) // wbytes + 1)
        Ws = []
        for _ in range(ncopy):
            w2 = copy.copy(W)
            w2.data = W.data.clone()
            Ws.append(w2)
        out = torch.empty(M, N, device="cuda")
        R = torch.randn(M, N, generator=g).cuda()
        reps = ncopy
        head_bytes = (HEAD if HEAD else W.K // 32) * W.N * 128                      # packed layout [K/32][N/16][2 KiB]: k-tile t of all columns is one contiguous run
        rows = head_bytes // 4096
        heads = [w.data[:rows * 4096].view(torch.float32).view(rows, 1024) for w in Ws]

        def run(i, Wl):
            hip.gemm(A, Wl[i % len(Wl)], out, prec=3, res=R, workspace=ws)

        def touch(i):
            hip.gemv(heads[(i + DIST) % ncopy], None, x, y[:, :rows])

        warm = bench(lambda i: run(i, [W]), reps)
        cold = bench(lambda i: run(i, Ws), reps)
        both = bench(lambda i: (touch(i), run(i, Ws)), reps)
        t_only = bench(touch, reps)
        print(f"{name:8s} {ncopy:6d} {warm:8.1f} {cold:8.1f} {both - t_only:12.1f} {t_only:8.1f}", flush=True)
        del Ws, heads
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
