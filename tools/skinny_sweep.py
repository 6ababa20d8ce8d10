"""Skinny GEMM (csrc/gemm_skinny.hip) vs the tiled kernel + split-K slab sum (csrc/gemm.hip) on the GPS backbone's launch
shapes; weights rotate through a pool larger than the caches so that every launch streams them from HBM (GPU box):
    python tools/skinny_sweep.py"""
import os, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from routeformer_amd import kernels as K
K.set_precision("bf16")
K.SKINNY_MAX_M = K.SKINNY_MAX_M_DEEP = 640  # (the sweep measures the kernels, not the dispatch thresholds)
dev = "cuda"
shapes = [(M, N, K_, bm) for M in (32, 40, 56, 96, 168, 320, 560) for (N, K_) in ((832, 832), (2496, 832), (3328, 832), (832, 3328), (832, 2496))
          for bm in (0, 1)]
g = torch.Generator(device=dev).manual_seed(0)
print(f"{'M':>4} {'N':>5} {'K':>5} mode  tiled us  skinny us   err")
for M, N, K_, bm in shapes:
    pool = max(2, int(600e6 // (N * K_ * 4)))
    Ws = [torch.randn((N, K_) if bm == 0 else (K_, N), device=dev, generator=g) / K_ ** 0.5 for _ in range(min(pool, 48))]
    A = torch.randn(M, K_, device=dev, generator=g)
    bias = torch.randn(N, device=dev, generator=g)
    C0, C1 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    def run(C, W):
        if bm == 0:
            K.gemm(A, K_, 1, W, 1, K_, C, N, M, N, K_, bias=bias)
        else:
            K.gemm(A, K_, 1, W, N, 1, C, N, M, N, K_, bias=bias)
    res = {}
    for skinny in (False, True):
        K.SKINNY_GEMM = skinny
        C = C1 if skinny else C0
        for W in Ws[:4]:
            run(C, W)
        torch.cuda.synchronize()
        st = torch.cuda.Stream()
        gr = torch.cuda.CUDAGraph()  # (graph replay: the host's launch rate is not what is being measured)
        with torch.cuda.stream(st):
            with torch.cuda.graph(gr):
                for W in Ws:
                    run(C, W)
            gr.replay()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                gr.replay()
            e.record()
            torch.cuda.synchronize()
        res[skinny] = s.elapsed_time(e) / (3 * len(Ws)) * 1e3
    err = float((C1 - C0).abs().max() / C0.abs().max())
    print(f"{M:4d} {N:5d} {K_:5d}  {bm}   {res[False]:8.1f}  {res[True]:8.1f}   {err:.1e}")
