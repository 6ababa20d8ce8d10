"""Micro-benchmark of the GEMM shapes of the C2 train step (per-launch time, TFLOP/s, GB/s)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from routeformer_amd import kernels as K
K.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
dev = "cuda"
shapes = [  # (tag, M, N, K)
    ("frame qkv", 12480, 384, 128), ("frame out", 12480, 128, 128), ("frame ffn1", 12480, 256, 128),
    ("frame ffn2", 12480, 128, 256), ("frame tok", 12480, 128, 720),
    ("fusion qkv", 1280, 384, 128), ("gaze qkv", 320, 384, 128),
    ("inf qkv L40", 320, 2496, 832), ("inf out L40", 320, 832, 832), ("inf ffn1 L40", 320, 3328, 832),
    ("inf ffn2 L40", 320, 832, 3328), ("inf qkv L5", 40, 2496, 832), ("inf ffn1 L5", 40, 3328, 832),
    ("dec qkv", 560, 2496, 832), ("dec ffn1", 560, 3328, 832), ("dec ffn2", 560, 832, 3328),
    ("distil conv", 336, 832, 2496), ("dec proj", 560, 66, 832),
]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us
print(f"{'shape':14s} {'M':>6s} {'N':>5s} {'K':>5s} | {'fwd us':>8s} {'TF/s':>6s} | {'dX us':>8s} {'TF/s':>6s} | {'dW us':>8s} {'TF/s':>6s} | wbytes/fwd GB/s")
for tag, M, N, Kd in shapes:
    x = torch.randn(M, Kd, device=dev); w = torch.randn(N, Kd, device=dev); b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev); y = torch.empty(M, N, device=dev)
    f = 2.0 * M * N * Kd
    t_f = timeit(lambda: K.gemm(x, Kd, 1, w, 1, Kd, y, N, M, N, Kd, bias=b))
    t_x = timeit(lambda: K._input_grad(dy, w))
    t_w = timeit(lambda: K._weight_grad(dy, x))
    print(f"{tag:14s} {M:6d} {N:5d} {Kd:5d} | {t_f:8.1f} {f/t_f/1e6:6.1f} | {t_x:8.1f} {f/t_x/1e6:6.1f} | {t_w:8.1f} {f/t_w/1e6:6.1f} | {4*N*Kd/t_f/1e3:8.1f}")
