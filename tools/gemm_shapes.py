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
def timeit(fn, n=20, reps=5):
    """n launches captured in a HIP graph (no host launch overhead in the number), replayed `reps` times."""
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (n * reps) * 1e3  # us
print(f"{'shape':14s} {'M':>6s} {'N':>5s} {'K':>5s} | {'fwd us':>8s} {'TF/s':>6s} | {'dX us':>8s} {'TF/s':>6s} | {'dW us':>8s} {'TF/s':>6s} | wbytes/fwd GB/s")
for tag, M, N, Kd in shapes:
    x = torch.randn(M, Kd, device=dev); w = torch.randn(N, Kd, device=dev); b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev); y = torch.empty(M, N, device=dev)
    f = 2.0 * M * N * Kd
    t_f = timeit(lambda: K.gemm(x, Kd, 1, w, 1, Kd, y, N, M, N, Kd, bias=b))
    t_x = timeit(lambda: K._input_grad(dy, w))
    gw = torch.zeros(N, Kd, device=dev)
    t_w = timeit(lambda: K._weight_grad(dy, x, into=gw))
    print(f"{tag:14s} {M:6d} {N:5d} {Kd:5d} | {t_f:8.1f} {f/t_f/1e6:6.1f} | {t_x:8.1f} {f/t_x/1e6:6.1f} | {t_w:8.1f} {f/t_w/1e6:6.1f} | {4*N*Kd/t_f/1e3:8.1f}")

# row-block kernels vs the generic path (bf16 only)
if K.get_precision() == "bf16":
    from routeformer_amd import _hip
    from routeformer_amd._hip import ptr
    print("row-block kernels:")
    for M in (12480, 9360, 1280, 320):
        x = torch.randn(M, 128, device=dev); r = torch.randn(M, 128, device=dev)
        wq = torch.randn(384, 128, device=dev); bq = torch.randn(384, device=dev)
        wo = torch.randn(128, 128, device=dev); bo = torch.randn(128, device=dev)
        w1 = torch.randn(256, 128, 1, device=dev); b1 = torch.randn(256, device=dev)
        w2 = torch.randn(128, 256, 1, device=dev); b2 = torch.randn(128, device=dev)
        gam = torch.ones(128, device=dev); bet = torch.zeros(128, device=dev)
        res = {}
        for fused in (True, False):
            K.ROWBLOCK = fused
            with torch.no_grad():
                res[fused] = (timeit(lambda: K.linear(x, wq, bq)), timeit(lambda: K.linear_add_layer_norm(x, wo, bo, r, gam, bet)),
                              timeit(lambda: K.ffn_add_layer_norm(x, w1, b1, w2, b2, "gelu", gam, bet)))
            with torch.enable_grad():
                res[fused] += (timeit(lambda: K.ffn_add_layer_norm(x, w1, b1, w2, b2, "gelu", gam, bet)),)
        K.ROWBLOCK = True
        print(f"M={M:6d}  qkv {res[True][0]:6.1f} (was {res[False][0]:6.1f})  out+ln {res[True][1]:6.1f} (was {res[False][1]:6.1f})  "
              f"ffn+ln nograd {res[True][2]:6.1f} (was {res[False][2]:6.1f})  ffn+ln saving h,z {res[True][3]:6.1f} (was {res[False][3]:6.1f}) us")
