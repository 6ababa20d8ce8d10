"""What does the vendor library (hipBLASLt through torch) reach on the GPS backbone's medium-M products?  A yardstick for
rf_gemm at M = 320 / 560, not a code path of the product.  GPU box only:  python tools/gemm_lib_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from routeformer_amd import kernels as K
K.set_precision("bf16")
dev = "cuda"
shapes = [(320, 2496, 832), (320, 832, 832), (320, 3328, 832), (320, 832, 3328), (336, 832, 2496),
          (560, 2496, 832), (560, 832, 832), (560, 3328, 832), (560, 832, 3328), (168, 3328, 832), (96, 832, 3328)]
NW = 24  # rotate through NW weight copies so that no launch finds its weight in L2 (as in the step: 300 MB of weights per step)
for M, N, Kd in shapes:
    x32 = torch.randn(M, Kd, device=dev)
    dy32 = torch.randn(M, N, device=dev)
    w32 = [torch.randn(N, Kd, device=dev) / Kd ** 0.5 for _ in range(NW)]
    xb, wb = x32.bfloat16(), [w.bfloat16() for w in w32]

    def rf(i, dx=False):
        if dx:
            return K._input_grad(dy32, w32[i])
        y = torch.empty(M, N, device=dev)
        K.gemm(x32, Kd, 1, w32[i], 1, Kd, y, N, M, N, Kd)
        return y
    res = {}
    for name, fn in (("rf_gemm y = x W^T", lambda i: rf(i)), ("rf_gemm dX = dY W", lambda i: rf(i, True)),
                     ("hipBLASLt bf16 x bf16", lambda i: torch.nn.functional.linear(xb, wb[i])),
                     ("vendor fp32", lambda i: torch.nn.functional.linear(x32, w32[i]))):
        for i in range(NW):
            fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(NW):
                y = fn(i)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / (10 * NW) * 1e6
    print(f"M {M:4d} N {N:5d} K {Kd:5d}: " + " | ".join(f"{k} {v:6.1f} us" for k, v in res.items()), flush=True)
