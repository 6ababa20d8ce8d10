"""Per-problem time of the grouped weight-gradient kernel (rf_wgrad_grouped, one entry per launch, graph replay) on the
step's shapes (GPU box):  python tools/wgrad_sweep.py"""
import ctypes, os, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from routeformer_amd import _hip
dev = "cuda"
lib = _hip.lib()
g = torch.Generator(device=dev).manual_seed(0)


def entry(e, dy, x, dw, M, N, K, splits, excl):
    e.dy, e.x, e.dw, e.db = dy.data_ptr(), x.data_ptr(), dw.data_ptr(), None
    e.M, e.N, e.K, e.ld_dy, e.ld_x, e.splits, e.kchunk, e.exclusive = M, N, K, N, K, splits, 0, excl


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    st = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gr):
            for _ in range(8):
                fn()
        gr.replay(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            gr.replay()
        e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (reps * 8) * 1e3


shapes = [(32, 832, 832), (32, 3328, 832), (32, 832, 3328), (96, 2496, 832), (168, 3328, 832), (320, 832, 832), (320, 2496, 832),
          (320, 3328, 832), (320, 832, 3328), (560, 832, 832), (560, 2496, 832), (560, 3328, 832), (560, 832, 3328),
          (12480, 384, 128), (12480, 128, 256), (12480, 256, 128), (12480, 128, 128), (12480, 128, 720), (1280, 384, 128)]
keep = []
print(f"{'M':>6} {'N':>5} {'K':>5} splits   us    GB/s(algorithmic: operands + dW once)")
for M, N, K in shapes:
    dy = torch.randn(M, N, device=dev, generator=g); x = torch.randn(M, K, device=dev, generator=g)
    dw = torch.zeros(N, K, device=dev)
    keep.append((dy, x, dw))
    tiles = -(-N // 64) * -(-K // 64)
    splits = max(1, min(64, M // 128, -(-512 // tiles)))
    arr = (_hip.WgradEntry * 1)()
    entry(arr[0], dy, x, dw, M, N, K, splits, 1 if splits == 1 else 0)
    us = timed(lambda: lib.rf_wgrad_grouped(arr, 1, 1, torch.cuda.current_stream().cuda_stream))
    by = 4.0 * (M * N + M * K + N * K)
    line = f"{M:6d} {N:5d} {K:5d} {splits:5d} {us:8.1f}  {by / us / 1e3:8.1f}"
    print(line)

