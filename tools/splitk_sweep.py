"""Split-K sweep for the small-M GEMMs of the step (Informer d=832, fusion / gaze d=128): per-launch time of the
forward (NT) and input-gradient (NN) forms for every split factor, graph-timed, next to what the heuristic in
kernels._auto_split picks.  GPU box only:  python tools/splitk_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from routeformer_amd import kernels as K  # noqa: E402

K.set_precision("bf16")
dev = "cuda"
SPLITS = (1, 2, 3, 4, 6, 8, 12, 16)
shapes = [  # (tag, M, N(out features), K(in features))
    ("enc qkv L40", 320, 2496, 832), ("enc out L40", 320, 832, 832), ("enc ffn1 L40", 320, 3328, 832),
    ("enc ffn2 L40", 320, 832, 3328), ("enc qkv L21", 168, 2496, 832), ("enc ffn1 L21", 168, 3328, 832),
    ("enc ffn2 L21", 168, 832, 3328), ("enc qkv L5", 40, 2496, 832), ("enc ffn2 L5", 40, 832, 3328),
    ("dec qkv", 560, 2496, 832), ("dec q", 560, 832, 832), ("dec kv L5", 40, 1664, 832), ("dec ffn1", 560, 3328, 832),
    ("dec ffn2", 560, 832, 3328), ("distil conv", 336, 832, 2496), ("fusion qkv", 1280, 384, 128),
    ("fusion ffn2", 1280, 128, 256),
]


def timeit(fn, n=20, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (n * reps) * 1e3


def row(tag, times, auto):
    best = min(times, key=times.get)
    cells = " ".join(f"{times[s]:6.1f}{'*' if s == best else ('a' if s == auto else ' ')}" for s in SPLITS if s in times)
    print(f"{tag:22s} {cells}   auto={auto} ({times.get(auto, float('nan')):.1f}) best={best} ({times[best]:.1f})")


print("columns: split factor " + " ".join(f"{s:7d}" for s in SPLITS) + "   (* best, a = heuristic)")
tot_auto = tot_best = 0.0
for tag, M, N, Kd in shapes:
    x = torch.randn(M, Kd, device=dev)
    w = torch.randn(N, Kd, device=dev)
    b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev)
    y = torch.empty(M, N, device=dev)
    dx = torch.empty(M, Kd, device=dev)
    for form in ("fwd", "dX"):
        times = {}
        depth = Kd if form == "fwd" else N
        for s in SPLITS:
            if s > 1 and depth // s < 64:
                continue
            if form == "fwd":
                times[s] = timeit(lambda: K.gemm(x, Kd, 1, w, 1, Kd, y, N, M, N, Kd, bias=b, splitk=s))
            else:
                times[s] = timeit(lambda: K.gemm(dy, N, 1, w, Kd, 1, dx, Kd, M, Kd, N, splitk=s))
        auto = K._auto_split(M, N, Kd) if form == "fwd" else K._auto_split(M, Kd, N)
        if auto not in times:
            times[auto] = timeit((lambda: K.gemm(x, Kd, 1, w, 1, Kd, y, N, M, N, Kd, bias=b, splitk=auto)) if form == "fwd"
                                 else (lambda: K.gemm(dy, N, 1, w, Kd, 1, dx, Kd, M, Kd, N, splitk=auto)))
        row(f"{tag} {form} {M}x{N if form == 'fwd' else Kd}x{depth}", times, auto)
        tot_auto += times[auto]
        tot_best += min(times.values())
print(f"sum over shapes: heuristic {tot_auto:.1f} us, best {tot_best:.1f} us")
