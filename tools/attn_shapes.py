"""Micro-benchmark of the attention problem shapes of the C2 train step."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from routeformer_amd import kernels as K
dev = "cuda"
# tag, B(seqs), H, LQ, LK, E, mode, factor, layout
shapes = [("frame enc", 192, 8, 65, 65, 16, 1, 5, 0), ("frame enc tgt", 144, 8, 65, 65, 16, 1, 5, 0),
          ("fusion", 8, 8, 160, 160, 16, 1, 5, 0), ("fusion tgt", 8, 8, 120, 120, 16, 1, 5, 0),
          ("gaze enc", 8, 8, 40, 40, 16, 1, 5, 0), ("dec self", 8, 8, 40, 40, 8, 2, 5, 0), ("dec cross", 8, 8, 40, 40, 8, 0, 5, 0),
          ("inf enc L40", 8, 8, 40, 40, 104, 1, 4, 1), ("inf enc L21", 8, 8, 21, 21, 104, 1, 4, 1), ("inf enc L4", 8, 8, 4, 4, 104, 1, 4, 1),
          ("inf dec self", 8, 8, 70, 70, 104, 2, 4, 1), ("inf dec cross", 8, 8, 70, 4, 104, 1, 4, 1)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
print(f"{'shape':14s} {'B*H':>5s} {'LQ':>4s} {'LK':>4s} {'E':>4s} mode | fwd us | bwd us")
for tag, B, H, LQ, LK, E, mode, factor, layout in shapes:
    HE = H * E
    if LQ == LK:
        a = torch.randn(B * LQ, 3 * HE, device=dev, requires_grad=True); b = a; offs = (0, HE, 2 * HE)
    else:
        a = torch.randn(B * LQ, HE, device=dev, requires_grad=True); b = torch.randn(B * LK, 2 * HE, device=dev, requires_grad=True); offs = (0, 0, HE)
    sk, nt = (0, 0) if mode == 0 else K.prob_sizes(LQ, LK, factor)
    idx = None if mode == 0 else torch.randint(LK, (LQ, sk), device=dev, dtype=torch.int32)
    out = K.attention(a, b, offs, (B, H, LQ, LK, E), mode, index_sample=idx, n_top=nt, out_layout=layout)
    g = torch.randn_like(out)
    ad = a.detach(); bd = ad if a is b else b.detach()
    t_f = timeit(lambda: K.attention(ad, bd, offs, (B, H, LQ, LK, E), mode, index_sample=idx, n_top=nt, out_layout=layout))
    def fb():
        o = K.attention(a, b, offs, (B, H, LQ, LK, E), mode, index_sample=idx, n_top=nt, out_layout=layout)
        o.backward(g)
    t_fb = timeit(fb)
    print(f"{tag:14s} {B*H:5d} {LQ:4d} {LK:4d} {E:4d} {mode:4d} | {t_f:6.1f} | {t_fb - t_f:6.1f}")
