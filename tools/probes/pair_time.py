"""Fused BasicBlock pair (rf_conv3x3_pair_group_bf16) against the two-launch form, per shape, under graph replay."""
import math, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from routeformer_amd import _hip, kernels as Kn
from routeformer_amd.models.video_backbone.hrnet16 import pack_conv3x3_weights
DEV = "cuda"
def timeit(fn, n=20, reps=3):
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
    return s.elapsed_time(e) / (n * reps) * 1e3
def case(shapes):
    g = torch.Generator().manual_seed(1)
    arr = (_hip.ConvPairEntry * len(shapes))(); carr = (_hip.ConvEntry * len(shapes))(); carr2 = (_hip.ConvEntry * len(shapes))(); keep = []
    for i, (N, H, W, C) in enumerate(shapes):
        x = torch.randn(N, H, W, C, generator=g).bfloat16().to(DEV)
        ws = [(torch.randn(C, 3, 3, C, generator=g) / math.sqrt(9 * C)).to(DEV) for _ in range(2)]
        bs = [(torch.randn(C, generator=g) * 0.2).to(DEV) for _ in range(2)]
        wp = [pack_conv3x3_weights(w) for w in ws]
        y, mid = torch.empty_like(x), torch.empty_like(x)
        e = arr[i]
        e.x, e.w1_packed, e.bias1, e.w2_packed, e.bias2, e.y = x.data_ptr(), wp[0].data_ptr(), bs[0].data_ptr(), wp[1].data_ptr(), bs[1].data_ptr(), y.data_ptr()
        e.N, e.H, e.W, e.c = N, H, W, C
        for a, (src, dst, k, res) in ((carr, (x, mid, 0, None)), (carr2, (mid, y, 1, x))):
            c = a[i]
            c.x, c.w_packed, c.bias, c.residual, c.y = src.data_ptr(), wp[k].data_ptr(), bs[k].data_ptr(), (res.data_ptr() if res is not None else None), dst.data_ptr()
            c.N, c.H, c.W, c.cin, c.cout, c.relu = N, H, W, C, C, 1
        keep.append((x, ws, bs, wp, y, mid))
    st = Kn._stream
    t_pair = timeit(lambda: _hip.lib().rf_conv3x3_pair_group_bf16(arr, len(shapes), st()))
    if len(shapes) == 1:
        N, H, W, C = shapes[0]; x, ws, bs, wp, y, mid = keep[0]
        def two():
            _hip.lib().rf_conv3x3_bf16(x.data_ptr(), wp[0].data_ptr(), bs[0].data_ptr(), None, mid.data_ptr(), 1, N, H, W, C, C, 1, st())
            _hip.lib().rf_conv3x3_bf16(mid.data_ptr(), wp[1].data_ptr(), bs[1].data_ptr(), x.data_ptr(), y.data_ptr(), 1, N, H, W, C, C, 1, st())
    else:
        def two():
            _hip.lib().rf_conv3x3_group_bf16(carr, len(shapes), 1, st())
            _hip.lib().rf_conv3x3_group_bf16(carr2, len(shapes), 1, st())
    t_two = timeit(two)
    print(f"{str(shapes):70s} pair {t_pair:7.1f} us   two launches {t_two:7.1f} us")
N2, N5 = 336, 252
for shapes in ([(N2, 28, 28, 16)], [(N2, 14, 14, 32)], [(N2, 14, 14, 32), (N2, 7, 7, 64)], [(N2, 14, 14, 32), (N2, 7, 7, 64), (N2, 4, 4, 128)],
               [(N5, 56, 56, 16)], [(N5, 28, 28, 32)], [(N5, 28, 28, 32), (N5, 14, 14, 64), (N5, 7, 7, 128)]):
    case(shapes)
