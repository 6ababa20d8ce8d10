import math, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from routeformer_amd import _hip, kernels as Kn
from routeformer_amd.models.video_backbone.hrnet16 import pack_conv3x3_weights
DEV = "cuda"
def run(shapes):
    g = torch.Generator().manual_seed(1)
    arr = (_hip.ConvPairEntry * len(shapes))(); keep = []; outs = []; want = []
    for i, (N, H, W, C) in enumerate(shapes):
        x = torch.randn(N, H, W, C, generator=g).bfloat16().to(DEV)
        ws = [(torch.randn(C, 3, 3, C, generator=g) / math.sqrt(9 * C)).to(DEV) for _ in range(2)]
        bs = [(torch.randn(C, generator=g) * 0.2).to(DEV) for _ in range(2)]
        wp = [pack_conv3x3_weights(w) for w in ws]
        y = torch.full((N, H, W, C), float("nan"), device=DEV, dtype=torch.bfloat16)
        e = arr[i]
        e.x, e.w1_packed, e.bias1, e.w2_packed, e.bias2, e.y = x.data_ptr(), wp[0].data_ptr(), bs[0].data_ptr(), wp[1].data_ptr(), bs[1].data_ptr(), y.data_ptr()
        e.N, e.H, e.W, e.c = N, H, W, C
        mid, two = torch.empty_like(x), torch.empty_like(x)
        _hip.check(_hip.lib().rf_conv3x3_bf16(x.data_ptr(), wp[0].data_ptr(), bs[0].data_ptr(), None, mid.data_ptr(), 1, N, H, W, C, C, 1, Kn._stream()), "c1")
        _hip.check(_hip.lib().rf_conv3x3_bf16(mid.data_ptr(), wp[1].data_ptr(), bs[1].data_ptr(), x.data_ptr(), two.data_ptr(), 1, N, H, W, C, C, 1, Kn._stream()), "c2")
        keep.append((x, ws, bs, wp, mid)); outs.append(y); want.append(two)
    _hip.check(_hip.lib().rf_conv3x3_pair_group_bf16(arr, len(shapes), Kn._stream()), "pair")
    torch.cuda.synchronize()
    for sh, y, two in zip(shapes, outs, want):
        d = (y.float() - two.float()).abs()
        bad = (d > 0).nonzero()
        print(sh, "max diff", float(d.max()), "n bad", len(bad), "first", bad[:4].tolist(), "rows", sorted(set((bad[:, 0] * sh[1] * sh[2] + bad[:, 1] * sh[2] + bad[:, 2]).tolist()))[:12])
for shapes in ([(3, 28, 28, 32)], [(3, 14, 14, 64)], [(3, 28, 28, 32), (3, 14, 14, 64)], [(3, 28, 28, 64)], [(2, 56, 56, 32)]):
    run(shapes)
