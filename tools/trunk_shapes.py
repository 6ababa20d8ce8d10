"""Per-launch times of the conv trunk's distinct kernel shapes at the bench batch (336 frames of 224x224):
records every convolution / upsample call of one encode pass, then times each distinct shape alone (20 launches
in a HIP graph) and prints time, count per pass, GB/s of map traffic and TFLOP/s.  GPU box only."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from routeformer_amd import kernels as K, synthetic
from routeformer_amd.models.video_backbone.hrnet16 import HRNet16Backbone

K.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
N = int(sys.argv[2]) if len(sys.argv) > 2 else 336
HW = int(sys.argv[3]) if len(sys.argv) > 3 else 224  # (C5: 252 frames of 448 x 448)
net = HRNet16Backbone()
net.load_state_dict(synthetic.synth_state_dict(net.state_dict(), 7))
net = net.to("cuda")
calls = []
orig_conv, orig_up = HRNet16Backbone._conv, HRNet16Backbone._upsample.__func__


def rec_conv(self, W, unit, x, stride=1, relu=False, residual=None):
    w, b, cin, cout, k, wb = W[unit]
    calls.append((f"conv{k}x{k}", tuple(x.shape), cout, k, stride, residual is not None, unit))
    return orig_conv(self, W, unit, x, stride, relu, residual)


HRNet16Backbone._conv = rec_conv
video = torch.rand(N // 12 or 1, 12, 3, HW, HW, device="cuda").half()[: max(1, N // 12)]
tok = net.encode_clips([(video, None)])
torch.cuda.synchronize()
HRNet16Backbone._conv = orig_conv
W = net._prepare(torch.device("cuda"))


def timeit(fn, n=20, reps=3):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (n * reps) * 1e3


groups = collections.OrderedDict()
for kind, shp, cout, k, stride, has_res, unit in calls:
    groups.setdefault((kind, shp, cout, k, stride, has_res), []).append(unit)
print(f"{'kernel':8s} {'in (N,H,W,C)':>22s} {'cout':>4s} k s res | {'us':>7s} x{'n':<3s} {'tot us':>8s} | {'GB/s':>6s} {'TF/s':>6s}  first unit")
total = 0.0
adt = net._act_dtype()
for (kind, shp, cout, k, stride, has_res), units in groups.items():
    x = torch.randn(*shp, device="cuda").to(adt)
    pad = 1 if k == 3 else 0
    Ho, Wo = (shp[1] + 2 * pad - k) // stride + 1, (shp[2] + 2 * pad - k) // stride + 1
    res = torch.randn(shp[0], Ho, Wo, cout, device="cuda").to(adt) if has_res else None
    t = timeit(lambda: orig_conv(net, W, units[0], x, stride, True, res))
    es = x.element_size()
    byts = es * (x.numel() + shp[0] * Ho * Wo * cout * (2 if has_res else 1))
    fl = 2.0 * shp[0] * Ho * Wo * cout * k * k * shp[3]
    total += t * len(units)
    print(f"{kind:8s} {str(shp):>22s} {cout:4d} {k} {stride} {int(has_res):3d} | {t:7.1f} x{len(units):<3d} {t*len(units):8.1f} | "
          f"{byts/t/1e3:6.0f} {fl/t/1e6:6.1f}  {units[0]}")
print(f"sum of convolutions: {total/1e3:.3f} ms per pass")
