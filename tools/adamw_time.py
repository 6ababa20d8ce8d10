"""Time of the fused clip + AdamW launch on the bench's parameter count (GPU box only)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from routeformer_amd import _hip, kernels as K
n = 75_300_000
p, g, m, v = (torch.randn(n, device="cuda") for _ in range(4))
v.abs_()
parts = int(_hip.lib().rf_sumsq_parts(n)); ss = torch.zeros(parts, device="cuda")
def step(t):
    _hip.check(_hip.lib().rf_sumsq(g.data_ptr(), n, ss.data_ptr(), K._stream()), "sumsq")
    _hip.check(_hip.lib().rf_adamw_clip(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, ss.data_ptr(), parts, 2.5, 1e-5, 0.9, 0.999,
                                        1e-8, 1e-4, t, 1.0, K._stream()), "adamw")
for t in range(1, 4): step(t)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for t in range(4, 24): step(t)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 20
print(f"sumsq + adamw: {ms*1e3:.1f} us per step; {(28 + 4) * n / ms / 1e9:.2f} TB/s")
