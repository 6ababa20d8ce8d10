"""One C2 train step (eager engine, same seeds) with and without the slab-carried FFN input gradients (kernels.LAZY_DX):
the whole gradient buffer must agree to the noise of the fp32 atomics."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from routeformer_amd import kernels as K
from routeformer_amd.engine import TrainEngine
from routeformer_amd.models.blocks import SAMPLER
out = {}
for lazy in (False, True, False):
    K.LAZY_DX = lazy
    torch.manual_seed(0)
    model, cfg, sd, c = bench.build("C2", "cuda", "bf16")
    item = bench.make_item(c, 0, "cuda")
    eng = TrainEngine(model)
    torch.manual_seed(1)
    res = eng._fwd_bwd(item, 10)
    torch.cuda.synchronize()
    g = eng.reducer.flat_grad.clone()
    out.setdefault(lazy, []).append((float(res["loss"]), g))
    SAMPLER.drop_static()
(l0, g0), (l0b, g0b) = out[False]
(l1, g1), = out[True]
rel = lambda a, b: float((a - b).norm() / b.norm())
print(f"loss off {l0:.6f} / {l0b:.6f}  on {l1:.6f}")
print(f"gradient buffer: off vs off (run-to-run noise) {rel(g0b, g0):.3e}   on vs off {rel(g1, g0):.3e}   max abs diff {float((g1 - g0).abs().max()):.3e} (|g| max {float(g0.abs().max()):.3e})")
