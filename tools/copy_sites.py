"""Where do the device-to-device copies of a train step come from?  (GPU box)  One eager step of the bench model under
torch.profiler with Python stacks; prints the call sites of aten::copy_ / aten::contiguous / aten::clone / fill_."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from routeformer_amd import kernels as K
from routeformer_amd.engine import TrainEngine
dev = torch.device("cuda", 0)
model, cfg, sd, c = bench.build("C2", dev, "bf16")
item = bench.make_item(c, 0, dev)
eng = TrainEngine(model)
for _ in range(2):
    eng._fwd_bwd(item, 10)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    eng._fwd_bwd(item, 10)
    torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::cat", "aten::index_select", "aten::index_put_", "aten::add", "aten::mul"):
        st = [s for s in ev.stack if "routeformer_amd" in s or "bench.py" in s]
        key = (ev.name, st[0].strip() if st else "(torch internal)", str(ev.input_shapes)[:60])
        sites[key] += 1
for (name, site, shp), n in sites.most_common(60):
    print(f"{n:4d} {name:18s} {site[-90:]:90s} {shp}")
