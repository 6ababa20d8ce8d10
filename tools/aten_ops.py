"""Which ATen ops (not librf_hip kernels) still run inside one train step?  Input shapes + python call
sites from torch.profiler, eager TrainEngine step of the bench workload.  GPU box only."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from routeformer_amd.engine import TrainEngine
case = sys.argv[1] if len(sys.argv) > 1 else "C2"
model, cfg, sd, c = bench.build(case, "cuda", "bf16")
item = bench.make_item(c, 0, "cuda")
eng = TrainEngine(model)
for _ in range(2):
    eng.step(item, epoch=10)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    eng.step(item, epoch=10)
    torch.cuda.synchronize()
tot = collections.Counter(); ex = {}
for ev in prof.events():
    leaf = not any(ch.name.startswith("aten::") for ch in (ev.cpu_children or []))
    launches = any("hipLaunchKernel" in ch.name or "hipModuleLaunch" in ch.name or "Memcpy" in ch.name or "Memset" in ch.name
                   for ch in (ev.cpu_children or []))
    if ev.name.startswith("aten::") and leaf and launches:
        stack = [s for s in (ev.stack or []) if "routeformer_amd" in s or "bench" in s]
        key = (ev.name, str(ev.input_shapes)[:70], stack[0].split("/")[-1][:60] if stack else "?")
        tot[key] += 1
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:60]:
    print(v, *k, sep=" | ")
