"""What does a second active branch cost the step's chain: dispatch or contention?  Replaces the conv trunk inside the
replayed graph by (i) 200 trivial kernels (1 KB each: no work, only dispatches), (ii) ONE streaming kernel of
about the trunk's duration (HBM traffic, every CU busy), (iii) nothing, and times the step.  GPU box only."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from routeformer_amd import _hip, kernels as K
from routeformer_amd.engine import GraphedTrainEngine
from routeformer_amd.models.blocks import SAMPLER

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
a = torch.randn(256, device=dev); b = torch.randn(256, device=dev); o = torch.empty(256, device=dev)
big_a = torch.randn(1 << 30, device=dev, dtype=torch.bfloat16); big_o = torch.empty_like(big_a)   # 2 GiB each


def tiny(n):
    for _ in range(n):
        _hip.check(_hip.lib().rf_add_relu(a.data_ptr(), b.data_ptr(), o.data_ptr(), 0, 256, 1, K._stream()), "add_relu")


def streaming(reps):
    for _ in range(reps):
        _hip.check(_hip.lib().rf_add_relu(big_a.data_ptr(), big_a.data_ptr(), big_o.data_ptr(), 1, big_a.numel(), 1, K._stream()),
                   "add_relu")


def run(label, stand_in):
    model, cfg, sd, c = bench.build("C2", dev, "bf16")
    items = [bench.make_item(c, 0, dev), bench.make_item(c, 500, dev)]
    eng = GraphedTrainEngine(model)
    eng.capture(items[0], epoch=10)
    if stand_in is not None:
        real = model.video_backbone.encode_clips
        def fake(clips, out=None):
            if out is None:
                return real(clips)
            stand_in()
            return out
        model.video_backbone.encode_clips = fake
        eng._trunk_g = None  # the trunk graph and the look-ahead variants of the step graph are captured again
        eng._graphs = {k: v for k, v in eng._graphs.items() if not k[0]}
    for i in range(4):
        eng.step(items[i % 2], epoch=10, next_item=items[(i + 1) % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        eng.step(items[i % 2], epoch=10, next_item=items[(i + 1) % 2])
    torch.cuda.synchronize()
    print(f"{label:58s} {(time.perf_counter() - t0) / 20 * 1e3:7.3f} ms/step", flush=True)
    SAMPLER.drop_static()


streaming(1); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); streaming(1); e.record(); torch.cuda.synchronize()
print(f"(one streaming pass over 6 GB: {s.elapsed_time(e):.2f} ms)")
run("real conv trunk as the second branch", None)
run("second branch = 200 trivial kernels (dispatch only)", lambda: tiny(200))
run("second branch = 1000 trivial kernels", lambda: tiny(1000))
run("second branch = one streaming kernel (~trunk duration)", lambda: streaming(2))
run("second branch empty", lambda: None)
