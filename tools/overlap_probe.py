"""How much does running the conv trunk of the next batch underneath the step cost the step's own chain?
Times, on the bench workload: (a) the pipelined step (trunk as a parallel branch of the replayed graph), (b) the same
step with the trunk replayed BEFORE it (serial), (c) the trunk graph alone.  (b) - (c) = the chain on an otherwise
idle chip; (a) - that = what the overlap really costs.  GPU box only:  python tools/overlap_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from routeformer_amd.engine import GraphedTrainEngine  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
model, cfg, sd, c = bench.build("C2", dev, "bf16")
items = [bench.make_item(c, 0, dev), bench.make_item(c, 500, dev)]
eng = GraphedTrainEngine(model)
eng.capture(items[0], epoch=10)


def timed(fn, n=20):
    for i in range(4):  # both look-ahead graphs captured before the clock starts
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


a = timed(lambda i: eng.step(items[i % 2], epoch=10, next_item=items[(i + 1) % 2]))
b = timed(lambda i: eng.step(items[i % 2], epoch=10, next_item=None))
g = eng._trunk_graph(items[0])
cc = timed(lambda i: g.replay())
print(f"(a) pipelined step            {a:7.3f} ms")
print(f"(b) trunk, then step (serial) {b:7.3f} ms")
print(f"(c) trunk alone               {cc:7.3f} ms")
print(f"step chain on an idle chip (b - c) {b - cc:7.3f} ms; cost of the overlap (a - (b - c)) {a - (b - cc):7.3f} ms")
