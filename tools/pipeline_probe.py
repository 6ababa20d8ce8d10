import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from routeformer_amd.engine import GraphedTrainEngine
from routeformer_amd.models.blocks import SAMPLER
dev = torch.device("cuda", 0)
model, cfg, sd, c = bench.build("C2", dev, "bf16")
items = [bench.make_item(c, 0, dev), bench.make_item(c, 500, dev)]
eng = GraphedTrainEngine(model)
eng.capture(items[0], epoch=10)
for i in range(3):
    eng.step(items[i % 2], epoch=10, next_item=items[(i + 1) % 2])
torch.cuda.synchronize()
gA = eng._trunk_graph(items[0]); gB = eng._trunk_graph(items[1])
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def main_only():
    SAMPLER.refill_static(); eng.graph.replay()
def trunk_only():
    gA.replay()
def both():
    SAMPLER.refill_static()
    eng._tstream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(eng._tstream):
        gB.replay()
    eng.graph.replay()
    torch.cuda.current_stream().wait_stream(eng._tstream)
def opt_only():
    eng.opt.step(1.0)
print("main graph only  ms", t(main_only))
print("trunk graph only ms", t(trunk_only))
print("both concurrent  ms", t(both))
print("optimizer only   ms", t(opt_only))
print("host refill only ms", t(lambda: SAMPLER.refill_static()))
