"""Worker of tests/test_gpu_dp.py: one rank of a 2-rank data-parallel run that shares the single GPU of the
test box (process group over gloo; on a real node the same code runs over RCCL, one GPU per rank).
Each rank trains on ITS OWN batches for a few graph-replayed steps; afterwards every rank must hold the same
parameters, and they must equal what a single process gets from the mean of the two ranks' gradients."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from conftest import build_product_model
    from routeformer_amd import kernels as K, synthetic
    from routeformer_amd.engine import GraphedTrainEngine, TrainEngine
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    K.set_precision("f32")
    mode = sys.argv[1]  # "graph" | "graph_split" | "eager" | "eager_nooverlap" | "graph_split_direct" | "graph_direct_bf16"
    os.environ["RF_SPLIT_BWD"] = "1" if mode.startswith("graph_split") else "0"
    if "direct" in mode:  # direct reduce-scatter + sharded AdamW + parameter all-gather (engine.GradReducer modes)
        os.environ["RF_DP_MODE"] = "direct_bf16" if mode.endswith("bf16") else "direct"
    model, cfg, sd, c = build_product_model("c2_small", "cuda:0")
    model.train()

    def batch(step, r):
        item = synthetic.synth_item(c["B"], c["T"], c["P"], 100 + 10 * step + r, c["H"], c["W"], streams=c["streams"],
                                    gaze=c["gaze"])
        return {k: {n: v.cuda() for n, v in d.items()} for k, d in item.items()}

    if mode.startswith("eager"):
        eng = TrainEngine(model, lr=1e-3, overlap=mode != "eager_nooverlap")
    else:
        eng = GraphedTrainEngine(model, lr=1e-3)
    if not mode.startswith("eager"):
        eng.capture(batch(0, rank), epoch=10)  # warm-up passes consume host-RNG draws: capture before seeding
    torch.manual_seed(1234)  # same host-RNG stream on every rank and engine: same ProbSparse key samples
    for step in range(3):
        eng.step(batch(step, rank), epoch=10)
    torch.cuda.synchronize()
    flat = eng.reducer.flat_param.detach().cpu()
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    if rank == 0:
        params = torch.cat([p.detach().reshape(-1).cpu() for p in model.parameters()])
        torch.save({"same": same, "flat": flat, "params": params}, sys.argv[2])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
