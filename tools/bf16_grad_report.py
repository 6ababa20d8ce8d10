#!/usr/bin/env python3
"""Per-parameter agreement of the product's bf16-mode train-step gradients with the CPU oracle's autograd gradients
(oracle selections imposed), grouped by module -- the bisecting aid behind tests/test_gpu_model.py::
test_model_train_step_bf16.  Kernel paths are switched with the usual environment variables (RF_SEQSTACK,
RF_SEQSTACK_BWD, RF_ROWBLOCK, RF_SKINNY, RF_WGRAD_TR, RF_GROUP_WGRAD); --precision f32 gives the fp32 floor.

    python tools/bf16_grad_report.py c2_paper [--precision bf16] [--top 12]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("case")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--top", type=int, default=12)
    ap.add_argument("--epoch", type=int, default=10)
    args = ap.parse_args()
    from conftest import RSEED, build_product_model, case_item
    from test_gpu_model import _oracle_train_step_grads, _to_dev
    from routeformer_amd import kernels as K
    from routeformer_amd.engine import TrainEngine
    model, cfg, sd, c = build_product_model(args.case, "cuda:0")
    item = case_item(c)
    item_d = {"train": _to_dev(item["train"]), "target": _to_dev(item["target"])}
    _, og, tops = _oracle_train_step_grads(cfg, sd, item, args.epoch)
    K.set_precision(args.precision)
    eng = TrainEngine(model)
    model.train()
    K.TOPS.forced = [t.clone() for t in tops]
    torch.manual_seed(RSEED)
    eng._fwd_bwd(item_d, args.epoch)
    torch.cuda.synchronize()
    K.TOPS.forced = None
    named = dict(model.named_parameters())
    rows, groups = [], {}
    for n, go in og.items():
        go = go.double().reshape(-1)
        g = named[n].grad.detach().cpu().double().reshape(-1)
        rows.append((n, float(g @ go), float(g @ g), float(go @ go)))
        grp = n.split(".")[0] + ("." + n.split(".")[1] if n.startswith("gps_backbone.") else "")
        a = groups.setdefault(grp, [0.0, 0.0, 0.0])
        a[0] += rows[-1][1]; a[1] += rows[-1][2]; a[2] += rows[-1][3]
    tot = [sum(r[i] for r in rows) for i in (1, 2, 3)]
    env = {k: v for k, v in os.environ.items() if k.startswith("RF_")}
    print(f"== {args.case} {args.precision} {env}: whole cosine {tot[0] / (tot[1] * tot[2]) ** 0.5:.5f} "
          f"norm ratio {(tot[1] / tot[2]) ** 0.5:.4f}")
    for grp, (d, gg, oo) in sorted(groups.items()):
        print(f"   {grp:42s} cos {d / max((gg * oo) ** 0.5, 1e-300):.5f}  norm ratio {(gg / max(oo, 1e-300)) ** 0.5:.4f}  "
              f"share of |grad|^2 {oo / tot[2]:.3e}")
    gmax = max(r[3] for r in rows) ** 0.5
    worst = sorted((r for r in rows if r[3] ** 0.5 > 1e-3 * gmax), key=lambda r: r[1] / max((r[2] * r[3]) ** 0.5, 1e-300))
    for n, d, gg, oo in worst[:args.top]:
        print(f"   worst: {n:80s} cos {d / (gg * oo) ** 0.5:.4f} norm ratio {(gg / oo) ** 0.5:.3f} |g_ref| {oo ** 0.5:.3e}")


if __name__ == "__main__":
    main()
