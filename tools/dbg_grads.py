import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import RSEED, build_product_model, case_item, golden
from oracle import routeformer_oracle as O
from routeformer_amd import kernels as K
from routeformer_amd.engine import train_step_losses
DEV = "cuda:0"
name = "c2_paper"
model, cfg, sd, c = build_product_model(name, DEV)
G = golden(name); item = case_item(c)
to = lambda d: {k: v.to(DEV) for k, v in d.items()}
item_d = {"train": to(item["train"]), "target": to(item["target"])}
epoch = 10; key = "train10."
torch.manual_seed(RSEED)
orc = O.OracleRouteformer(cfg, sd, training=True)
with torch.no_grad(): orc.train_step(item, epoch)
model.load_state_dict(sd); model.train(); model.zero_grad(set_to_none=True)
K.TOPS.forced = [t_.clone() for t_ in orc.idx.tops]
torch.manual_seed(RSEED)
res = train_step_losses(model, item_d, epoch)
K.TOPS.forced = None
res["loss"].backward()
named = dict(model.named_parameters())
rows = []
for n, (nrm, _) in zip((str(s) for s in G[key + "grad_names"]), G[key + "grad_stats"]):
    g = named[n].grad; got = 0.0 if g is None else float(g.double().norm())
    rows.append((abs(got - nrm) / max(nrm, 1e-12), n, got, nrm))
mx = max(r[3] for r in rows)
rows = [r for r in rows if r[3] > 1e-3 * mx]
rows.sort(reverse=True)
for r in rows[:12]: print(f"{r[0]:.3e} {r[1]:70s} {r[2]:.6g} {r[3]:.6g}")
