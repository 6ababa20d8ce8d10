"""Gradients (and a few activations) of one c1_paper train step through plain autograd, saved to a file: run under two
builds of the library (RF_HIP_LIB) and compare:  python tools/probes/c1_grads.py out.pt [other.pt]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import build_product_model, case_item
from routeformer_amd.engine import train_step_losses
from routeformer_amd import kernels as K
dev = "cuda"
if len(sys.argv) > 2:
    a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
    gmax = max(float(v.abs().max()) for n, v in b.items() if not n.startswith("act::") and n != "loss")
    rows = sorted(((float((a[n] - b[n]).abs().max() / max(float(b[n].abs().max()), 1e-3 * gmax if not n.startswith("act::") else 1e-12)), n)
                   for n in a), reverse=True)
    for e, n in rows[:3]:
        print(f"{e:10.3e}  {n}")
    for n in ("gps_backbone.encoder.attn_layers.5.conv1.weight", "gps_backbone.encoder.attn_layers.5.conv2.weight",
              "gps_backbone.encoder.attn_layers.4.conv1.weight", "gps_backbone.encoder.attn_layers.0.conv1.weight",
              "gps_backbone.encoder.norm.weight", "gps_backbone.decoder.layers.0.cross_attention.key_projection.weight",
              "gps_backbone.decoder.layers.0.cross_attention.value_projection.weight"):
        x, y = a[n].double().reshape(-1), b[n].double().reshape(-1)
        print(f"cos {float(x @ y / (x.norm() * y.norm())):.6f}  norm ratio {float(x.norm() / y.norm()):.6f}  {n}")
    for n in sorted(a):
        if n.startswith("act::"):
            print(f"{float((a[n] - b[n]).abs().max() / b[n].abs().max()):10.3e}  {n}  shape {tuple(a[n].shape)}")
    sys.exit(0)
model, cfg, sd, c = build_product_model("c1_paper", dev)
item = case_item(c)
item_d = {p: {k: v.to(dev) for k, v in item[p].items()} for p in ("train", "target")}
model.train()
acts = {}
def hook(name):
    def f(m, i, o):
        acts["act::" + name] = (o[0] if isinstance(o, tuple) else o).detach().float().cpu().clone()
    return f
for n, m in model.named_modules():
    if n.endswith("conv_layers.4") or n.endswith("conv_layers.3") or n.endswith("attn_layers.5") or n.endswith("attn_layers.4"):
        m.register_forward_hook(hook(n))
torch.manual_seed(1234)
res = train_step_losses(model, item_d, 0)
res["loss"].backward()
torch.cuda.synchronize()
out = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None}
out.update(acts)
out["loss"] = res["loss"].detach().float().cpu().reshape(1)
torch.save(out, sys.argv[1])
print("saved", len(out), float(res["loss"]))
