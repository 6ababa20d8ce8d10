"""Debug: compare per-layer activations of the HIP product (GPU) with the CPU oracle, selections forced."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import build_product_model, case_item, draws, golden, rel_err, RSEED
from oracle import routeformer_oracle as O
from routeformer_amd import kernels as K

name = sys.argv[1] if len(sys.argv) > 1 else "c2_paper"
forced = (sys.argv[2] == "forced") if len(sys.argv) > 2 else True
model, cfg, sd, c = build_product_model(name, "cuda:0")
G = golden(name)
item = case_item(c)
O.TAPS = {}
src = O.IndexSource(draws(G, "eval."))
with torch.no_grad():
    out_o = O.OracleRouteformer(cfg, sd, training=False, idx=src).forward(item["train"])
taps_o = O.TAPS; O.TAPS = None
taps_d = {}
def hook(nm):
    def f(mod, inp, out):
        taps_d.setdefault(nm, []).append(out.detach().cpu())
    return f
for nm, mod in model.named_modules():
    if nm.endswith(tuple(f"attn_layers.{i}" for i in range(10))) or nm.endswith(tuple(f"decoder.layers.{i}" for i in range(4))) \
       or nm.endswith("conv_layers.0") or nm in ("frame_encoder", "video_encoder", "gaze_encoder", "gaze_video_decoder", "gps_backbone"):
        mod.register_forward_hook(hook(nm))
model.eval()
if forced:
    K.TOPS.forced = [t.clone() for t in src.tops]
torch.manual_seed(RSEED)
with torch.no_grad():
    out_d = model({k: v.cuda() for k, v in item["train"].items()})
print("final", rel_err(out_d[0], out_o[0]) if isinstance(out_d, tuple) else rel_err(out_d, out_o))
for nm, lst in taps_d.items():
    ref = taps_o.get(nm)
    if ref is None:
        print("no oracle tap for", nm); continue
    for i, (a, b) in enumerate(zip(lst, ref)):
        print(f"{nm}[{i}] shape {tuple(a.shape)} rel_err {rel_err(a, b):.3e}  max|ref| {float(b.abs().max()):.3g}")
