"""Which ATen operators (elementwise / fill / copy / index kernels -- everything that is not a librf_hip.so launch) does one
train step run, and from which line of routeformer_amd?  TorchDispatchMode + the Python stack at dispatch time (works for
custom-Function backward code too).  GPU box:
    python tools/copy_sites.py"""
import collections, os, sys, traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from routeformer_amd import kernels as K, presets, synthetic
from routeformer_amd.engine import TrainEngine
from routeformer_amd.models import Routeformer, RouteformerConfig
from routeformer_amd.models.gps_backbone import GPSBackboneConfig, Informer
from routeformer_amd.models.video_backbone import HRNet16Backbone, VideoBackboneConfig
K.set_precision("bf16")
c = presets.case("C2")
_, cfg = presets.build_configs(c, GPSBackboneConfig, RouteformerConfig, VideoBackboneConfig)
model = Routeformer(cfg, gps_backbone=Informer, video_backbone=HRNet16Backbone).to("cuda")
it = synthetic.synth_item(c["B"], c["T"], c["P"], 1, c["H"], c["W"], streams=c["streams"], gaze=c["gaze"])
item = {p: {k: v.to("cuda") for k, v in it[p].items()} for p in ("train", "target")}
eng = TrainEngine(model)
model.train()
eng._fwd_bwd(item, 10)
torch.cuda.synchronize()
VIEW = ("view", "reshape", "alias", "detach", "t.default", "transpose", "permute", "expand", "slice", "select", "unsqueeze",
        "squeeze", "as_strided", "_unsafe_view", "empty", "unbind", "split", "_local_scalar", "is_", "sym_", "stride", "size",
        "lift_fresh", "narrow", "unfold", "chunk", "_reshape_alias", "record_stream", "contiguous")
hist = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        out = func(*args, **(kwargs or {}))
        if not any(v in name for v in VIEW):
            site = "?"
            for fr in reversed(traceback.extract_stack()):
                if "routeformer_amd" in fr.filename and "copy_sites" not in fr.filename:
                    site = f"{os.path.relpath(fr.filename, root)}:{fr.lineno} {fr.name}"
                    break
            shp = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:2]
            hist[(name, site, str(shp))] += 1
        return out


with Spy():
    eng._fwd_bwd(item, 10)
torch.cuda.synchronize()
print(f"{sum(hist.values())} device-side ATen calls in one eager step (views / allocations excluded)")
for (name, site, shp), n in sorted(hist.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d}  {name:38s} {site:70s} {shp}")
