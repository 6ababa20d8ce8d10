"""Histogram of the rf_gemm launches of one eager train step on the bench configuration (GPU box):
    python tools/gemm_hist.py"""
import collections, os, sys
import torch
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
hist = collections.Counter()
real = K.gemm
def spy(A, lda_m, lda_k, B, ldb_k, ldb_n, C, ldc, M, N, K_, **kw):
    sk = kw.get("splitk", 0) or K._auto_split(M, N, K_)
    hist[(M, N, K_, sk, "A^T" if lda_m == 1 else "A", "B^T" if ldb_n == 1 else "B", "atomic" if kw.get("atomic") else "",
          "act" if kw.get("act") else "", "res" if kw.get("residual") is not None else "")] += 1
    return real(A, lda_m, lda_k, B, ldb_k, ldb_n, C, ldc, M, N, K_, **kw)
K.gemm = spy
eng._fwd_bwd(item, 10)
torch.cuda.synchronize()
print(f"{sum(hist.values())} rf_gemm launches in one step")
for k, v in sorted(hist.items(), key=lambda kv: -kv[1]):
    print(v, k)
print(cfg.gps_backbone_config)
