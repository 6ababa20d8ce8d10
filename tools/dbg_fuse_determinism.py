import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from routeformer_amd import kernels as K, synthetic
from routeformer_amd.models.video_backbone import HRNet16Backbone
import routeformer_amd.models.video_backbone.hrnet16 as H
for prec in ("f32", "bf16"):
    K.set_precision(prec)
    net = HRNet16Backbone()
    net.load_state_dict(synthetic.synth_state_dict(net.state_dict(), 7))
    net = net.to("cuda")
    vids = [synthetic.synth_video(2, 40, 224, 224, 11 + i, "dbg")[0:2].to("cuda") for i in range(3)]
    idx = torch.flip(torch.arange(39, 0, -5), dims=[0])
    outs = {}
    for fs in (True, False, True):
        H.FUSE_SUM = fs
        t = net.encode_clips([(v, idx) for v in vids]).clone()
        outs.setdefault(fs, []).append(t)
    a, b = outs[True][0], outs[False][0]
    print(prec, "fused vs unfused max abs diff", float((a - b).abs().max()), "scale", float(b.abs().max()),
          "fused repeat identical", bool(torch.equal(outs[True][0], outs[True][1])))
# determinism of the fp32 train step gradients on c2_paper
from conftest import build_product_model, case_item, golden, RSEED
from test_gpu_model import _oracle_train_step_grads, _to_dev
from routeformer_amd.engine import train_step_losses
K.set_precision("f32")
model, cfg, sd, c = build_product_model("c2_paper", "cuda:0")
item = case_item(c)
item_d = {"train": _to_dev(item["train"]), "target": _to_dev(item["target"])}
_, og, tops = _oracle_train_step_grads(cfg, sd, item, 10)
grads = {}
for run, fs in enumerate((True, True, False)):
    H.FUSE_SUM = fs
    model.load_state_dict(sd); model.train(); model.zero_grad(set_to_none=True)
    K.TOPS.forced = [t.clone() for t in tops]
    torch.manual_seed(RSEED)
    res = train_step_losses(model, item_d, 10)
    K.TOPS.forced = None
    res["loss"].backward()
    torch.cuda.synchronize()
    grads[run] = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
    gmax = max(float(g.abs().max()) for g in og.values())
    rows = sorted(((float((grads[run][n].double() - go.double()).abs().max()) / max(float(go.abs().max()), 1e-3 * gmax), n) for n, go in og.items()), reverse=True)
    print("run", run, "fuse", fs, "loss", float(res["loss"]), "worst vs oracle:", [(round(e, 5), n[-60:]) for e, n in rows[:6]])
for a_, b_ in ((0, 1), (0, 2)):
    d = max(float((grads[a_][n] - grads[b_][n]).abs().max() / max(1e-12, float(grads[a_][n].abs().max()))) for n in grads[a_])
    print("runs", a_, b_, "max rel param-grad diff", d)
