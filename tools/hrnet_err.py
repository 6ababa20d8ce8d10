"""Error of the HIP conv trunk against the reference's golden features, per precision mode (GPU box only)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from routeformer_amd import kernels as Kn, synthetic
from routeformer_amd.models.video_backbone import HRNet16Backbone
G = np.load(os.path.join(ROOT, "tests", "golden", "hrnet.npz"))
for prec in ("f32", "bf16"):
    Kn.set_precision(prec)
    net = HRNet16Backbone()
    net.load_state_dict(synthetic.synth_state_dict(net.state_dict(), 7))
    net = net.to("cuda")
    for tag, n, hw in (("s64", 2, 64), ("s96", 1, 96), ("s224", 2, 224)):
        x = synthetic.synth_video(1, n, hw, hw, 11, "hrnet." + tag)[0].to("cuda")
        y = net(x).float().cpu()
        ref = torch.from_numpy(G[tag + ".y"])
        print(prec, tag, "rel err (max-norm)", float((y - ref).abs().max() / ref.abs().max()),
              "rel L2", float((y - ref).norm() / ref.norm()))
