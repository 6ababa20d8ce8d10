"""Bootstrap that makes the *reference's own* hot-path modules importable in the
build container (CPU only), WITHOUT copying any of its source.

Used only by ``tests/golden/make_golden.py`` (fixture generation) -- never by the
product, never on the GPU box (``/root/reference`` does not exist there).

Recipe = SURVEY.md Appendix C: stub ``lightning``, register empty namespace packages so the
heavy ``__init__``s (datasets, timm, wandb ...) are skipped, then import the pure-torch files.
"""
import importlib
import importlib.util
import sys
import types

REF = "/root/reference"


def bootstrap():
    sys.dont_write_bytecode = True
    import torch
    from torch import nn

    if "lightning" not in sys.modules:
        L = types.ModuleType("lightning")

        class LightningModule(nn.Module):
            current_epoch = 0

            @property
            def device(self):
                try:
                    return next(self.parameters()).device
                except StopIteration:
                    return torch.device("cpu")

        L.LightningModule = LightningModule
        sys.modules["lightning"] = L

    def ns(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
        return m

    ns("routeformer", f"{REF}/routeformer")
    ns("routeformer.io", f"{REF}/routeformer/io")
    ns("routeformer.utils", f"{REF}/routeformer/utils")
    ns("routeformer.models", f"{REF}/routeformer/models")
    vb = ns("routeformer.models.video_backbone", f"{REF}/routeformer/models/video_backbone")
    ds = types.ModuleType("routeformer.io.dataset")
    ds.Data = dict
    ds.Item = dict
    sys.modules["routeformer.io.dataset"] = ds

    vbc = importlib.import_module("routeformer.models.video_backbone.config")
    for k in ("VideoBackboneConfig", "VideoBackboneModule", "TimmBackboneConfig",
              "InverseFormBackboneConfig"):
        setattr(vb, k, getattr(vbc, k))

    gps = importlib.import_module("routeformer.models.gps_backbone")
    rf = importlib.import_module("routeformer.models.routeformer")
    cfg = importlib.import_module("routeformer.models.config")
    cmt = importlib.import_module("routeformer.models.cross_modal_transformer")

    def by_path(name, rel):
        spec = importlib.util.spec_from_file_location(name, f"{REF}/{rel}")
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    loss = by_path("_ref_loss", "routeformer/losses/future_discounted_mse.py")
    score = by_path("_ref_score", "routeformer/score/error.py")
    vec = importlib.import_module("routeformer.utils.vector")
    flt = importlib.import_module("routeformer.utils.filter")

    out = types.SimpleNamespace(
        torch=torch, gps=gps, rf=rf, cfg=cfg, cmt=cmt, loss=loss, score=score, vec=vec,
        flt=flt, vbc=vbc,
    )
    return out


def build_hrnet16(ref):
    """The frozen conv encoder exactly as InverseForm.py:51-71 builds it, minus the checkpoint
    download: HRNet16 trunk + AdaptiveAvgPool2d((8, 8)), eval, no grad."""
    import torch
    from torch import nn

    icfg = importlib.import_module(
        "routeformer.models.video_backbone.inverse_form_layers.config")
    icfg.assert_and_infer_cfg(result_dir=None, global_rank=None, apex=False, syncbn=False,
                              arch="lighthrnet.HRNet16", hrnet_base=16, fp16=False,
                              has_edge=True)
    hr = importlib.import_module(
        "routeformer.models.video_backbone.inverse_form_layers.hrnetv2")

    class RefHRNet16(ref.vbc.VideoBackboneModule):
        def __init__(self, configs=None):
            super().__init__()
            self.configs = configs
            self._Backbone = hr.get_seg_model()
            self.adaptive_pool = nn.AdaptiveAvgPool2d((8, 8))
            self._Backbone.eval()
            self._Backbone.requires_grad_(False)

        @property
        def output_feature_shape(self):
            return (240, 8, 8)

        def forward(self, images):
            images = images.to(torch.float32)
            return self.adaptive_pool(self._Backbone(images)[-1])

        def train(self, mode=True):  # frozen: BN stays in eval mode (InverseForm.py:69-71)
            super().train(mode)
            self._Backbone.eval()
            return self

    return RefHRNet16
