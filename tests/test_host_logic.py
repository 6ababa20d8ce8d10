"""CPU: host-side logic of the drop-in boundary -- configs, state_dict layout, losses / scores /
helpers against the reference's golden vectors, synthetic-data determinism, sampler order."""
import numpy as np
import pytest
import torch

from conftest import build_product_model, golden, rel_err, t

ALL_CASES = ["c1_default", "c1_paper", "c1_recursive", "c2_small", "c4_small", "c5_small", "ar_small", "c2_paper"]


@pytest.mark.parametrize("name", ALL_CASES)
def test_state_dict_layout_matches_reference(name):
    """Same keys, shapes and (synthetic) values as the reference model: the digest covers key names."""
    from routeformer_amd import synthetic
    model, cfg, sd, c = build_product_model(name)
    G = golden(name)
    assert sum(p.numel() for p in model.parameters()) == int(G["n_params"])
    assert abs(synthetic.state_dict_digest(sd) - float(G["digest"])) < 1e-6 * float(G["digest"])
    frozen = [n for n, p in model.named_parameters() if not p.requires_grad]
    assert all("video_backbone" in n for n in frozen)


def test_losses_scores_helpers_golden():
    from routeformer_amd.losses.future_discounted_mse import FutureDiscountedLoss
    from routeformer_amd.score import ade, fde
    from routeformer_amd.utils import estimate_angle_and_norm, median_downsampler, rotate
    G = golden("helpers")
    pred, true = t(G["pred"]), t(G["true"])
    for kind in ("mse", "mae", "smooth_l1"):
        loss = FutureDiscountedLoss({0: 0.97, 100: 0.98}, 1.0, loss_function=kind)
        assert abs(float(loss(pred, true)) - float(G["loss." + kind])) < 1e-6
    dense = FutureDiscountedLoss(0.9, 0.3, loss_function="smooth_l1")
    assert abs(float(dense(t(G["feat_p"]), t(G["feat_t"]))) - float(G["loss.dense"])) < 1e-6
    assert abs(float(ade(pred, true)) - float(G["ade"])) < 1e-6
    assert abs(float(fde(pred, true)) - float(G["fde"])) < 1e-6
    assert torch.equal(median_downsampler(t(G["gaze"]), 40), t(G["gaze_ds40"]))
    assert torch.equal(median_downsampler(t(G["gaze"])[:, :100], 7), t(G["gaze_ds7"]))
    assert rel_err(rotate(t(G["v"]), t(G["ang"])), G["rot"]) < 1e-6
    a, n = estimate_angle_and_norm(t(G["v"]))
    assert rel_err(a, G["angle"]) < 1e-6 and rel_err(n, G["norm"]) < 1e-6


def test_loss_quirks():
    from routeformer_amd.losses import FutureDiscountedLoss
    from routeformer_amd.score import ade
    from routeformer_amd.utils import median_downsampler
    with pytest.raises(ValueError):
        FutureDiscountedLoss(0.9, 1.0, loss_function="huber")
    with pytest.raises(TypeError):  # epsilon=None is evaluated even on the smooth-L1 path
        FutureDiscountedLoss(0.9, None, loss_function="smooth_l1")(torch.zeros(1, 2, 2), torch.zeros(1, 2, 2))
    loss = FutureDiscountedLoss({0: 0.5, 3: 1.0}, 1.0, loss_function="mse")
    x, y = torch.ones(1, 4, 2), torch.zeros(1, 4, 2)
    x = x * 2
    assert abs(float(loss(x, y)) - 4 * (1 + .5 + .25 + .125) / 4) < 1e-6
    loss.current_epoch = 3
    assert abs(float(loss(x, y)) - 4.0) < 1e-6
    loss.current_epoch = 4  # sticky: stays at the last switched value
    assert abs(float(loss(x, y)) - 4.0) < 1e-6
    with pytest.raises(AssertionError):
        ade(torch.zeros(2, 3, 2), torch.zeros(2, 4, 2))
    with pytest.raises(ValueError):
        median_downsampler(torch.zeros(1, 5, 2), 5)


def test_configs():
    from routeformer_amd.models import Routeformer, RouteformerConfig
    from routeformer_amd.models.gps_backbone import GPSBackboneConfig, Informer
    from routeformer_amd.models.video_backbone import VideoBackboneConfig
    g = GPSBackboneConfig(seq_len=40, label_len=40, pred_len=30)
    r = RouteformerConfig(gps_backbone_config=g)
    assert r.with_video is False and g.enc_in == 5 and g.c_out == 2 and g.smart_decoder is False
    r2 = r.override(video_backbone_config=VideoBackboneConfig(torchcache_enabled=False), with_video=True,
                    dense_prediction=True, decoder_mode="smart", encoder_hidden_size=64)
    g2 = r2.gps_backbone_config
    assert g2.enc_in == 69 and g2.c_out == 66 and g2.dec_in == 69 and g2.smart_decoder is True
    assert r.gps_backbone_config.enc_in == 5, "override must deep-copy"
    assert r2["encoder_heads"] == 8 and r2.get("nope", 3) == 3
    with pytest.raises(AssertionError):
        RouteformerConfig(gps_backbone_config=g, video_fps=2)
    with pytest.raises(AssertionError):
        RouteformerConfig(gps_backbone_config=g, with_gaze=True, with_video=False)
    with pytest.raises(ValueError):
        VideoBackboneConfig(torchcache_enabled=True, train_backbone=True)
    m = Routeformer(r, gps_backbone=Informer)
    m.configs.rotate_motion = True
    assert r.rotate_motion is False, "model must copy its config"


def test_hrnet_units_and_keys():
    from routeformer_amd.models.video_backbone import HRNet16Backbone
    from routeformer_amd.models.video_backbone.hrnet16 import UNITS
    assert len(UNITS) == 145  # 145 Conv2d
    net = HRNet16Backbone()
    assert len(net.state_dict()) == 865 and net.output_feature_shape == (240, 8, 8)
    assert sum(p.numel() for p in net.parameters()) == 3102788
    net.train()
    assert all(not p.requires_grad for p in net.parameters())


def test_synthetic_is_deterministic_and_gem_shaped():
    from routeformer_amd import synthetic
    a = synthetic.synth_item(2, 40, 30, 5, 32, 32)
    b = synthetic.synth_item(2, 40, 30, 5, 32, 32)
    for part in ("train", "target"):
        for k in a[part]:
            assert torch.equal(a[part][k], b[part][k])
    tr = a["train"]
    assert tr["gps"].shape == (2, 40, 2) and tr["left_video"].dtype == torch.float16
    assert tr["left_video"].shape == (2, 40, 3, 32, 32) and tr["gaze"].shape == (2, 1600, 2)
    assert a["target"]["gaze"].shape == (2, 1200, 2)
    assert torch.allclose(a["target"]["gps"][:, 0] - tr["gps"][:, -1], a["target"]["gps"][:, 0] - tr["gps"][:, -1])
    t1 = synthetic.synth_tensor("a.weight", torch.empty(4, 3), 1)
    t2 = synthetic.synth_tensor("a.weight", torch.empty(4, 3), 1)
    assert torch.equal(t1, t2) and not torch.equal(t1, synthetic.synth_tensor("b.weight", torch.empty(4, 3), 1))


def test_index_sampler_and_prob_sizes():
    from routeformer_amd import kernels as K
    from routeformer_amd.models.blocks import IndexSampler
    # (L_Q, L_K, factor) -> (sample_k, n_top): SURVEY Appendix B
    assert K.prob_sizes(65, 65, 5) == (25, 25) and K.prob_sizes(160, 160, 5) == (30, 30)
    assert K.prob_sizes(40, 40, 4) == (16, 16) and K.prob_sizes(70, 4, 4) == (4, 20)
    assert K.prob_sizes(5, 5, 4) == (5, 5) and K.prob_sizes(25, 6, 1) == (2, 4)
    s = IndexSampler()
    s.log = []
    torch.manual_seed(3)
    a = s.draw(65, 65, 25, "cpu")
    torch.manual_seed(3)
    ref = torch.randint(65, (65, 25))
    assert a.dtype == torch.int32 and torch.equal(a.long(), ref) and torch.equal(s.log[0], ref)
    s.replay = [ref[:40, :16] % 40]
    assert torch.equal(s.draw(40, 40, 16, "cpu").long(), ref[:40, :16] % 40)


def test_product_refuses_cpu_execution():
    """No CPU fallback: running the product model on CPU tensors raises instead of silently computing."""
    from routeformer_amd import _hip
    model, cfg, sd, c = build_product_model("c1_default")
    with pytest.raises(_hip.HipLibraryError):
        model({"gps": torch.zeros(4, 10, 2)})


def test_split_k_heuristics():
    """Split-K choices for the step's small-M GEMMs (tools/splitk_sweep.py picked them on the GPU): no split once the
    tiles alone reach ~half the chip, ~256 workgroups for a one-d_model-deep reduction, ~512 for a deep one."""
    from routeformer_amd import kernels as K
    assert K._auto_split(12480, 128, 128) == 1            # plenty of tiles
    assert K._auto_split(320, 2496, 832) == 1             # 195 tiles
    assert K._auto_split(320, 832, 832) == 3              # 65 tiles, shallow
    assert K._auto_split(320, 832, 3328) == 8             # 65 tiles, deep
    assert K._auto_split(40, 2496, 832) == 6              # capped by >= 128 of depth per slice
    assert K._auto_split(560, 66, 832) in (6,)            # skinny output
    assert all(1 <= K._auto_split(m, n, k) <= 16 for m in (5, 40, 320, 560) for n in (64, 832, 3328) for k in (128, 832, 3328))
    assert K._splits(4, 12480) == 64 and K._splits(169, 560) == 4 and K._splits(1000, 560) == 1


def test_fused_adamw_hyper_vector():
    """The device-side hyper-parameter vector of the graph-replayed update (rf_adamw_clip_dev layout)."""
    import torch
    from routeformer_amd.engine import FusedAdamW
    opt = FusedAdamW.__new__(FusedAdamW)
    opt.betas, opt.eps, opt.wd, opt.max_norm, opt.param_groups, opt.t = (0.9, 0.999), 1e-8, 1e-4, 2.5, [{"lr": 3e-4}], 7
    h = opt.hyper(0.125)
    assert len(h) == 10 and h[0] == 1.0 and h[1] == 2.5 and h[2] == 3e-4 and h[9] == 0.125
    assert abs(h[7] - (1 - 0.9 ** 7)) < 1e-12 and abs(h[8] - (1 - 0.999 ** 7) ** 0.5) < 1e-12
    assert opt.hyper(1.0, pending=False)[0] == 0.0


def test_deferred_update_segments_and_pending_rows():
    """Host side of the deferred optimizer update with skippable ranges (engine.GraphedTrainEngine._plan_segments /
    _pending_rows): the flat buffers are cut at the GPS backbone's range (side stream) and at the gaze slots; a segment the
    previous step skipped gets pending = 0, every other one the bias corrections of ITS OWN update count (torch.optim.AdamW
    keeps state['step'] per parameter and advances it only for parameters that had a gradient)."""
    from types import SimpleNamespace
    from routeformer_amd.engine import FusedAdamW, GraphedTrainEngine, LagMap
    eng = GraphedTrainEngine.__new__(GraphedTrainEngine)
    opt = FusedAdamW.__new__(FusedAdamW)
    opt.betas, opt.eps, opt.wd, opt.max_norm, opt.param_groups, opt.t, opt._lag = (0.9, 0.999), 1e-8, 1e-4, 2.5, [{"lr": 1e-4}], 0, LagMap()
    eng.opt = opt
    eng.reducer = SimpleNamespace(flat_param=torch.zeros(1000))
    eng.model = SimpleNamespace(configs=SimpleNamespace(gaze_dropout=0.2), with_gaze=True)
    eng._gps_range = (0, 600)
    gaze = ((640, 704), (832, 896))
    eng._prefix_ranges = lambda prefixes: gaze
    segs = eng._plan_segments()
    assert segs == [(0, 600, True), (600, 640, False), (640, 704, False), (704, 832, False), (832, 896, False), (896, 1000, False)]
    eng._segments = segs
    # step 1 keeps the gaze branch, steps 2 and 3 drop it, step 4 keeps it again
    for skip in ((), gaze, gaze, ()):
        opt.t += 1
        opt._note_skipped(skip)
        rows = eng._pending_rows(0.5, skip)
        for (a, b, _), row in zip(segs, rows):
            is_gaze = (a, b) in gaze
            assert row[0] == (0.0 if (is_gaze and skip) else 1.0) and row[9] == 0.5
            t_seg = opt.t - (opt._lag.lag(a, b) if is_gaze else 0)
            assert abs(row[7] - (1 - 0.9 ** t_seg)) < 1e-12 and abs(row[8] - (1 - 0.999 ** t_seg) ** 0.5) < 1e-12
    assert opt.t == 4 and opt._lag.as_dict() == {gaze[0]: 2, gaze[1]: 2}   # the gaze slots are at their 2nd update, the rest at the 4th
    # overlapping skip sets ADD on the overlap: [640, 704) skipped twice above, now [600, 704) once more
    opt._note_skipped(((600, 704),))
    assert opt._lag.as_dict() == {(600, 640): 1, (640, 704): 3, (832, 896): 2}
    opt.t += 1
    assert [(a, b, t) for a, b, t in opt._segments(590, 720, ())] == [(590, 600, 5), (600, 640, 4), (640, 704, 2), (704, 720, 5)]
    with pytest.raises(ValueError):
        opt._lag.lag(620, 660)
    back = LagMap(list(opt._lag.cuts), list(opt._lag.vals))   # what state_dict() / load_state_dict() carry
    assert back.as_dict() == opt._lag.as_dict()
    # without gaze dropout (or without a backbone range) the plan degenerates
    eng.model.configs.gaze_dropout = 0.0
    assert eng._plan_segments() == [(0, 600, True), (600, 1000, False)]
    eng._gps_range = None
    assert eng._plan_segments() == [(0, 1000, False)]


def test_producer_name_looks_through_views_only():
    """kernels._producer_name (round 4, slab-carried tensors): the autograd node that produced a tensor, looking through
    reshape / view nodes ONLY -- a placeholder gradient may pass a view backward untouched, but not a copy, a transpose
    or an arithmetic node (then the slab path must stay off)."""
    import torch
    from routeformer_amd import kernels as K

    class _BnEluPool(torch.autograd.Function):  # (same class name as the product's node: the name is what is compared)
        @staticmethod
        def forward(ctx, x):
            return x * 2.0

        @staticmethod
        def backward(ctx, g):
            return g * 2.0

    x = torch.randn(2, 3, 4, requires_grad=True)
    y = _BnEluPool.apply(x)
    assert K._producer_name(y) == "_BnEluPoolBackward"
    assert K._producer_name(y.reshape(6, 4)) == "_BnEluPoolBackward"
    assert K._producer_name(y.view(2, 12).view(24)) == "_BnEluPoolBackward"
    assert K._producer_name(y.transpose(1, 2).contiguous()) != "_BnEluPoolBackward"
    assert K._producer_name(y + 1.0) != "_BnEluPoolBackward"
    assert K._producer_name(x) == "" and K._producer_name(torch.zeros(3)) == ""
