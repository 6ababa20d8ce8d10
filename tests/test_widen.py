"""SURVEY 8(f) rows built on top of the hot path: the vanilla Transformer GPS backbone (#4), the LR
schedule of the training recipe (#1) and the 5-pass evaluation protocol (#2).  CPU tests pin the oracle
restatement and the host logic to fixtures produced by the reference's own classes
(tests/golden/make_golden.py: gen_widen); GPU tests hold the product to the same fixtures."""
import math
import os
import sys

import numpy as np
import pytest
import torch

from conftest import DSEED, build_product_model, draws, golden, t

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import routeformer_oracle as O  # noqa: E402

DEV = "cuda:0"


def rel_err(a, b):
    a, b = torch.as_tensor(np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a), dtype=torch.float64), \
        torch.as_tensor(np.asarray(b), dtype=torch.float64)
    return float((a - b).abs().max() / max(1.0, float(b.abs().max())))


def _gps_cfg(tag):
    from routeformer_amd import presets
    from routeformer_amd.models.gps_backbone import GPSBackboneConfig
    kw, B, T, P, cin = {"tiny": (presets.GPS_TINY, 3, 20, 10, 69), "default": (presets.GPS_DEFAULT, 4, 10, 15, 5)}[tag]
    g = GPSBackboneConfig(seq_len=T, label_len=T, pred_len=P, **kw)
    g.output_attention, g._enc_in, g._c_out = False, cin, cin - 3
    return g, P


# ------------------------------------------------------------------------------------------------
# CPU: oracle + host logic
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["tiny", "default"])
def test_oracle_transformer_backbone(tag):
    from routeformer_amd import synthetic
    from routeformer_amd.models.gps_backbone import Transformer
    G = golden("widen")
    gcfg, P = _gps_cfg(tag)
    net = Transformer(gcfg)
    sd0 = synthetic.synth_state_dict(net.state_dict(), 7)
    # same parameter names / shapes / values as the reference module
    want = float(G[f"transformer.{tag}.digest"])
    assert abs(synthetic.state_dict_digest(sd0) - want) < 1e-6 * want
    x = t(G[f"transformer.{tag}.x"])
    sd = {"m." + k: v.clone().requires_grad_(v.is_floating_point() and k[-3:] != ".pe") for k, v in sd0.items()}
    y = O.transformer_gps(sd, "m", x, pred_len=P, n_heads=gcfg.n_heads, activation=gcfg.activation)
    assert rel_err(y, G[f"transformer.{tag}.eval.y"]) < 5e-5
    assert rel_err(y, G[f"transformer.{tag}.train.y"]) < 5e-5  # dropout 0: train == eval
    y.square().mean().backward()
    for n, (nrm, _) in zip((str(s) for s in G[f"transformer.{tag}.train.grad_names"]), G[f"transformer.{tag}.train.grad_stats"]):
        g = sd["m." + n].grad
        assert g is not None and abs(float(g.double().norm()) - nrm) <= 1e-3 * max(1e-3, nrm), n  # fp32 CPU BLAS noise


def test_lr_schedule_bit_exact():
    """Oracle restatement and the product scheduler vs the reference class stepped once per epoch."""
    from routeformer_amd.optimizers import LinearWarmupCosineAnnealingLR
    G = golden("widen")
    for tag in ("driver", "short"):
        base, warm, mx, n = (float(v) for v in G[f"lr.{tag}.args"])
        want = [float(v) for v in G[f"lr.{tag}"]]
        assert O.warmup_cosine_lr(base, int(n), int(warm), int(mx)) == want

        class Opt:
            param_groups = None
        opt = Opt()
        opt.param_groups = [{"lr": base}]
        sch = LinearWarmupCosineAnnealingLR(opt, warmup_epochs=int(warm), max_epochs=int(mx))
        got = []
        for _ in range(int(n)):
            got.append(opt.param_groups[0]["lr"])
            sch.step()
        assert got == want, tag
        # closed form agrees with the chainable form to rounding
        sch.step(epoch=17)
        assert abs(opt.param_groups[0]["lr"] - want[17]) <= 1e-12 * max(1.0, want[17]) + 1e-18


@pytest.mark.parametrize("name", ["c1_default", "c2_small"])
def test_oracle_eval_protocol(name):
    from test_oracle_golden import case_item
    model, cfg, sd, c = build_product_model(name)
    G = golden("widen")
    item = case_item(c)
    with torch.no_grad():
        orc = O.OracleRouteformer(cfg, sd, training=False, idx=O.IndexSource(draws(G, f"eval.{name}.")))
        losses, ades, fdes, mean = orc.eval_step(item, epoch=0, passes=5)
    assert len(orc.idx.log) == int(G[f"eval.{name}.n_draws"])
    assert rel_err(mean, G[f"eval.{name}.mean_gps"]) < 1e-4
    rows = torch.stack([losses, ades, fdes], dim=1)
    assert rel_err(rows, G[f"eval.{name}.rows"]) < 1e-4


def test_oracle_transformer_inside_routeformer():
    from routeformer_amd import presets, synthetic
    from routeformer_amd.models import Routeformer, RouteformerConfig
    from routeformer_amd.models.gps_backbone import GPSBackboneConfig, Transformer
    from routeformer_amd.models.video_backbone import VideoBackboneConfig
    from test_oracle_golden import case_item
    G = golden("widen")
    c = presets.case("c1_default")
    _, cfg = presets.build_configs(c, GPSBackboneConfig, RouteformerConfig, VideoBackboneConfig)
    model = Routeformer(cfg, gps_backbone=Transformer, video_backbone=None)
    sd = synthetic.synth_state_dict(model.state_dict(), 7)
    sdg = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(".pe")) for k, v in sd.items()}
    orc = O.OracleRouteformer(cfg, sdg, training=True)
    orc.gps_kind = "transformer"
    res = orc.train_step(case_item(c), 0)
    assert rel_err(res["future_gps"], G["transformer.c1.train.future_gps"]) < 1e-4
    for k, want in zip(("loss", "ade", "fde"), G["transformer.c1.train.scalars"]):
        assert abs(float(res[k]) - float(want)) < 1e-4 * max(1.0, abs(float(want))), k


def _mmt(device="cpu"):
    from routeformer_amd import presets, synthetic
    from routeformer_amd.experiments import MultiModalTransformer
    from routeformer_amd.models import RouteformerConfig
    from routeformer_amd.models.gps_backbone import GPSBackboneConfig
    from routeformer_amd.models.video_backbone import HRNet16Backbone, VideoBackboneConfig
    c = presets.case("mmt_small")
    _, cfg = presets.build_configs(c, GPSBackboneConfig, RouteformerConfig, VideoBackboneConfig)
    model = MultiModalTransformer(cfg, video_backbone=HRNet16Backbone)
    sd = synthetic.synth_state_dict(model.state_dict(), 7)
    model.load_state_dict(sd)
    item = synthetic.synth_item(c["B"], c["T"], c["P"], DSEED, c["H"], c["W"], streams=c["streams"], gaze=c["gaze"])
    return model.to(device), cfg, sd, item


def test_oracle_multimodal_transformer_baseline():
    """experiments/multimodal_transformer: state_dict layout, forward, loss and gradients vs the reference."""
    from routeformer_amd import synthetic
    model, cfg, sd, item = _mmt()
    G = golden("widen")
    want = float(G["mmt.digest"])
    assert abs(synthetic.state_dict_digest(sd) - want) < 1e-6 * want
    sdg = {k: v.clone().requires_grad_(v.is_floating_point() and "video_backbone" not in k and not k.endswith(".pe"))
           for k, v in sd.items()}
    src = O.IndexSource(draws(G, "mmt."))
    y = O.multimodal_transformer(sdg, cfg, item["train"], src)
    assert len(src.log) == len(draws(G, "mmt."))
    assert rel_err(y, G["mmt.future_gps"]) < 1e-4
    loss = O.future_discounted_loss(y, item["target"]["gps"].to(torch.float32), O.discount_for_epoch(cfg.discount_factor, 0))
    assert abs(float(loss) - float(G["mmt.loss"])) < 1e-4 * max(1.0, float(G["mmt.loss"]))
    loss.backward()
    for n, (nrm, _) in zip((str(s) for s in G["mmt.grad_names"]), G["mmt.grad_stats"]):
        g = sdg[n].grad
        got = 0.0 if g is None else float(g.double().norm())
        assert abs(got - nrm) <= 1e-3 * max(1e-3 * float(G["mmt.grad_stats"][:, 0].max()), nrm), (n, got, nrm)


# ------------------------------------------------------------------------------------------------
# GPU: the product
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_gpu_multimodal_transformer_baseline():
    """The baseline on the HIP kernels vs the reference: same host-RNG draws, trajectory within 1e-3 (fp32 mode,
    oracle selections imposed), loss and per-parameter gradient norms."""
    from routeformer_amd import kernels as K
    from routeformer_amd.losses import FutureDiscountedLoss
    from routeformer_amd.models import blocks
    model, cfg, sd, item = _mmt(DEV)
    G = golden("widen")
    with torch.no_grad():
        src = O.IndexSource(draws(G, "mmt."))
        O.multimodal_transformer(sd, cfg, item["train"], src)
    model.train()
    batch = {k: v.to(DEV) for k, v in item["train"].items()}
    log = []
    blocks.SAMPLER.log = log
    K.TOPS.forced = [tp.clone() for tp in src.tops]
    torch.manual_seed(1234)
    try:
        y = model(batch)
    finally:
        K.TOPS.forced = None
        blocks.SAMPLER.log = None
    want = draws(G, "mmt.")
    assert len(log) == len(want) and all(torch.equal(a.cpu().long(), b) for a, b in zip(log, want))
    assert rel_err(y, G["mmt.future_gps"]) < 1e-3
    tl = FutureDiscountedLoss(cfg.discount_factor, cfg.epsilon, loss_function="smooth_l1")
    loss = tl(y, item["target"]["gps"].to(DEV).float())
    assert abs(float(loss) - float(G["mmt.loss"])) < 1e-3 * max(1.0, float(G["mmt.loss"]))
    loss.backward()
    named = dict(model.named_parameters())
    floor = 1e-3 * float(G["mmt.grad_stats"][:, 0].max())
    for n, (nrm, _) in zip((str(s) for s in G["mmt.grad_names"]), G["mmt.grad_stats"]):
        g = named[n].grad
        got = 0.0 if g is None else float(g.double().norm())
        assert abs(got - nrm) <= 5e-3 * max(floor, nrm), (n, got, nrm)
    for f in G.files:
        if f.startswith("mmt.grad::"):
            assert rel_err(named[f[len("mmt.grad::"):]].grad, G[f]) < 5e-3, f
@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["tiny", "default"])
@pytest.mark.parametrize("prec,tol", [("f32", 1e-3), ("bf16", 3e-2)])
def test_gpu_transformer_backbone(tag, prec, tol):
    """Stand-alone backbone outputs (random weights, O(1) activations, not trajectories): 1e-3 in fp32 mode,
    3e-2 with bf16 matrix-core inputs (the block-level bf16 bound used throughout tests/test_gpu_*), relative
    to max(1, max|ref|); per-parameter gradient norms vs the reference."""
    from routeformer_amd import kernels as K, synthetic
    from routeformer_amd.models.gps_backbone import Transformer
    G = golden("widen")
    gcfg, P = _gps_cfg(tag)
    K.set_precision(prec)
    try:
        net = Transformer(gcfg)
        net.load_state_dict(synthetic.synth_state_dict(net.state_dict(), 7))
        net = net.to(DEV).train()
        x = t(G[f"transformer.{tag}.x"]).to(DEV)
        y = net(x)
        assert rel_err(y, G[f"transformer.{tag}.train.y"]) < tol
        y.square().mean().backward()
        named = dict(net.named_parameters())
        gtol = 2e-3 if prec == "f32" else 6e-2
        stats = G[f"transformer.{tag}.train.grad_stats"]
        # bf16: structurally tiny gradients (|g| ~ 1e-3 of the typical norm) are rounding noise; floor the scale
        floor = 1e-3 if prec == "f32" else 0.02 * float(stats[:, 0].max())
        for n, (nrm, _) in zip((str(s) for s in G[f"transformer.{tag}.train.grad_names"]), stats):
            got = float(named[n].grad.double().norm())
            assert abs(got - nrm) <= gtol * max(floor, nrm), (n, got, nrm)
        with torch.no_grad():
            assert rel_err(net.eval()(x), G[f"transformer.{tag}.eval.y"]) < tol
    finally:
        K.set_precision("f32")


@pytest.mark.gpu
def test_gpu_transformer_inside_routeformer_train_step():
    from routeformer_amd import presets, synthetic
    from routeformer_amd.engine import train_step_losses
    from routeformer_amd.models import Routeformer, RouteformerConfig
    from routeformer_amd.models.gps_backbone import GPSBackboneConfig, Transformer
    from routeformer_amd.models.video_backbone import VideoBackboneConfig
    from test_oracle_golden import case_item
    G = golden("widen")
    c = presets.case("c1_default")
    _, cfg = presets.build_configs(c, GPSBackboneConfig, RouteformerConfig, VideoBackboneConfig)
    model = Routeformer(cfg, gps_backbone=Transformer, video_backbone=None)
    model.load_state_dict(synthetic.synth_state_dict(model.state_dict(), 7))
    model = model.to(DEV).train()
    item = {k: {n: v.to(DEV) for n, v in d.items()} for k, d in case_item(c).items()}
    res = train_step_losses(model, item, 0)
    res["loss"].backward()
    assert rel_err(res["future_gps"], G["transformer.c1.train.future_gps"]) < 1e-3
    for k, want in zip(("loss", "ade", "fde"), G["transformer.c1.train.scalars"]):
        assert abs(float(res[k]) - float(want)) < 1e-3 * max(1.0, abs(float(want))), k
    named = dict(model.named_parameters())
    for n, (nrm, _) in zip((str(s) for s in G["transformer.c1.train.grad_names"]), G["transformer.c1.train.grad_stats"]):
        got = float(named[n].grad.double().norm())
        assert abs(got - nrm) <= 2e-3 * max(1e-3, nrm), (n, got, nrm)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c1_default", "c2_small"])
def test_gpu_eval_protocol(name):
    """engine.eval_step: same host-RNG draws as the reference (seed 12345, 5 passes), mean trajectory and
    per-sample loss / ADE / FDE within 1e-3; selections teacher-forced from the oracle (ProbSparse top-u is
    discontinuous)."""
    from routeformer_amd import kernels as K
    from routeformer_amd.engine import eval_step
    from routeformer_amd.models import blocks
    from test_oracle_golden import case_item
    model, cfg, sd, c = build_product_model(name, DEV)
    G = golden("widen")
    item = case_item(c)
    with torch.no_grad():
        orc = O.OracleRouteformer(cfg, sd, training=False, idx=O.IndexSource(draws(G, f"eval.{name}.")))
        orc.eval_step(item, epoch=0, passes=5)
    model.eval()
    dev_item = {k: {n: v.to(DEV) for n, v in d.items()} for k, d in item.items()}
    log = []
    blocks.SAMPLER.log = log
    K.TOPS.forced = [tp.clone() for tp in orc.idx.tops]
    try:
        losses, ades, fdes, mean = eval_step(model, dev_item, epoch=0)
    finally:
        K.TOPS.forced = None
        blocks.SAMPLER.log = None
    want = draws(G, f"eval.{name}.")
    assert len(log) == len(want) and all(torch.equal(a.cpu().long(), b) for a, b in zip(log, want))
    assert rel_err(mean, G[f"eval.{name}.mean_gps"]) < 1e-3
    assert rel_err(torch.stack([losses, ades, fdes], dim=1), G[f"eval.{name}.rows"]) < 1e-3
