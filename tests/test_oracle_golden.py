"""CPU: the oracle (oracle/routeformer_oracle.py) against golden vectors produced by the reference.

This is what pins the oracle (the reference ships no tests of its own)."""
import math
import sys

import numpy as np
import pytest
import torch

from conftest import RSEED, build_product_model, case_item, draws, golden, rel_err, t

sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__file__), ".."))
from oracle import routeformer_oracle as O  # noqa: E402

TOL = 2e-5  # fp32 CPU restatement vs fp32 CPU reference (different op order only)


def _attn_inputs(meta_seed, B, LQ, LK, H, E):
    g = torch.Generator().manual_seed(meta_seed)
    q = torch.randn(B, LQ, H, E, generator=g).requires_grad_()
    k = torch.randn(B, LK, H, E, generator=g).requires_grad_()
    v = torch.randn(B, LK, H, E, generator=g).requires_grad_()
    return g, q, k, v


ATTN_TAGS = ["frame", "fusion", "decself", "gps_enc", "gps_enc5", "gps_decself", "gps_deccross", "gps_def_cross"]


@pytest.mark.parametrize("tag", ATTN_TAGS)
def test_prob_attention(tag):
    G = golden("attention")
    LQ, LK, H, E, masked, factor, gps = (int(x) for x in G[tag + ".meta"])
    g, q, k, v = _attn_inputs(100 + LQ * 7 + LK, 2, LQ, LK, H, E)
    idx = t(G[tag + ".idx"]).long()
    ctx = O.prob_attention(q, k, v, idx, factor, bool(masked), gps_variant=bool(gps))
    w = torch.randn(ctx.shape, generator=g)
    (ctx * w).sum().backward()
    assert rel_err(ctx, G[tag + ".ctx"]) < TOL
    for name, grad in (("dq", q.grad), ("dk", k.grad), ("dv", v.grad)):
        assert rel_err(grad, G[f"{tag}.{name}"]) < TOL, name


def test_full_attention():
    G = golden("attention")
    g, q, k, v = _attn_inputs(55, 2, 40, 40, 8, 8)
    ctx = O.full_attention(q, k, v)
    w = torch.randn(ctx.shape, generator=g)
    (ctx * w).sum().backward()
    assert rel_err(ctx, G["full.ctx"]) < TOL
    assert rel_err(q.grad, G["full.dq"]) < TOL and rel_err(k.grad, G["full.dk"]) < TOL
    assert rel_err(v.grad, G["full.dv"]) < TOL


def _block_sd(module_ctor):
    from routeformer_amd import synthetic
    m = module_ctor()
    return synthetic.synth_state_dict(m.state_dict(), 7)


def test_perceive_blocks():
    from routeformer_amd.models.blocks import PerceiveDecoder, PerceiveEncoder
    G = golden("blocks")
    sd = _block_sd(lambda: PerceiveEncoder(in_channels=240, out_channels=64, out_len=1, n_heads=8, layers=2,
                                           d_ff=64, dropout=0.0))
    sd = {"m." + k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    y = O.perceive_encoder(sd, "m", t(G["enc.x"]), 8, 1, O.IndexSource(draws(G, "enc.")))
    assert rel_err(y, G["enc.y"]) < TOL
    y.square().sum().backward()
    names = [str(n) for n in G["enc.grad_names"]]
    stats = G["enc.grad_stats"]
    for n, (nrm, sm) in zip(names, stats):
        g = sd["m." + n].grad.double()
        assert abs(float(g.norm()) - nrm) <= 1e-4 * max(1.0, nrm), n
    for key in G.files:
        if key.startswith("enc.grad::"):
            assert rel_err(sd["m." + key[len("enc.grad::"):]].grad, G[key]) < 1e-4, key

    sd2 = _block_sd(lambda: PerceiveEncoder(in_channels=2, out_channels=64, out_len=40, n_heads=8, layers=2,
                                            d_ff=256, dropout=0.0))
    torch.manual_seed(RSEED)  # same seed => same host draws as the reference
    y2 = O.perceive_encoder({"m." + k: v for k, v in sd2.items()}, "m", t(G["enc2.x"]), 8, 40, O.IndexSource())
    assert rel_err(y2, G["enc2.y"]) < TOL

    sd3 = _block_sd(lambda: PerceiveDecoder(query_channels=64, value_channels=64, out_channels=64, out_len=40,
                                            dropout=0.0, d_ff=256, n_heads=8, layers=2, mix=False))
    mem, qry = t(G["dec.mem"]).requires_grad_(), t(G["dec.qry"]).requires_grad_()
    yd = O.perceive_decoder({"m." + k: v for k, v in sd3.items()}, "m", mem, qry, 8, 40,
                            O.IndexSource(draws(G, "dec.")))
    assert rel_err(yd, G["dec.y"]) < TOL
    yd.square().sum().backward()
    assert rel_err(mem.grad, G["dec.dmem"]) < 1e-4 and rel_err(qry.grad, G["dec.dqry"]) < 1e-4


@pytest.mark.parametrize("tag,preset,B,T,P,cin", [("tiny", "GPS_TINY", 3, 20, 10, 69),
                                                   ("default", "GPS_DEFAULT", 4, 10, 15, 5),
                                                   ("paper", "GPS_PAPER", 2, 40, 30, 69)])
def test_informer(tag, preset, B, T, P, cin):
    from routeformer_amd import presets, synthetic
    from routeformer_amd.models.gps_backbone import GPSBackboneConfig, Informer
    G = golden("informer")
    kw = getattr(presets, preset)
    x = t(G[tag + ".x"])
    for smart in (False, True):
        gcfg = GPSBackboneConfig(seq_len=T, label_len=T, pred_len=P, **kw)
        gcfg.output_attention, gcfg.smart_decoder, gcfg._enc_in, gcfg._c_out = False, smart, cin, cin - 3
        sd0 = synthetic.synth_state_dict(Informer(gcfg).state_dict(), 7)
        for mode in ("eval", "train"):
            key = f"{tag}.{'smart' if smart else 'vanilla'}.{mode}"
            if key + ".y" not in G.files:
                continue
            sd = {"m." + k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k and k[-3:] != ".pe")
                  for k, v in sd0.items()}
            bn = {}
            y = O.informer(sd, "m", x, pred_len=P, n_heads=gcfg.n_heads, factor=gcfg.factor,
                           activation=gcfg.activation, smart_decoder=smart, training=(mode == "train"),
                           idx=O.IndexSource(draws(G, key + ".")), bn_state=bn)
            assert rel_err(y, G[key + ".y"]) < 5e-5, key
            if mode == "train":
                y.square().mean().backward()
                for n, (nrm, _) in zip((str(s) for s in G[key + ".grad_names"]), G[key + ".grad_stats"]):
                    g = sd["m." + n].grad
                    assert g is not None, n
                    assert abs(float(g.double().norm()) - nrm) <= 2e-4 * max(1e-3, nrm), (key, n)
                assert rel_err(bn["m.encoder.conv_layers.0.norm.running_mean"], G[key + ".bn0_running_mean"]) < 1e-5
                assert rel_err(bn["m.encoder.conv_layers.0.norm.running_var"], G[key + ".bn0_running_var"]) < 1e-5


def test_hrnet16():
    from routeformer_amd import synthetic
    from routeformer_amd.models.video_backbone import HRNet16Backbone
    G = golden("hrnet")
    sd = synthetic.synth_state_dict(HRNet16Backbone().state_dict(), 7)
    for tag, n, hw in (("s64", 2, 64), ("s96", 1, 96), ("s224", 2, 224)):
        x = synthetic.synth_video(1, n, hw, hw, 11, "hrnet." + tag)[0]
        y = O.hrnet16_features(sd, "_Backbone", x)
        assert rel_err(y, G[tag + ".y"]) < TOL, tag


def test_helpers():
    G = golden("helpers")
    assert torch.equal(O.median_downsampler(t(G["gaze"]), 40), t(G["gaze_ds40"]))
    assert torch.equal(O.median_downsampler(t(G["gaze"])[:, :100], 7), t(G["gaze_ds7"]))
    assert rel_err(O.rotate(t(G["v"]), t(G["ang"])), G["rot"]) < 1e-6
    a, n = O.angle_and_norm(t(G["v"]))
    assert rel_err(a, G["angle"]) < 1e-6 and rel_err(n, G["norm"]) < 1e-6
    pred, true = t(G["pred"]), t(G["true"])
    for kind in ("mse", "mae", "smooth_l1"):
        assert abs(float(O.future_discounted_loss(pred, true, 0.97, kind, 1.0)) - float(G["loss." + kind])) < 1e-6
    assert abs(float(O.future_discounted_loss(t(G["feat_p"]), t(G["feat_t"]), 0.9)) - float(G["loss.dense"])) < 1e-6
    assert abs(float(O.ade(pred, true)) - float(G["ade"])) < 1e-6
    assert abs(float(O.fde(pred, true)) - float(G["fde"])) < 1e-6
    with pytest.raises(ValueError):
        O.median_downsampler(t(G["gaze"])[:, :10], 10)


CASES = ["c1_default", "c1_paper", "c1_recursive", "c1_noise", "c2_small", "c4_small", "c5_small", "ar_small", "c2_paper"]


@pytest.mark.parametrize("name", CASES)
def test_model_eval_forward(name):
    model, cfg, sd, c = build_product_model(name)  # product module used ONLY as a state-dict template
    G = golden(name)
    item = case_item(c)
    with torch.no_grad():
        orc = O.OracleRouteformer(cfg, sd, training=False, idx=O.IndexSource(draws(G, "eval.")))
        out = orc.forward(item["train"])
    pos, vis = out if isinstance(out, tuple) else (out, None)
    assert rel_err(pos, G["eval.future_gps"]) < 1e-4, name
    if vis is not None:
        assert rel_err(vis, G["eval.future_vis"]) < 1e-4


@pytest.mark.parametrize("name", ["c1_default", "c1_recursive", "c1_noise", "c2_small", "c4_small", "c5_small", "c1_paper"])
def test_model_train_step(name):
    model, cfg, sd, c = build_product_model(name)
    G = golden(name)
    item = case_item(c)
    for epoch in (0, 10):
        key = f"train{epoch}"
        if key + ".loss" not in G.files:
            continue
        sdg = {k: v.clone().requires_grad_(v.is_floating_point() and "video_backbone" not in k
                                           and "running" not in k and not k.endswith(".pe"))
               for k, v in sd.items()}
        torch.manual_seed(RSEED)  # the oracle draws from the host RNG in the reference's order
        orc = O.OracleRouteformer(cfg, sdg, training=True)
        res = orc.train_step(item, epoch)
        assert len(orc.idx.log) == int(G[key + ".n_draws"])
        assert rel_err(res["future_gps"], G[key + ".future_gps"]) < 1e-4
        for k in ("loss", "traj_loss", "ade", "fde"):
            assert abs(float(res[k]) - float(G[f"{key}.{k}"])) < 1e-4 * max(1.0, abs(float(G[f"{key}.{k}"]))), k
        if "dense_loss" in res:
            assert abs(float(res["dense_loss"]) - float(G[key + ".dense_loss"])) < 1e-4
        res["loss"].backward()
        # the oracle's autograd gradients are the element-wise gradient reference of the GPU tests: pinned here by the
        # reference's recorded L2 norm AND sum of every parameter gradient
        floor = 1e-3 * float(G[key + ".grad_stats"][:, 0].max())
        # two cases added in round 4 need two documented allowances: (i) c5_small's first Informer layer has nearly uniform
        # softmax rows (L = 80 keys): dS = P (dP - sum P dP) cancels 3-4 digits, so dW_q / dW_k carry ~4e-3 relative
        # summation-order noise between the reference's kernels and the oracle's (same effect as c2_paper's);
        # (ii) c1_paper's 832 x 832 weights: |sum| <= sqrt(numel) * norm, so the sum is compared on that scale (as the GPU
        # tests do) instead of the norm's
        ill = (".attn_layers.0.attention.query_projection.weight", ".attn_layers.0.attention.key_projection.weight")
        wide = name in ("c5_small", "c1_paper")
        for n, (nrm, total) in zip((str(s) for s in G[key + ".grad_names"]), G[key + ".grad_stats"]):
            g = sdg[n].grad
            got = 0.0 if g is None else float(g.double().norm())
            tol_n = 5e-3 if (name == "c5_small" and n.startswith("gps_backbone.encoder") and n.endswith(ill)) else 5e-4
            assert abs(got - nrm) <= tol_n * max(1e-3, nrm), (name, key, n, got, nrm)
            got_sum = 0.0 if g is None else float(g.double().sum())
            scale = max(1.0, float(np.sqrt(sdg[n].numel())) / 8) if wide else 1.0
            assert abs(got_sum - total) <= 2 * tol_n * max(floor, nrm) * scale, (name, key, n, got_sum, total)


# ------------------------------------------------------------------------------------------------
# nn.Dropout on the trainable path: the oracle with the reference's RECORDED masks (tests/golden/dropout.npz)
# ------------------------------------------------------------------------------------------------
def _check_grad_summary(G, key, sd, prefix, tol=1e-4):
    names = [str(n) for n in G[key + "grad_names"]]
    for n, (nrm, _) in zip(names, G[key + "grad_stats"]):
        g = sd[prefix + n].grad
        got = 0.0 if g is None else float(g.double().norm())
        assert abs(got - nrm) <= tol * max(1.0, nrm), (n, got, nrm)
    for f in G.files:
        if f.startswith(key + "grad::"):
            assert rel_err(sd[prefix + f[len(key + "grad::"):]].grad, G[f]) < tol, f


def test_dropout_blocks_vs_reference_masks():
    """Every dropout site of the encoder / decoder / GPS blocks, in the reference's call order and layouts: with the
    recorded keep-masks and key samples the oracle reproduces the reference's train-mode outputs and gradients."""
    from conftest import masks
    from routeformer_amd import presets, synthetic
    from routeformer_amd.models.blocks import PerceiveDecoder, PerceiveEncoder
    from routeformer_amd.models.gps_backbone import GPSBackboneConfig, Informer, Transformer
    G = golden("dropout")
    P = 0.1
    sd = _block_sd(lambda: PerceiveEncoder(in_channels=240, out_channels=64, out_len=1, n_heads=8, layers=2, d_ff=256,
                                           dropout=P))
    sd = {"m." + k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    x = t(G["enc.x"]).requires_grad_()
    src = O.DropoutSource(masks(G, "enc."))
    assert len(src.replay) == 2 * 3  # per layer: attention output, FFN hidden (B, d_ff, L), FFN output
    y = O.perceive_encoder(sd, "m", x, 8, 1, O.IndexSource(draws(G, "enc.")), dropout=P, drop=src)
    assert not src.replay
    assert rel_err(y, G["enc.y"]) < TOL
    y.square().sum().backward()
    assert rel_err(x.grad, G["enc.dx"]) < 1e-4
    _check_grad_summary(G, "enc.", sd, "m.")

    sd = _block_sd(lambda: PerceiveDecoder(query_channels=64, value_channels=64, out_channels=64, out_len=40, dropout=P,
                                           d_ff=256, n_heads=8, layers=2, mix=False))
    sd = {"m." + k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    mem, qry = t(G["dec.mem"]).requires_grad_(), t(G["dec.qry"]).requires_grad_()
    src = O.DropoutSource(masks(G, "dec."))
    assert len(src.replay) == 2 * 5  # per layer: self out, cross probabilities (B,H,L,S), cross out, FFN hidden, FFN out
    yd = O.perceive_decoder(sd, "m", mem, qry, 8, 40, O.IndexSource(draws(G, "dec.")), dropout=P, drop=src)
    assert not src.replay
    assert rel_err(yd, G["dec.y"]) < TOL
    yd.square().sum().backward()
    assert rel_err(mem.grad, G["dec.dmem"]) < 1e-4 and rel_err(qry.grad, G["dec.dqry"]) < 1e-4
    _check_grad_summary(G, "dec.", sd, "m.")

    for tag, cls in (("inf", Informer), ("tf", Transformer)):
        gcfg = GPSBackboneConfig(seq_len=20, label_len=20, pred_len=10, **dict(presets.GPS_TINY, dropout=P))
        gcfg.output_attention, gcfg.smart_decoder, gcfg._enc_in, gcfg._c_out = False, True, 69, 66
        sd0 = synthetic.synth_state_dict(cls(gcfg).state_dict(), 7)
        sd = {"g." + k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k and not k.endswith(".pe"))
              for k, v in sd0.items()}
        xg = t(G[tag + ".x"]).requires_grad_()
        src = O.DropoutSource(masks(G, tag + "."))
        if tag == "inf":
            yg = O.informer(sd, "g", xg, pred_len=10, n_heads=gcfg.n_heads, factor=gcfg.factor, activation=gcfg.activation,
                            smart_decoder=True, training=True, idx=O.IndexSource(draws(G, "inf.")), dropout=P, drop=src)
        else:
            yg = O.transformer_gps(sd, "g", xg, pred_len=10, n_heads=gcfg.n_heads, activation=gcfg.activation, dropout=P,
                                   drop=src)
        assert not src.replay, tag
        assert rel_err(yg, G[tag + ".y"]) < TOL, tag
        yg.square().mean().backward()
        assert rel_err(xg.grad, G[tag + ".dx"]) < 1e-4, tag
        _check_grad_summary(G, tag + ".", sd, "g.")


@pytest.mark.parametrize("kind", ["none", "view", "gaze"])
def test_dropout_train_step_vs_reference(kind):
    """One whole train step with view / gaze / feature dropout (the paper run's kinds, full_comparison.py:272-275):
    same host seed -> same view / gaze decisions and key samples as the reference (dropout masks do not consume the
    host generator), recorded masks -> same losses, trajectories and gradients; and the host generator ends where
    the reference left it."""
    from conftest import masks
    from routeformer_amd import presets
    from routeformer_amd.models import RouteformerConfig
    from routeformer_amd.models.gps_backbone import GPSBackboneConfig
    from routeformer_amd.models.video_backbone import VideoBackboneConfig
    G = golden("dropout")
    c = presets.case("c2_small")
    c["rf"] = dict(c["rf"], feature_dropout=0.1, view_dropout=0.6, gaze_dropout=0.2)
    c["gps"] = dict(c["gps"], dropout=0.1)
    _, cfg = presets.build_configs(c, GPSBackboneConfig, RouteformerConfig, VideoBackboneConfig)
    model, _, sd0, _ = build_product_model("c2_small")
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and "video_backbone" not in k and "running" not in k
                                      and not k.endswith(".pe")) for k, v in sd0.items()}
    item = case_item(c)
    seed = int(G["model.seeds"][["none", "view", "gaze"].index(kind)])
    key = f"model.{kind}."
    src = O.DropoutSource(masks(G, key))
    torch.manual_seed(seed)
    orc = O.OracleRouteformer(cfg, sd, training=True, drop=src)
    res = orc.train_step(item, 10)
    assert not src.replay
    assert len(orc.idx.log) == int(G[key + "n_draws"])
    assert all(torch.equal(a, b) for a, b in zip(orc.idx.log, draws(G, key)))
    assert torch.equal(torch.rand(1), t(G[key + "rng_after"]))
    assert rel_err(res["future_gps"], G[key + "future_gps"]) < 1e-4
    assert rel_err(res["target_vis"], G[key + "target_vis"]) < 1e-4
    for k in ("loss", "traj_loss", "dense_loss", "ade", "fde"):
        assert abs(float(res[k]) - float(G[key + k])) < 1e-4 * max(1.0, abs(float(G[key + k]))), k
    res["loss"].backward()
    _check_grad_summary(G, key, sd, "", tol=2e-4)
