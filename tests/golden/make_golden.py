"""Generate the golden fixtures under tests/golden/*.npz by running the REFERENCE itself.

Runs only in the build container (needs /root/reference, CPU).  It imports the reference's own
hot-path modules through ``ref_bootstrap`` (no reference source is copied), loads seeded synthetic
weights (``routeformer_amd.synthetic``) into the reference modules, feeds seeded synthetic inputs
and stores ONLY data: outputs, the recorded ``torch.randint`` index samples (host RNG, SURVEY
Appendix D), losses/metrics and gradient summaries.  Inputs and weights are re-derivable from the
seed, so fixtures stay small.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [case ...]
"""
import contextlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True

import torch  # noqa: E402

import ref_bootstrap  # noqa: E402
from routeformer_amd import presets, synthetic  # noqa: E402

REF = ref_bootstrap.bootstrap()
RefHRNet16 = ref_bootstrap.build_hrnet16(REF)
WSEED, DSEED, RSEED = 7, 11, 1234  # weights / data / host RNG seeds used by every fixture


@contextlib.contextmanager
def record_randint(log):
    real = torch.randint

    def wrapped(*a, **k):
        r = real(*a, **k)
        log.append(r.clone())
        return r

    torch.randint = wrapped
    try:
        yield
    finally:
        torch.randint = real


def load_synth(module, seed=WSEED):
    sd = synthetic.synth_state_dict(module.state_dict(), seed)
    module.load_state_dict(sd)
    return sd


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB, {len(out)} arrays)")


def pack_draws(log, prefix="draw"):
    return {f"{prefix}{i:03d}": t.to(torch.int16) for i, t in enumerate(log)}


def grad_summary(model, full=()):
    """Per-parameter (L2 norm, sum) of .grad + a few full small grads."""
    names, stats, out = [], [], {}
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        names.append(n)
        g = p.grad.double()
        stats.append([float(g.norm()), float(g.sum())])
    out["grad_names"] = np.array(names)
    out["grad_stats"] = np.array(stats, dtype=np.float64)
    named = dict(model.named_parameters())
    for n in full:
        if n in named and named[n].grad is not None:
            out["grad::" + n] = named[n].grad.float()
    return out


# ----------------------------------------------------------------------------------------------
# block-level fixtures
# ----------------------------------------------------------------------------------------------
def gen_attention():
    """ProbAttention (both variants), FullAttention: outputs + input grads.
    Shapes = SURVEY Appendix B rows (b,h reduced)."""
    out = {}
    cfgs = [  # tag, variant, L_Q, L_K, H, E, masked, factor
        ("frame", "cm", 65, 65, 8, 16, False, 5),
        ("fusion", "cm", 160, 160, 4, 16, False, 5),
        ("decself", "cm", 40, 40, 8, 8, True, 5),
        ("gps_enc", "gps", 40, 40, 2, 104, False, 4),
        ("gps_enc5", "gps", 5, 5, 2, 104, False, 4),
        ("gps_decself", "gps", 70, 70, 2, 104, True, 4),
        ("gps_deccross", "gps", 70, 4, 2, 104, False, 4),
        ("gps_def_cross", "gps", 25, 6, 8, 16, False, 1),
    ]
    B = 2  # q, k, v, w are NOT stored: tests regenerate them from the same generator sequence
    for tag, variant, LQ, LK, H, E, masked, factor in cfgs:
        g = torch.Generator().manual_seed(100 + LQ * 7 + LK)
        q = torch.randn(B, LQ, H, E, generator=g).requires_grad_()
        k = torch.randn(B, LK, H, E, generator=g).requires_grad_()
        v = torch.randn(B, LK, H, E, generator=g).requires_grad_()
        cls = REF.cmt.ProbAttention if variant == "cm" else \
            sys.modules["routeformer.models.gps_backbone.layers.SelfAttentionFamily"].ProbAttention
        attn = cls(masked, factor, attention_dropout=0.0)
        log = []
        torch.manual_seed(RSEED)
        with record_randint(log):
            ctx, _ = attn(q, k, v, None)
        w = torch.randn(ctx.shape, generator=g)
        (ctx * w).sum().backward()
        out.update({f"{tag}.ctx": ctx, f"{tag}.dq": q.grad, f"{tag}.dk": k.grad,
                    f"{tag}.dv": v.grad, f"{tag}.idx": log[0].to(torch.int16),
                    f"{tag}.meta": np.array([LQ, LK, H, E, int(masked), factor,
                                             int(variant == "gps")])})
    # full attention (cross-modal decoder cross attention), L=S=40, H=8, E=8
    g = torch.Generator().manual_seed(55)
    q = torch.randn(B, 40, 8, 8, generator=g).requires_grad_()
    k = torch.randn(B, 40, 8, 8, generator=g).requires_grad_()
    v = torch.randn(B, 40, 8, 8, generator=g).requires_grad_()
    fa = REF.cmt.FullAttention(False, 5, attention_dropout=0.0)
    ctx, _ = fa(q, k, v, None)
    w = torch.randn(ctx.shape, generator=g)
    (ctx * w).sum().backward()
    out.update({"full.ctx": ctx,
                "full.dq": q.grad, "full.dk": k.grad, "full.dv": v.grad})
    save("attention", **out)


def gen_blocks():
    """PerceiveEncoder / PerceiveDecoder / Informer at small sizes: outputs + grad summaries."""
    out = {}
    g = torch.Generator().manual_seed(5)
    # frame-encoder-like: in 240 -> out_len 1, 2 layers
    enc = REF.cmt.PerceiveEncoder(in_channels=240, out_channels=64, out_len=1, n_heads=8, layers=2,
                                  d_ff=64, dropout=0.0)
    load_synth(enc)
    x = torch.randn(3, 65, 240, generator=g)
    log = []
    torch.manual_seed(RSEED)
    with record_randint(log):
        y = enc(x)
    y.square().sum().backward()
    out.update({"enc.x": x, "enc.y": y, **{"enc." + k: v for k, v in pack_draws(log).items()},
                **{"enc." + k: v for k, v in grad_summary(enc, full=(
                    "projection.bias", "encoder.norm.weight", "value_embedding.tokenConv.bias",
                    "encoder.attn_layers.0.attention.query_projection.bias")).items()}})
    # gaze encoder-like: in 2 -> out_len 40
    enc2 = REF.cmt.PerceiveEncoder(in_channels=2, out_channels=64, out_len=40, n_heads=8, layers=2,
                                   d_ff=256, dropout=0.0)
    load_synth(enc2)
    x2 = torch.rand(2, 40, 2, generator=g)
    log = []
    torch.manual_seed(RSEED)
    with record_randint(log):
        y2 = enc2(x2)
    out.update({"enc2.x": x2, "enc2.y": y2})
    # decoder: gaze tokens query FoV-video features
    dec = REF.cmt.PerceiveDecoder(query_channels=64, value_channels=64, out_channels=64, out_len=40,
                                  dropout=0.0, d_ff=256, n_heads=8, layers=2, mix=False)
    load_synth(dec)
    mem = torch.randn(2, 40, 64, generator=g, requires_grad=True)
    qry = torch.randn(2, 40, 64, generator=g, requires_grad=True)
    log = []
    torch.manual_seed(RSEED)
    with record_randint(log):
        yd = dec(mem, qry)
    yd.square().sum().backward()
    out.update({"dec.mem": mem, "dec.qry": qry, "dec.y": yd, "dec.dmem": mem.grad,
                "dec.dqry": qry.grad, **{"dec." + k: v for k, v in pack_draws(log).items()},
                **{"dec." + k: v for k, v in grad_summary(dec, full=(
                    "projection.bias", "decoder.layers.0.norm2.weight")).items()}})
    save("blocks", **out)


def gen_informer():
    out = {}
    for tag, kw, B, T, P, cin in (("tiny", presets.GPS_TINY, 3, 20, 10, 69),
                                  ("default", presets.GPS_DEFAULT, 4, 10, 15, 5),
                                  ("paper", presets.GPS_PAPER, 2, 40, 30, 69)):
        for smart in (False, True):
            gcfg = REF.gps.GPSBackboneConfig(seq_len=T, label_len=T, pred_len=P, **kw)
            gcfg.output_attention = False
            gcfg.smart_decoder = smart
            gcfg._enc_in = cin
            gcfg._c_out = cin - 3
            torch.manual_seed(0)
            net = REF.gps.Informer(gcfg)
            load_synth(net)
            g = torch.Generator().manual_seed(17)
            x = torch.randn(B, T, cin, generator=g)
            for mode in ("eval", "train"):
                if tag == "paper" and (mode == "train") != smart:
                    continue  # keep the fixture small: paper = (vanilla, eval) and (smart, train)
                net.train(mode == "train")
                load_synth(net)  # reset BN running stats
                net.zero_grad()
                log = []
                torch.manual_seed(RSEED)
                with record_randint(log):
                    y = net(x)
                key = f"{tag}.{'smart' if smart else 'vanilla'}.{mode}"
                out[key + ".y"] = y
                out.update({f"{key}.{k}": v for k, v in pack_draws(log).items()})
                if mode == "train":
                    y.square().mean().backward()
                    out.update({f"{key}.{k}": v for k, v in grad_summary(net, full=(
                        "decoder.projection.bias", "encoder.norm.weight",
                        "encoder.conv_layers.0.norm.weight", "encoder.conv_layers.0.downConv.bias",
                        "enc_embedding.temporal_embedding.embed.weight")).items()})
                    sd = net.state_dict()
                    out[key + ".bn0_running_mean"] = sd["encoder.conv_layers.0.norm.running_mean"]
                    out[key + ".bn0_running_var"] = sd["encoder.conv_layers.0.norm.running_var"]
            out[tag + ".x"] = x
    save("informer", **out)


def gen_hrnet():
    net = RefHRNet16()
    load_synth(net)
    out = {}
    for tag, n, hw in (("s64", 2, 64), ("s96", 1, 96), ("s224", 2, 224)):
        x = synthetic.synth_video(1, n, hw, hw, DSEED, "hrnet." + tag)[0]
        y = net(x)
        out[tag + ".y"] = y
        if tag == "s64":
            f2, f3, f4 = net._Backbone(x.float())
            out["s64.feats_abs_mean"] = np.array([float(f.abs().mean()) for f in (f2, f3, f4)])
            out["s64.trunk"] = f4
    save("hrnet", **out)


def gen_helpers():
    g = torch.Generator().manual_seed(3)
    gaze = torch.rand(2, 1600, 2, generator=g)
    v = torch.randn(3, 40, 2, generator=g)
    ang = torch.randn(3, 1, 1, generator=g)
    a, n = REF.vec.estimate_angle_and_norm(v)
    out = {"gaze": gaze, "gaze_ds40": REF.flt.median_downsampler(gaze, 40),
           "gaze_ds7": REF.flt.median_downsampler(gaze[:, :100], 7),
           "v": v, "ang": ang, "rot": REF.vec.rotate(v, ang), "angle": a, "norm": n}
    pred = torch.randn(4, 30, 2, generator=g)
    true = torch.randn(4, 30, 2, generator=g) * 2
    for lf in ("mse", "mae", "smooth_l1"):
        loss = REF.loss.FutureDiscountedLoss({0: 0.97, 100: 0.98}, 1.0, loss_function=lf)
        out["loss." + lf] = loss(pred, true)
    loss = REF.loss.FutureDiscountedLoss(0.9, 0.3, loss_function="smooth_l1")
    feat_p, feat_t = torch.randn(4, 30, 64, generator=g), torch.randn(4, 30, 64, generator=g)
    out.update({"pred": pred, "true": true, "feat_p": feat_p, "feat_t": feat_t,
                "loss.dense": loss(feat_p, feat_t), "ade": REF.score.ade(pred, true),
                "fde": REF.score.fde(pred, true)})
    save("helpers", **out)


# ----------------------------------------------------------------------------------------------
# whole-model fixtures
# ----------------------------------------------------------------------------------------------
def build_ref_model(c):
    gps_cfg, rf_cfg = presets.build_configs(c, REF.gps.GPSBackboneConfig, REF.cfg.RouteformerConfig,
                                            REF.vbc.VideoBackboneConfig)
    torch.manual_seed(0)
    model = REF.rf.Routeformer(rf_cfg, gps_backbone=REF.gps.Informer,
                               video_backbone=RefHRNet16 if rf_cfg.with_video else None)
    return model, rf_cfg


def ref_train_step(model, item, epoch):
    """The Routeformer branch of ParallelTrainer.training_step (full_comparison.py:476-521)."""
    cfg = model.configs
    tl = REF.loss.FutureDiscountedLoss(cfg.discount_factor, cfg.epsilon, loss_function="smooth_l1")
    dl = REF.loss.FutureDiscountedLoss(cfg.discount_factor, cfg.visual_epsilon,
                                       loss_function="smooth_l1")
    tl.current_epoch = dl.current_epoch = epoch
    inp, target = item["train"], item["target"]
    target_gps = target["gps"].to(torch.float32)
    res = {}
    if cfg.dense_prediction:
        future_gps, future_vis = model(inp)
        _, target_vis = model.preprocess_batch(target, training=False)
        target_vis = target_vis[:, : future_vis.shape[1]]
        step = cfg.autoregressive_step_size
        if cfg.autoregressive:
            future_gps, target_gps = future_gps[:, :step], target_gps[:, :step]
        traj = tl(future_gps, target_gps)
        if cfg.autoregressive:
            traj = traj * (cfg.gps_backbone_config.pred_len / step)
        target_vis = target_vis.detach()
        if cfg.autoregressive:
            future_vis, target_vis = future_vis[:, :step], target_vis[:, :step]
        dense = dl(future_vis, target_vis)
        w = (cfg.dense_loss_ratio * traj / max(dense, 1e-6)).detach()
        if epoch < 10:
            w = 0
        loss = traj + w * dense
        res.update(dense_loss=dense, future_vis=future_vis, target_vis=target_vis)
    else:
        future_gps = model(inp)
        traj = tl(future_gps, target_gps)
        loss = traj
    res.update(loss=loss, traj_loss=traj, future_gps=future_gps,
               ade=REF.score.ade(future_gps, target_gps), fde=REF.score.fde(future_gps, target_gps))
    return res


FULL_GRADS = ("left_video_embedding", "video_output_embedding", "gaze_video_embedding",
              "gps_backbone.decoder.projection.bias", "gps_backbone.encoder.norm.weight",
              "frame_encoder.projection.bias", "video_encoder.projection.bias",
              "gaze_encoder.projection.bias", "gaze_video_decoder.projection.bias",
              "frame_encoder.value_embedding.tokenConv.bias")


def gen_case(name):
    c = presets.case(name)
    model, cfg = build_ref_model(c)
    sd = load_synth(model)
    item = synthetic.synth_item(c["B"], c["T"], c["P"], DSEED, c["H"], c["W"],
                                streams=c["streams"], gaze=c["gaze"])
    out = {"digest": np.array(synthetic.state_dict_digest(sd)),
           "n_params": np.array(sum(p.numel() for p in model.parameters()))}
    # (1) eval forward
    model.eval()
    log = []
    torch.manual_seed(RSEED)
    with torch.no_grad(), record_randint(log):
        y = model(item["train"])
    if isinstance(y, tuple):
        out["eval.future_gps"], out["eval.future_vis"] = y
    else:
        out["eval.future_gps"] = y
    out.update({"eval." + k: v for k, v in pack_draws(log).items()})
    if cfg.with_video:
        log = []
        torch.manual_seed(RSEED)
        with torch.no_grad(), record_randint(log):
            md, vf = model.preprocess_batch(item["train"])
        out["eval.motion_dynamics"], out["eval.visual_features"] = md, vf
    # (2) train steps at epoch 0 (dense weight 0) and epoch 10 (dense loss active)
    if not cfg.autoregressive:
        for epoch in (0, 10):
            if epoch == 10 and not cfg.dense_prediction:
                continue
            model.train()
            model.load_state_dict(sd)
            model.zero_grad()
            log = []
            torch.manual_seed(RSEED)
            with record_randint(log):
                res = ref_train_step(model, item, epoch)
            res["loss"].backward()
            key = f"train{epoch}"
            for k, v in res.items():
                out[f"{key}.{k}"] = v
            out[f"{key}.n_draws"] = np.array(len(log))
            out.update({f"{key}.{k}": v for k, v in grad_summary(model, full=FULL_GRADS).items()})
            if epoch == 0:
                out.update({f"{key}.{k}": v for k, v in pack_draws(log).items()})
    save(name, **out)


def gen_widen():
    """Fixtures for the SURVEY 8(f) rows: the Transformer GPS backbone, the LR schedule and the 5-pass
    evaluation protocol -- each produced by the reference's own classes / call sequence."""
    out = {}
    # (1) vanilla Transformer GPS backbone (gps_backbone/Transformer.py), stand-alone and inside Routeformer
    for tag, kw, B, T, P, cin in (("tiny", presets.GPS_TINY, 3, 20, 10, 69), ("default", presets.GPS_DEFAULT, 4, 10, 15, 5)):
        gcfg = REF.gps.GPSBackboneConfig(seq_len=T, label_len=T, pred_len=P, **kw)
        gcfg.output_attention = False
        gcfg._enc_in = cin
        gcfg._c_out = cin - 3
        torch.manual_seed(0)
        net = REF.gps.Transformer(gcfg)
        load_synth(net)
        g = torch.Generator().manual_seed(17)
        x = torch.randn(B, T, cin, generator=g)
        net.eval()
        out[f"transformer.{tag}.eval.y"] = net(x)
        net.train()
        net.zero_grad()
        y = net(x)
        y.square().mean().backward()
        out[f"transformer.{tag}.train.y"] = y
        out.update({f"transformer.{tag}.train.{k}": v for k, v in grad_summary(net, full=(
            "decoder.projection.bias", "encoder.norm.weight", "decoder.layers.0.self_attention.query_projection.bias",
            "enc_embedding.temporal_embedding.embed.weight")).items()})
        out[f"transformer.{tag}.x"] = x
        out[f"transformer.{tag}.digest"] = np.array(synthetic.state_dict_digest(net.state_dict()))
    c = presets.case("c1_default")
    gps_cfg, rf_cfg = presets.build_configs(c, REF.gps.GPSBackboneConfig, REF.cfg.RouteformerConfig, REF.vbc.VideoBackboneConfig)
    torch.manual_seed(0)
    model = REF.rf.Routeformer(rf_cfg, gps_backbone=REF.gps.Transformer, video_backbone=None)
    load_synth(model)
    item = synthetic.synth_item(c["B"], c["T"], c["P"], DSEED, c["H"], c["W"], streams=c["streams"], gaze=c["gaze"])
    model.train()
    model.zero_grad()
    res = ref_train_step(model, item, 0)
    res["loss"].backward()
    out["transformer.c1.train.future_gps"] = res["future_gps"]
    out["transformer.c1.train.scalars"] = np.array([float(res[k]) for k in ("loss", "ade", "fde")])
    out.update({f"transformer.c1.train.{k}": v for k, v in grad_summary(model, full=("gps_backbone.decoder.projection.bias",)).items()})

    # (2) LinearWarmupCosineAnnealingLR as the driver steps it (full_comparison.py:702-709): one step per epoch
    import importlib.util
    spec = importlib.util.spec_from_file_location("_ref_lr", f"{ref_bootstrap.REF}/routeformer/optimizers/lr_scheduler.py")
    lrmod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lrmod)
    for tag, (base, warm, mx, n) in {"driver": (1e-5, 2, 200, 205), "short": (3e-4, 5, 30, 70)}.items():
        opt = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=base)
        sch = lrmod.LinearWarmupCosineAnnealingLR(opt, warmup_epochs=warm, max_epochs=mx)
        lrs = []
        for _ in range(n):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        out[f"lr.{tag}"] = np.array(lrs, dtype=np.float64)
        out[f"lr.{tag}.args"] = np.array([base, warm, mx, n], dtype=np.float64)

    # (3) the evaluation protocol, ParallelTrainer._eval_step (full_comparison.py:654-679), call for call
    for name in ("c1_default", "c2_small"):
        c = presets.case(name)
        model, cfg = build_ref_model(c)
        load_synth(model)
        model.eval()
        item = synthetic.synth_item(c["B"], c["T"], c["P"], DSEED, c["H"], c["W"], streams=c["streams"], gaze=c["gaze"])
        tl = REF.loss.FutureDiscountedLoss(cfg.discount_factor, cfg.epsilon, loss_function="smooth_l1")
        tl.current_epoch = 0
        log = []
        with torch.no_grad(), record_randint(log):
            torch.manual_seed(12345)
            runs = []
            for _ in range(5):
                o = model(item["train"])
                runs.append(o[0] if cfg.dense_prediction else o)
            mean = torch.stack(runs).mean(dim=0)
            tgt = item["target"]["gps"]
            rows = [[float(tl(mean[i:i + 1], tgt[i:i + 1])), float(REF.score.ade(mean[i:i + 1], tgt[i:i + 1])),
                     float(REF.score.fde(mean[i:i + 1], tgt[i:i + 1]))] for i in range(mean.shape[0])]
        out[f"eval.{name}.mean_gps"] = mean
        out[f"eval.{name}.rows"] = np.array(rows, dtype=np.float64)
        out[f"eval.{name}.n_draws"] = np.array(len(log))
        out.update({f"eval.{name}.{k}": v for k, v in pack_draws(log).items()})
    # (4) the PerceiveEncoder-based baseline, experiments/multimodal_transformer/multimodal_transformer.py
    import types
    vb = sys.modules["routeformer.models.video_backbone"]
    if not hasattr(vb, "SwinV2"):
        vb.SwinV2 = None  # the experiment file imports the (timm) class by name only; it is not used here
    spec = importlib.util.spec_from_file_location(
        "_ref_mmt", f"{ref_bootstrap.REF}/experiments/multimodal_transformer/multimodal_transformer.py")
    mmt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mmt)
    c = presets.case("mmt_small")
    gps_cfg, rf_cfg = presets.build_configs(c, REF.gps.GPSBackboneConfig, REF.cfg.RouteformerConfig, REF.vbc.VideoBackboneConfig)
    torch.manual_seed(0)
    model = mmt.MultiModalTransformer(rf_cfg, video_backbone=RefHRNet16)
    load_synth(model)
    item = synthetic.synth_item(c["B"], c["T"], c["P"], DSEED, c["H"], c["W"], streams=c["streams"], gaze=c["gaze"])
    model.train()
    model.zero_grad()
    log = []
    torch.manual_seed(RSEED)
    with record_randint(log):
        y = model(item["train"])
    loss = REF.loss.FutureDiscountedLoss(rf_cfg.discount_factor, rf_cfg.epsilon, loss_function="smooth_l1")(
        y, item["target"]["gps"].to(torch.float32))
    loss.backward()
    out["mmt.future_gps"] = y
    out["mmt.loss"] = np.array(float(loss))
    out["mmt.digest"] = np.array(synthetic.state_dict_digest(model.state_dict()))
    out.update({f"mmt.{k}": v for k, v in pack_draws(log).items()})
    out.update({f"mmt.{k}": v for k, v in grad_summary(model, full=("motion_linear.bias", "gaze_linear.weight",
                                                                      "frame_encoder.projection.bias")).items()})
    save("widen", **out)


@contextlib.contextmanager
def record_dropout(log, seed=99):
    """Record the keep-mask of every active ``F.dropout`` call (nn.Dropout.forward) of the reference, in call order.

    The reference's own ``F.dropout`` still does the arithmetic.  On the CPU it would draw its mask from the GLOBAL
    generator -- the one the ProbSparse key samples and the view / gaze dropout decisions come from -- whereas on a GPU
    (where the reference trains) it draws from the device generator and leaves the host stream alone.  So each call
    runs under a seed taken from a private generator, the mask is recovered by re-drawing under the same seed, and
    the global CPU generator is restored afterwards: host draws are exactly those of a GPU run."""
    import torch.nn.functional as F_
    real = F_.dropout
    gen = torch.Generator().manual_seed(seed)

    def wrapped(input, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return real(input, p, training, inplace)
        state = torch.get_rng_state()
        s = int(torch.empty((), dtype=torch.int64).random_(0, 2 ** 31 - 1, generator=gen))  # (not torch.randint: that is recorded)
        torch.manual_seed(s)
        y = real(input, p, training, False)
        torch.manual_seed(s)
        keep = torch.empty_like(input).bernoulli_(1 - p).bool()
        torch.set_rng_state(state)
        if not torch.allclose(y, input * keep / (1 - p), rtol=1e-6, atol=0):  # other draw order inside F.dropout
            keep = torch.where(input != 0, y != 0, torch.ones_like(keep))
            assert torch.allclose(y, input * keep / (1 - p), rtol=1e-6, atol=0)
        log.append(keep.detach().clone())
        return y

    F_.dropout = wrapped
    try:
        yield
    finally:
        F_.dropout = real


def pack_masks(log, prefix):
    """Keep-masks as one bit string + their shapes (padded to 4 dims with 0)."""
    bits = np.concatenate([m.numpy().reshape(-1) for m in log]) if log else np.zeros(0, dtype=bool)
    shapes = np.array([list(m.shape) + [0] * (4 - m.dim()) for m in log], dtype=np.int32).reshape(-1, 4)
    return {prefix + "maskbits": np.packbits(bits), prefix + "maskshapes": shapes}


def gen_dropout():
    """nn.Dropout on the trainable path (cross_modal_transformer.py:49,63,220-231,285-299; layers/Embedding.py:122-126):
    reference blocks and one whole train step per host-draw variant in TRAIN mode with dropout > 0 -- outputs, input
    and parameter gradients, the recorded key samples AND the recorded keep-masks of every dropout call."""
    out = {}
    g = torch.Generator().manual_seed(8)
    P = 0.1
    # (a) PerceiveEncoder (frame-encoder-like) -- sites per layer: attention output, FFN hidden, FFN output
    enc = REF.cmt.PerceiveEncoder(in_channels=240, out_channels=64, out_len=1, n_heads=8, layers=2, d_ff=256, dropout=P)
    load_synth(enc)
    enc.train()
    x = torch.randn(3, 65, 240, generator=g, requires_grad=True)
    log, masks = [], []
    torch.manual_seed(RSEED)
    with record_randint(log), record_dropout(masks):
        y = enc(x)
    y.square().sum().backward()
    out.update({"enc.x": x, "enc.y": y, "enc.dx": x.grad, **{"enc." + k: v for k, v in pack_draws(log).items()},
                **pack_masks(masks, "enc."),
                **{"enc." + k: v for k, v in grad_summary(enc, full=(
                    "projection.bias", "encoder.norm.weight", "encoder.attn_layers.0.conv1.bias",
                    "encoder.attn_layers.1.attention.out_projection.bias")).items()}})
    # (b) PerceiveDecoder: + dropout on the FullAttention probabilities of the cross attention
    dec = REF.cmt.PerceiveDecoder(query_channels=64, value_channels=64, out_channels=64, out_len=40, dropout=P, d_ff=256,
                                  n_heads=8, layers=2, mix=False)
    load_synth(dec)
    dec.train()
    mem = torch.randn(2, 40, 64, generator=g, requires_grad=True)
    qry = torch.randn(2, 40, 64, generator=g, requires_grad=True)
    log, masks = [], []
    torch.manual_seed(RSEED)
    with record_randint(log), record_dropout(masks):
        yd = dec(mem, qry)
    yd.square().sum().backward()
    out.update({"dec.mem": mem, "dec.qry": qry, "dec.y": yd, "dec.dmem": mem.grad, "dec.dqry": qry.grad,
                **{"dec." + k: v for k, v in pack_draws(log).items()}, **pack_masks(masks, "dec."),
                **{"dec." + k: v for k, v in grad_summary(dec, full=(
                    "projection.bias", "decoder.layers.0.norm2.weight", "decoder.layers.1.conv2.bias")).items()}})
    # (c) Informer / Transformer GPS backbones (DataEmbedding dropout, GPS-variant layers, causal FullAttention)
    for tag, cls in (("inf", REF.gps.Informer), ("tf", REF.gps.Transformer)):
        gcfg = REF.gps.GPSBackboneConfig(seq_len=20, label_len=20, pred_len=10, **dict(presets.GPS_TINY, dropout=P))
        gcfg.output_attention = False
        gcfg.smart_decoder = True
        gcfg._enc_in, gcfg._c_out = 69, 66
        torch.manual_seed(0)
        net = cls(gcfg)
        load_synth(net)
        net.train()
        xg = torch.randn(3, 20, 69, generator=torch.Generator().manual_seed(17), requires_grad=True)
        log, masks = [], []
        torch.manual_seed(RSEED)
        with record_randint(log), record_dropout(masks):
            yg = net(xg)
        yg.square().mean().backward()
        out.update({f"{tag}.x": xg, f"{tag}.y": yg, f"{tag}.dx": xg.grad,
                    **{f"{tag}." + k: v for k, v in pack_draws(log).items()}, **pack_masks(masks, f"{tag}."),
                    **{f"{tag}." + k: v for k, v in grad_summary(net, full=(
                        "decoder.projection.bias", "encoder.norm.weight",
                        "enc_embedding.temporal_embedding.embed.weight")).items()}})
    # (d) whole model, the paper run's dropouts in kind (full_comparison.py:272-275: view 0.6, gaze 0.2, feature 0.05;
    # feature 0.1 here so a small model sees enough dropped elements): one train step per host seed, seeds chosen so
    # that the host draws (torch.rand(1), routeformer.py:301,406-407) cover: nothing dropped / a view dropped / gaze dropped
    c = presets.case("c2_small")
    c["rf"] = dict(c["rf"], feature_dropout=P, view_dropout=0.6, gaze_dropout=0.2)
    c["gps"] = dict(c["gps"], dropout=P)
    model, cfg = build_ref_model(c)
    sd = load_synth(model)
    item = synthetic.synth_item(c["B"], c["T"], c["P"], DSEED, c["H"], c["W"], streams=c["streams"], gaze=c["gaze"])
    want = {"none": None, "view": None, "gaze": None}
    for seed in range(200):
        torch.manual_seed(seed)
        drop_one = bool(torch.rand(1) < 0.6)
        if drop_one:
            torch.rand(1)
        # the right and left stream draws sit between the view and the gaze decision: replay their count
        n_frame = (1 if drop_one else 2) * cfg.encoder_layers
        for _ in range(n_frame):
            torch.randint(65, (65, 25))
        drop_gaze = bool(torch.rand(1) < 0.2)
        kind = "gaze" if (drop_gaze and not drop_one) else ("view" if (drop_one and not drop_gaze) else
                                                           ("none" if not (drop_one or drop_gaze) else None))
        if kind and want[kind] is None:
            want[kind] = seed
        if all(v is not None for v in want.values()):
            break
    out["model.seeds"] = np.array([want["none"], want["view"], want["gaze"]])
    for kind, seed in want.items():
        model.train()
        model.load_state_dict(sd)
        model.zero_grad()
        log, masks = [], []
        torch.manual_seed(seed)
        with record_randint(log), record_dropout(masks):
            res = ref_train_step(model, item, 10)
        res["loss"].backward()
        key = f"model.{kind}."
        for k in ("loss", "traj_loss", "dense_loss", "ade", "fde", "future_gps", "target_vis"):
            out[key + k] = res[k]
        out[key + "n_draws"] = np.array(len(log))
        out[key + "rng_after"] = torch.rand(1)  # the host generator must end where the reference left it
        out.update({key + k: v for k, v in pack_draws(log).items()})
        out.update(pack_masks(masks, key))
        out.update({key + k: v for k, v in grad_summary(model, full=FULL_GRADS).items()})
    save("dropout", **out)


def gen_attn_out():
    """``output_attention=True``: the dense attention maps the reference hands back next to its outputs -- the Informer's
    and the Transformer's encoder maps (ProbAttention: uniform rows + the selected queries' softmax rows; FullAttention:
    the softmax), a PerceiveEncoder's, and Routeformer._forward's second return value."""
    out = {}
    g = torch.Generator().manual_seed(17)
    B, T, P, cin = 3, 20, 10, 69
    x = torch.randn(B, T, cin, generator=g)
    out["x"] = x
    for tag, cls in (("informer", REF.gps.Informer), ("transformer", REF.gps.Transformer)):
        gcfg = REF.gps.GPSBackboneConfig(seq_len=T, label_len=T, pred_len=P, **presets.GPS_TINY)
        gcfg.output_attention, gcfg.smart_decoder, gcfg._enc_in, gcfg._c_out = True, True, cin, cin - 3
        torch.manual_seed(0)
        net = cls(gcfg)
        load_synth(net)
        net.eval()
        log = []
        torch.manual_seed(RSEED)
        with record_randint(log), torch.no_grad():
            y, attns = net(x)
        out[f"{tag}.y"] = y
        for i, a in enumerate(attns):
            out[f"{tag}.attn{i}"] = a
        out[f"{tag}.n"] = np.array(len(attns))
        out.update({f"{tag}.{k}": v for k, v in pack_draws(log).items()})
    # a PerceiveEncoder with output_attention (cross_modal_transformer.py:385-433)
    torch.manual_seed(0)
    enc = REF.cmt.PerceiveEncoder(24, 16, 7, factor=5, d_model=128, n_heads=8, layers=3, dropout=0.0, output_attention=True)
    load_synth(enc)
    enc.eval()
    xe = torch.randn(2, 21, 24, generator=g)
    log = []
    torch.manual_seed(RSEED)
    with record_randint(log), torch.no_grad():
        y, attns = enc(xe)
    out["perceive.x"], out["perceive.y"], out["perceive.n"] = xe, y, np.array(len(attns))
    for i, a in enumerate(attns):
        out[f"perceive.attn{i}"] = a
    out.update({f"perceive.{k}": v for k, v in pack_draws(log).items()})
    # Routeformer._forward's (output, attention)
    c = dict(presets.case("c1_default"))
    gps_cfg, rf_cfg = presets.build_configs(c, REF.gps.GPSBackboneConfig, REF.cfg.RouteformerConfig, REF.vbc.VideoBackboneConfig)
    rf_cfg.output_attention = True
    rf_cfg.gps_backbone_config.output_attention = True
    torch.manual_seed(0)
    model = REF.rf.Routeformer(rf_cfg, gps_backbone=REF.gps.Informer, video_backbone=None)
    load_synth(model)
    model.eval()
    item = synthetic.synth_item(c["B"], c["T"], c["P"], DSEED, c["H"], c["W"], streams=c["streams"], gaze=c["gaze"])
    log = []
    torch.manual_seed(RSEED)
    with record_randint(log), torch.no_grad():
        motion, visual = model.preprocess_batch(item["train"])[:2]
        y, attns = model._forward(motion, visual)
    out["model.y"], out["model.n"] = y, np.array(len(attns))
    for i, a in enumerate(attns):
        out[f"model.attn{i}"] = a
    out.update({f"model.{k}": v for k, v in pack_draws(log).items()})
    save("attn_out", **out)


GENERATORS = {"attn_out": gen_attn_out, "dropout": gen_dropout, "widen": gen_widen, "attention": gen_attention, "blocks": gen_blocks, "informer": gen_informer,
              "hrnet": gen_hrnet, "helpers": gen_helpers}

if __name__ == "__main__":
    torch.set_num_threads(8)
    todo = sys.argv[1:] or list(GENERATORS) + list(presets.CASES)
    for t in todo:
        if t in GENERATORS:
            GENERATORS[t]()
        else:
            gen_case(t)
