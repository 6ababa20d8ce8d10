"""GPU parity tests at block and whole-model level: the HIP-backed product modules against
(i) the reference's golden vectors and (ii) the CPU oracle on the same seeded inputs.

Tolerance (stated per BASELINE.json north_star): trajectories within 1e-3 (fp32 mode) / 1e-2
(bf16-MFMA mode) of the reference, measured as max|err| / max(1, max|ref|)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import RSEED, build_product_model, case_item, draws, golden, rel_err, t

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import routeformer_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL_F32, TOL_BF16 = 1e-3, 1e-2
# Per-parameter agreement of bf16-mode gradients with the fp32 CPU oracle.  What bounds it is the arithmetic mode itself,
# not a kernel: profiles/r03/bf16_grad_report.txt (tools/bf16_grad_report.py) shows the same figures for the fused
# stacks, the layer-by-layer kernels and the plain tiled GEMMs (whole-gradient cosine 0.985 / 0.985 / 0.983 on c2_paper,
# 1.00000 in fp32 mode): bf16 operand rounding moves ReLU masks, MaxPool arg-maxes of the distilling layers and the
# nearly-uniform softmax rows of the first Informer layer (dS cancels 3-4 digits: dWq / dWk there carry 25 % norm error
# at 1e-6 of the gradient energy).  The reference itself trains with torch.set_float32_matmul_precision("medium")
# (full_comparison.py:48), i.e. bf16 operand rounding inside its fp32 matmuls on a GPU.  Observed worst cases over the
# three fixtures: cosine 0.956, norm error 0.24 (the two ill-conditioned parameters), whole-gradient cosine 0.985.
BF16_GRAD_COS, BF16_GRAD_NORM, BF16_GRAD_WHOLE = 0.94, 0.25, 0.975


@pytest.fixture(autouse=True)
def _reset():
    from routeformer_amd import kernels as K
    from routeformer_amd.models.blocks import SAMPLER
    K.set_precision("f32")
    SAMPLER.replay, SAMPLER.log = None, None
    SAMPLER.drop_static()
    K.TOPS.record, K.TOPS.forced = None, None
    yield
    K.set_precision("f32")
    SAMPLER.replay, SAMPLER.log = None, None
    SAMPLER.drop_static()  # a failed test must not leave the process-wide sampler in graph mode for the next one
    K.TOPS.record, K.TOPS.forced = None, None


def _to_dev(batch):
    return {k: v.to(DEV) for k, v in batch.items()}


def _load(module, seed=7):
    from routeformer_amd import synthetic
    module.load_state_dict(synthetic.synth_state_dict(module.state_dict(), seed))
    return module.to(DEV)


def _check_grads(G, key, named_params, tol, oracle_grads=None, tol_full=None, loose=(), tol_loose=0.0):
    """Parameter gradients against the reference's record (tests/golden/make_golden.py: L2 norm AND sum of every
    parameter gradient, a handful of full tensors) and -- ``oracle_grads`` -- against the CPU oracle's autograd
    gradients ELEMENT BY ELEMENT for every parameter (the oracle's gradients themselves match the reference's norms and
    sums to <= 6e-4, tests/test_oracle_golden.py::test_model_train_step): a permutation or a sign
    error inside a large parameter cannot hide behind its norm.  Errors are relative to the parameter's own largest
    gradient element (floored at 1e-3 of the largest gradient norm: e.g. key-projection biases have an exactly-zero true
    gradient)."""
    names = [str(s) for s in G[key + "grad_names"]]
    stats = G[key + "grad_stats"]
    bad = []
    floor = 1e-3 * float(stats[:, 0].max())
    for n, (nrm, total) in zip(names, stats):
        g = named_params[n].grad
        got = 0.0 if g is None else float(g.double().norm())
        got_sum = 0.0 if g is None else float(g.double().sum())
        if abs(got - nrm) > tol * max(floor, nrm):
            bad.append((n, "norm", got, nrm))
        # |sum| <= sqrt(numel) * norm: compared on the norm's scale
        if abs(got_sum - total) > tol * max(floor, nrm) * max(1.0, float(np.sqrt(named_params[n].numel())) / 8):
            bad.append((n, "sum", got_sum, total))
    assert not bad, bad[:8]
    for f in G.files:
        if f.startswith(key + "grad::"):
            n = f[len(key + "grad::"):]
            assert rel_err(named_params[n].grad, G[f]) < tol, n
    if oracle_grads is not None:
        tol_full = tol if tol_full is None else tol_full
        gmax = max(float(g.abs().max()) for g in oracle_grads.values())
        rows = []
        for n, go in oracle_grads.items():
            g = named_params[n].grad
            assert g is not None and g.shape == go.shape, n
            scale = max(float(go.abs().max()), 1e-3 * gmax)
            rows.append((float((g.detach().cpu().double() - go.double()).abs().max()) / scale, n))
        rows.sort(reverse=True)
        print(f"[{key}] full-tensor gradient error vs oracle autograd, worst five of {len(rows)}: "
              + ", ".join(f"{n} {e:.2e}" for e, n in rows[:5]))
        bad = [(e, n) for e, n in rows if e >= (tol_loose if n in loose else tol_full)]
        assert not bad, bad[:5]


def _oracle_train_step_grads(cfg, sd, item, epoch, seed=RSEED, drop=None):
    """The CPU oracle's train step WITH autograd: -> (results, {parameter name: gradient}, top-u selections)."""
    sdg = {}
    for k, v in sd.items():
        v = v.clone()
        if (v.is_floating_point() and not k.startswith("video_backbone.") and "running_" not in k
                and not k.endswith(".pe")):
            v.requires_grad_(True)
        sdg[k] = v
    torch.manual_seed(seed)
    orc = O.OracleRouteformer(cfg, sdg, training=True, **({"drop": drop} if drop is not None else {}))
    res = orc.train_step(item, epoch)
    res["loss"].backward()
    grads = {k: v.grad.detach() for k, v in sdg.items() if v.requires_grad and v.grad is not None}
    return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in res.items()}, grads, orc.idx.tops


def test_perceive_blocks_golden():
    from routeformer_amd.models.blocks import SAMPLER, PerceiveDecoder, PerceiveEncoder
    G = golden("blocks")
    enc = _load(PerceiveEncoder(in_channels=240, out_channels=64, out_len=1, n_heads=8, layers=2, d_ff=64, dropout=0.0))
    SAMPLER.replay = draws(G, "enc.")
    y = enc(t(G["enc.x"]).to(DEV))
    assert rel_err(y, G["enc.y"]) < 1e-4
    y.square().sum().backward()
    _check_grads(G, "enc.", dict(enc.named_parameters()), 1e-3)

    enc2 = _load(PerceiveEncoder(in_channels=2, out_channels=64, out_len=40, n_heads=8, layers=2, d_ff=256, dropout=0.0))
    SAMPLER.replay = None
    torch.manual_seed(RSEED)  # host RNG reproduces the reference's draws
    assert rel_err(enc2(t(G["enc2.x"]).to(DEV)), G["enc2.y"]) < 1e-4

    dec = _load(PerceiveDecoder(query_channels=64, value_channels=64, out_channels=64, out_len=40, dropout=0.0,
                                d_ff=256, n_heads=8, layers=2, mix=False))
    SAMPLER.replay = draws(G, "dec.")
    mem, qry = t(G["dec.mem"]).to(DEV).requires_grad_(), t(G["dec.qry"]).to(DEV).requires_grad_()
    yd = dec(mem, qry)
    assert rel_err(yd, G["dec.y"]) < 1e-4
    yd.square().sum().backward()
    assert rel_err(mem.grad, G["dec.dmem"]) < 1e-3 and rel_err(qry.grad, G["dec.dqry"]) < 1e-3
    _check_grads(G, "dec.", dict(dec.named_parameters()), 1e-3)


@pytest.mark.parametrize("tag,preset,B,T,P,cin", [("tiny", "GPS_TINY", 3, 20, 10, 69), ("default", "GPS_DEFAULT", 4, 10, 15, 5),
                                                   ("paper", "GPS_PAPER", 2, 40, 30, 69)])
def test_informer_golden(tag, preset, B, T, P, cin):
    from routeformer_amd import presets
    from routeformer_amd.models.blocks import SAMPLER
    from routeformer_amd.models.gps_backbone import GPSBackboneConfig, Informer
    G = golden("informer")
    x = t(G[tag + ".x"]).to(DEV)
    for smart in (False, True):
        gcfg = GPSBackboneConfig(seq_len=T, label_len=T, pred_len=P, **getattr(presets, preset))
        gcfg.output_attention, gcfg.smart_decoder, gcfg._enc_in, gcfg._c_out = False, smart, cin, cin - 3
        for mode in ("eval", "train"):
            key = f"{tag}.{'smart' if smart else 'vanilla'}.{mode}."
            if key + "y" not in G.files:
                continue
            net = _load(Informer(gcfg))
            net.train(mode == "train")
            SAMPLER.replay = draws(G, key)
            y = net(x)
            assert rel_err(y, G[key + "y"]) < 2e-4, key
            if mode == "train":
                y.square().mean().backward()
                _check_grads(G, key, dict(net.named_parameters()), 2e-3)
                sd = net.state_dict()
                assert rel_err(sd["encoder.conv_layers.0.norm.running_mean"], G[key + "bn0_running_mean"]) < 1e-4
                assert rel_err(sd["encoder.conv_layers.0.norm.running_var"], G[key + "bn0_running_var"]) < 1e-4


def test_output_attention_golden():
    """``output_attention=True`` (cross_modal_transformer.py:66,134-138,430; gps_backbone Informer.py:164, Transformer.py:138;
    routeformer.py:237-252): the dense maps handed back next to the outputs, against the reference's own -- uniform 1 / L rows
    with the SELECTED queries' softmax rows scattered in (so the kernel's top-u selection is checked too), the dense softmax of
    FullAttention, and Routeformer._forward's second return value."""
    from routeformer_amd import presets, synthetic
    from routeformer_amd.models import Routeformer, RouteformerConfig
    from routeformer_amd.models.blocks import SAMPLER, PerceiveEncoder
    from routeformer_amd.models.gps_backbone import GPSBackboneConfig, Informer, Transformer
    from routeformer_amd.models.video_backbone import VideoBackboneConfig
    G = golden("attn_out")
    x = t(G["x"]).to(DEV)
    B, T, P, cin = 3, 20, 10, 69

    def check(tag, y, attns, tol=2e-4):
        assert len(attns) == int(G[tag + ".n"])
        assert rel_err(y, G[tag + ".y"]) < tol, tag
        for i, a in enumerate(attns):
            want = t(G[f"{tag}.attn{i}"])
            assert tuple(a.shape) == tuple(want.shape), (tag, i, a.shape, want.shape)
            assert float((a.sum(-1) - 1).abs().max()) < 1e-5  # rows are distributions
            assert rel_err(a, want) < tol, (tag, i)

    for tag, cls in (("informer", Informer), ("transformer", Transformer)):
        gcfg = GPSBackboneConfig(seq_len=T, label_len=T, pred_len=P, **presets.GPS_TINY)
        gcfg.output_attention, gcfg.smart_decoder, gcfg._enc_in, gcfg._c_out = True, True, cin, cin - 3
        net = _load(cls(gcfg)).eval()
        SAMPLER.replay = draws(G, tag + ".")
        with torch.no_grad():
            y, attns = net(x)
        check(tag, y, attns)
    enc = _load(PerceiveEncoder(24, 16, 7, factor=5, d_model=128, n_heads=8, layers=3, dropout=0.0, output_attention=True)).eval()
    SAMPLER.replay = draws(G, "perceive.")
    with torch.no_grad():
        y, attns = enc(t(G["perceive.x"]).to(DEV))
    check("perceive", y, attns)
    # the whole model: _forward returns (output, attention) as routeformer.py:252 does
    c = dict(presets.case("c1_default"))
    gps_cfg, cfg = presets.build_configs(c, GPSBackboneConfig, RouteformerConfig, VideoBackboneConfig)
    cfg.output_attention = True
    cfg.gps_backbone_config.output_attention = True
    model = _load(Routeformer(cfg, gps_backbone=Informer, video_backbone=None)).eval()
    item = synthetic.synth_item(c["B"], c["T"], c["P"], 11, c["H"], c["W"], streams=c["streams"], gaze=c["gaze"])
    SAMPLER.replay = draws(G, "model.")
    with torch.no_grad():
        motion, visual = model.preprocess_batch(_to_dev(item["train"]))
        y, attns = model._forward(motion, visual)
        check("model", y, attns)
        SAMPLER.replay = draws(G, "model.")
        pos = model(_to_dev(item["train"]))  # forward() itself is unchanged by the flag (routeformer.py:157)
    assert pos.shape[-1] == 2 and bool(torch.isfinite(pos).all())
    SAMPLER.replay = None


CASES = ["c1_default", "c1_paper", "c1_recursive", "c1_noise", "c2_small", "c4_small", "c5_small", "ar_small", "c2_paper"]


def _first_flip(tops_gpu, src):
    """Compare the top-u selections call by call.  Returns None when all agree, else
    (call index, #differing (b,h) rows, largest oracle margin among the differing rows of that call)."""
    for i, (a, b) in enumerate(zip(tops_gpu, src.tops)):
        diff = (a.cpu().long() != b).any(dim=-1)  # (B,H)
        if diff.any():
            return i, int(diff.sum()), float(src.margins[i][diff].max())
    return None


def _oracle_eval(cfg, sd, batch, replay):
    src = O.IndexSource(replay)
    with torch.no_grad():
        out = O.OracleRouteformer(cfg, sd, training=False, idx=src).forward(batch)
    return (out if isinstance(out, tuple) else (out, None)), src


@pytest.mark.parametrize("name", CASES)
def test_model_eval_forward_golden(name):
    """Whole Routeformer forward (eval) vs the reference's trajectories.

    ProbSparse's top-u query selection is discontinuous: a near-tie in the sparsity measure can resolve
    differently under a different fp32 summation order and move the output by O(1e-2) ("hard
    parts").  So: (a) free-running run -- host RNG draws must equal the reference's, and if any
    selection differs from the oracle's, the FIRST difference must be a near-tie (relative margin
    < 1e-4); (b) run with the oracle's selections imposed -- trajectories within 1e-3 of the reference."""
    from routeformer_amd import kernels as K
    from routeformer_amd.models.blocks import SAMPLER
    model, cfg, sd, c = build_product_model(name, DEV)
    G = golden(name)
    item = case_item(c)
    batch = _to_dev(item["train"])
    rec = draws(G, "eval.")
    (pos_o, vis_o), src = _oracle_eval(cfg, sd, item["train"], list(rec))
    model.eval()
    # (b) oracle selections imposed: strict parity
    K.TOPS.forced = [t_.clone() for t_ in src.tops]
    torch.manual_seed(RSEED)
    with torch.no_grad():
        out = model(batch)
    assert not K.TOPS.forced, "not all forced selections were consumed"
    K.TOPS.forced = None
    pos, vis = out if isinstance(out, tuple) else (out, None)
    assert pos.shape == G["eval.future_gps"].shape and pos.dtype == torch.float32
    assert rel_err(pos, pos_o) < 3e-4 and rel_err(pos, G["eval.future_gps"]) < TOL_F32, name
    if vis is not None:
        assert rel_err(vis, G["eval.future_vis"]) < TOL_F32
    # (a) free running
    SAMPLER.log, K.TOPS.record = [], []
    torch.manual_seed(RSEED)
    with torch.no_grad():
        out = model(batch)
    pos = out[0] if isinstance(out, tuple) else out
    assert len(SAMPLER.log) == len(rec), "host RNG call count differs from the reference"
    for a, b in zip(SAMPLER.log, rec):
        assert torch.equal(a, b), "host RNG draw order differs from the reference (SURVEY Appendix D)"
    flip = _first_flip(K.TOPS.record, src)
    SAMPLER.log, K.TOPS.record = None, None
    if flip is None:
        assert rel_err(pos, G["eval.future_gps"]) < TOL_F32, name
    else:
        print(f"[{name}] selection flip at ProbSparse call {flip[0]}: {flip[1]} rows, margin/|qk| {flip[2]:.2e}; "
              f"free-running rel err {rel_err(pos, G['eval.future_gps']):.2e}")
        assert flip[2] < 2e-4, f"selection differs from the oracle away from a rounding-level tie: {flip}"
        assert rel_err(pos, G["eval.future_gps"]) < 0.1


@pytest.mark.parametrize("name", ["c2_small", "c4_small", "c2_paper"])
def test_model_eval_forward_bf16(name):
    """bf16 matrix-core mode: trajectories within 1e-2 of the reference with the oracle's top-u
    selections imposed (the selection is discontinuous -- see above); free-running error is reported."""
    from routeformer_amd import kernels as K
    model, cfg, sd, c = build_product_model(name, DEV)
    G = golden(name)
    item = case_item(c)
    (pos_o, _), src = _oracle_eval(cfg, sd, item["train"], draws(G, "eval."))
    K.set_precision("bf16")
    model.eval()
    K.TOPS.forced = [t_.clone() for t_ in src.tops]
    torch.manual_seed(RSEED)
    with torch.no_grad():
        out = model(_to_dev(item["train"]))
    K.TOPS.forced = None
    pos = out[0] if isinstance(out, tuple) else out
    forced = rel_err(pos, G["eval.future_gps"])
    print(f"[{name}] bf16 rel err with imposed selections {forced:.2e} (bound {TOL_BF16:.0e})")
    assert forced < TOL_BF16, name
    torch.manual_seed(RSEED)
    with torch.no_grad():
        out = model(_to_dev(item["train"]))
    free = rel_err(out[0] if isinstance(out, tuple) else out, G["eval.future_gps"])
    # the same quantity for the REFERENCE's own GPU training precision (oracle.ARITH = "medium": bf16-rounded matmul
    # operands per torch.set_float32_matmul_precision("medium"), full_comparison.py:48, TF32 convolutions) on this case
    try:
        O.ARITH = "medium"
        (pos_m, _), _ = _oracle_eval(cfg, sd, item["train"], draws(G, "eval."))
    finally:
        O.ARITH = None
    free_ref = rel_err(pos_m, G["eval.future_gps"])
    print(f"[{name}] bf16 free-running rel err {free:.2e}; the fp32 oracle in the reference's 'medium' precision: {free_ref:.2e}")
    # free-running: the selections leave the fp32 path.  Measured in round 4 (bench.py ade_vs_cpu_ref, 16 samples x 2 seeds,
    # profiles/r04/bench_n1_default_20_medium_leg.json): the product's bf16 mode flips 42 % of the teacher-forced selections
    # (12.7 % in the first frame-encoder layer) and ends at median 3.1e-3 / p90 9.0e-3 / max 2.1e-2 of the trajectory
    # scale; the CPU oracle run in the reference's own GPU precision mode flips 34 % (10.9 %) and ends at 5.3e-3 / 1.1e-2 /
    # 2.0e-2 -- the product sits inside what the reference's training arithmetic does to itself.  Bound = 1.5 x the larger of
    # the two observed maxima (was 5e-2).
    assert free < 3e-2


# (norm / sum tolerance, full-tensor tolerance, names with their own full-tensor tolerance, that tolerance)
_QK0 = ("gps_backbone.encoder.attn_layers.0.attention.query_projection.weight",
        "gps_backbone.encoder.attn_layers.0.attention.key_projection.weight")
TRAIN_TOL = {"c2_paper": (1.5e-2, 3e-2, (), 0.0), "c1_paper": (5e-3, 2e-3, (), 0.0), "c5_small": (1.5e-2, 1e-3, _QK0, 3e-2)}


@pytest.mark.parametrize("name", ["c1_default", "c1_recursive", "c1_noise", "c2_small", "c4_small", "c2_paper", "c5_small",
                                  "c1_paper"])
def test_model_train_step_golden(name):
    """The train-step recipe (loss, ADE, FDE, gradients) vs the reference, epochs 0 and 10, with the
    oracle's top-u selections imposed (train-mode oracle run with the same seed)."""
    from routeformer_amd import kernels as K
    from routeformer_amd.engine import train_step_losses
    model, cfg, sd, c = build_product_model(name, DEV)
    G = golden(name)
    item = case_item(c)
    item_d = {"train": _to_dev(item["train"]), "target": _to_dev(item["target"])}
    for epoch in (0, 10):
        key = f"train{epoch}."
        if key + "loss" not in G.files:
            continue
        if name == "c2_paper" and epoch == 0:
            continue  # keep the suite short: the dense-loss epoch covers a superset of the graph
        _, ograds, otops = _oracle_train_step_grads(cfg, sd, item, epoch)
        model.load_state_dict(sd)
        model.train()
        model.zero_grad(set_to_none=True)
        K.TOPS.forced = [t_.clone() for t_ in otops]
        torch.manual_seed(RSEED)
        res = train_step_losses(model, item_d, epoch)
        assert not K.TOPS.forced
        K.TOPS.forced = None
        assert rel_err(res["future_gps"], G[key + "future_gps"]) < TOL_F32
        for k in ("loss", "traj_loss", "ade", "fde"):
            ref = float(G[key + k])
            assert abs(float(res[k]) - ref) < 1e-3 * max(1.0, abs(ref)), (k, float(res[k]), ref)
        if "dense_loss" in res:
            assert abs(float(res["dense_loss"]) - float(G[key + "dense_loss"])) < 1e-3
            assert rel_err(res["target_vis"], G[key + "target_vis"]) < TOL_F32
        res["loss"].backward()
        # 5e-3 on per-parameter gradient norms; observed ~1e-5 everywhere except dW_q / dW_k of an attention layer
        # whose softmax rows are nearly uniform (first Informer layer of c2_paper): there dS = P * (dP - sum(P dP))
        # cancels 3-4 leading digits, so ANY fp32 implementation (the reference's included) carries ~1e-3 relative
        # noise in dQ / dK that moves with the summation order (measured 1.6e-3 .. 6.2e-3 for two thread counts of
        # the same kernel, bit-identical on well-conditioned random inputs; table: tools/dbg_grads.py) -> 1.5e-2 there
        # full tensors vs oracle autograd: observed <= 5e-5 on the small cases for EVERY parameter (bound 2e-4).  c2_paper
        # (d_model 832, 6 distilling layers) has two rounding-level discrete events: (i) the two ill-conditioned
        # parameters above, where the reference's own fp32 CPU gradient and the oracle's differ by 1.5e-2 already
        # (tests/test_oracle_golden.py holds c2_paper's norms to that); (ii) a MaxPool arg-max of the second distilling
        # layer that sits on a tie: perturbing the trunk tokens by 5e-7 (the fused vs the unfused concat + pool kernels,
        # tools/dbg_fuse_determinism.py) moves a handful of gradients from <= 3e-4 to 0.7-1.7e-2 of their largest element,
        # deterministically run to run.  Bound 3e-2 there: still far below a sign / permutation error (O(1)).
        # c5_small (round 4: fusion length 320, gaze length 80, decoder length 105 -- the row-tiled stack's shapes): its
        # first Informer layer has the same nearly-uniform softmax rows (80 keys); the reference and the oracle differ by
        # 3.7e-3 on those two weights' norms (tests/test_oracle_golden.py); observed here 2.8e-3 / 2.3e-3 on them, 5.6e-4 on
        # the GPS value embedding (the one parameter upstream of that layer's q / k: it inherits a share of their noise, as in
        # c2_paper), <= 7.3e-5 on the other 265 parameters -- bounds 3e-2 / 1e-3.
        # c1_paper (round 4: the d_model-832 backbone alone, B = 4, 10 -> 15 steps).
        tol, tol_full, loose, tol_loose = TRAIN_TOL.get(name, (5e-3, 2e-4, (), 0.0))
        _check_grads(G, key, dict(model.named_parameters()), tol, oracle_grads=ograds, tol_full=tol_full, loose=loose,
                     tol_loose=tol_loose)


def _grad_agreement(named_params, oracle_grads):
    """Per parameter: (cosine with the oracle gradient, relative norm error), for parameters whose reference gradient
    is not negligible (>= 1e-3 of the largest gradient norm: a key-projection bias has an exactly-zero true gradient)."""
    gmax = max(float(g.double().norm()) for g in oracle_grads.values())
    rows = []
    for n, go in oracle_grads.items():
        go = go.double().reshape(-1)
        g = named_params[n].grad.detach().cpu().double().reshape(-1)
        no, ng = float(go.norm()), float(g.norm())
        if no < 1e-3 * gmax:
            continue
        rows.append((float(g @ go) / max(ng * no, 1e-300), abs(ng - no) / no, n))
    return rows


@pytest.mark.parametrize("name", ["c2_small", "c4_small", "c2_paper", "c5_small"])
def test_model_train_step_bf16(name):
    """The arithmetic mode bench.py times -- bf16 matrix-core operands, fused encoder stacks forward AND backward,
    gradient sinks, grouped weight gradients (``TrainEngine._fwd_bwd``) -- against the reference's train step
    (epoch 10: both losses on) with the oracle's top-u selections imposed: trajectories <= 1e-2 (north_star's bf16
    tolerance), loss / ADE / FDE <= 1e-2 relative, and EVERY parameter gradient compared with the CPU oracle's autograd
    gradient (cosine and norm); the worst five of each are printed."""
    from routeformer_amd import kernels as K
    from routeformer_amd.engine import TrainEngine
    model, cfg, sd, c = build_product_model(name, DEV)
    G = golden(name)
    item = case_item(c)
    item_d = {"train": _to_dev(item["train"]), "target": _to_dev(item["target"])}
    epoch, key = 10, "train10."
    ores, ograds, otops = _oracle_train_step_grads(cfg, sd, item, epoch)
    K.set_precision("bf16")
    calls, real = [], K._seqstack_bwd_launch
    eng = TrainEngine(model)
    model.train()
    K.TOPS.forced = [t_.clone() for t_ in otops]
    K._seqstack_bwd_launch = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    torch.manual_seed(RSEED)
    try:
        res = eng._fwd_bwd(item_d, epoch)
        torch.cuda.synchronize()
        assert not K.TOPS.forced, "not all imposed selections were consumed"
    finally:
        K.TOPS.forced, K._seqstack_bwd_launch = None, real
    assert calls, "the fused encoder-stack backward was not taken"
    e_pos = rel_err(res["future_gps"], G[key + "future_gps"])
    print(f"[{name}] bf16 train step: future_gps rel err {e_pos:.2e}")
    assert e_pos < TOL_BF16
    assert rel_err(res["target_vis"], G[key + "target_vis"]) < 5 * TOL_BF16
    for k in ("loss", "traj_loss", "dense_loss", "ade", "fde"):
        ref = float(G[key + k])
        assert abs(float(res[k]) - ref) < 1e-2 * max(1.0, abs(ref)), (k, float(res[k]), ref)
    rows_all = _grad_agreement(dict(model.named_parameters()), ograds)
    # gradients w.r.t. attention LOGITS (query / key projections) go through dS = P * (dP - sum(P dP)): with the nearly
    # uniform softmax rows of these random-init models the difference cancels 3-4 digits, so bf16-level noise in dP is
    # amplified to O(1) of these (tiny) gradients -- run-to-run they move between cosine 0.6 and 0.96 for the first
    # Informer layer (the fp32 reference itself carries 1.5e-2 there, see test_model_train_step_golden).  They are
    # reported, and they count in the whole-gradient figures, but the per-parameter bounds cover all the others.
    # c5_small (round 4; 80 keys in the first Informer layer): there those two gradients are pure noise in this mode (cosine
    # 0.05 / 0.12 at 0.7 % of the gradient energy; fp32 mode: 2.8e-3, test_model_train_step_golden) and the one parameter
    # UPSTREAM of that layer's q / k, the GPS value embedding, carries their noise on top of its own gradient: observed
    # cosine 0.880, norm error 0.109 = 1 / sqrt(1 + r^2) and sqrt(1 + r^2) - 1 for a noise-to-gradient ratio r = 0.54.  It is
    # held to cosine >= 0.75 (a sign or permutation error gives <= 0) outside the common bound.
    upstream = ("gps_backbone.enc_embedding.value_embedding.tokenConv.weight",) if name == "c5_small" else ()
    logit = lambda n: ".query_projection." in n or ".key_projection." in n or n in upstream  # noqa: E731
    for c_, e, n in rows_all:
        if n in upstream:
            assert c_ >= 0.75, (n, c_, e)
    rows = [r for r in rows_all if not logit(r[2])]
    soft = sorted(r for r in rows_all if logit(r[2]))
    print(f"[{name}] query / key projection gradients (excluded from the per-parameter bounds), worst: "
          + ", ".join(f"{n} cos {c_:.3f} norm err {e:.2f}" for c_, e, n in soft[:3]))
    by_cos, by_norm = sorted(rows), sorted(rows, key=lambda r: -r[1])
    print(f"[{name}] bf16 gradients vs oracle autograd over {len(rows)} parameters: worst cosines "
          + ", ".join(f"{n} {c_:.4f}" for c_, _, n in by_cos[:5]) + "; worst norm errors "
          + ", ".join(f"{n} {e:.3f}" for _, e, n in by_norm[:5]))
    flat_o = torch.cat([ograds[n].double().reshape(-1) for n in sorted(ograds)])
    flat_g = torch.cat([dict(model.named_parameters())[n].grad.detach().cpu().double().reshape(-1) for n in sorted(ograds)])
    whole = float(flat_g @ flat_o / (flat_g.norm() * flat_o.norm()))
    print(f"[{name}] whole-gradient cosine {whole:.5f}, norm ratio {float(flat_g.norm() / flat_o.norm()):.4f}")
    assert whole > BF16_GRAD_WHOLE and abs(float(flat_g.norm() / flat_o.norm()) - 1) < 2e-2
    assert by_cos[0][0] >= BF16_GRAD_COS, by_cos[:5]
    assert by_norm[0][1] <= BF16_GRAD_NORM, by_norm[:5]


def test_model_vs_oracle_seeded():
    """Same seed, no recorded indices: product (HIP) vs CPU oracle on c2_small in train mode."""
    from routeformer_amd import kernels as K
    model, cfg, sd, c = build_product_model("c2_small", DEV)
    item = case_item(c)
    torch.manual_seed(4321)
    src = O.IndexSource()
    with torch.no_grad():
        pos_o, vis_o = O.OracleRouteformer(cfg, sd, training=True, idx=src).forward(item["train"])
    model.train()
    K.TOPS.forced = [t_.clone() for t_ in src.tops]
    torch.manual_seed(4321)
    pos_d, vis_d = model(_to_dev(item["train"]))
    K.TOPS.forced = None
    assert rel_err(pos_d, pos_o) < TOL_F32 and rel_err(vis_d, vis_o) < TOL_F32


def test_graphed_step_matches_eager():
    """HIP-graph replay of forward+backward, with the conv trunk of the NEXT batch pipelined under the
    current step, == eager launches: same losses on an alternating two-batch stream, same gradients, and the
    host RNG is consumed identically (reference draw order) in both modes."""
    from routeformer_amd import synthetic
    from routeformer_amd.engine import GraphedTrainEngine, TrainEngine
    from routeformer_amd.models.blocks import SAMPLER
    results = {}
    from routeformer_amd import engine as E
    for mode in ("eager", "graph", "graph_nolookahead", "graph_deferred", "graph_deferred_split", "graph_deferred_early"):
        model, cfg, sd, c = build_product_model("c2_small", DEV)
        items = []
        for seed in (11, 12):
            it = synthetic.synth_item(c["B"], c["T"], c["P"], seed, c["H"], c["W"], streams=c["streams"], gaze=c["gaze"])
            items.append({"train": _to_dev(it["train"]), "target": _to_dev(it["target"]), "id": seed})
        if mode == "eager":
            eng = TrainEngine(model, lr=1e-3)
        else:
            # deferred: clip + AdamW of step k replayed at the start of step k+1 (backbone share on a side stream)
            eng = GraphedTrainEngine(model, lr=1e-3, defer_update=mode.startswith("graph_deferred"))
            eng.split = mode.endswith("_split")  # two-graph step (the N > 1 form)
            was = E.EARLY_SUMSQ  # "_early": the backbone's share of the clip norm summed mid-backward (measurement switch)
            E.EARLY_SUMSQ = mode.endswith("_early")
            try:
                eng.capture(items[0], epoch=10)
            finally:
                E.EARLY_SUMSQ = was
            assert eng._early_sumsq == mode.endswith("_early")
        torch.manual_seed(99)
        SAMPLER.log = []
        losses = []
        for i in range(4):
            nxt = items[(i + 1) % 2] if mode in ("graph", "graph_deferred", "graph_deferred_early") else None
            losses.append(float(eng.step(items[i % 2], epoch=10, next_item=nxt)["loss"]))
        torch.cuda.synchronize()
        grads = eng.reducer.flat_grad.clone()
        if mode.startswith("graph_deferred"):
            assert eng._pending is not None
            # the clip norm's early partial sums (the GPS backbone's range, summed mid-backward on a side stream:
            # engine._backward_gps_first) cover every gradient of that range -- none was still queued when they were taken
            for a, b, o, early in (eng.opt._ss_plan if eng._early_sumsq else ()):
                if early:
                    from routeformer_amd import _hip
                    k = int(_hip.lib().rf_sumsq_parts(b - a))
                    got = float(eng.opt.sumsq[o:o + k].double().sum())
                    want = float(grads[a:b].double().pow(2).sum())
                    assert want > 0 and abs(got - want) < 1e-5 * want, (mode, got, want)
            eng.flush()  # the last step's update is still pending
            torch.cuda.synchronize()
            assert eng._pending is None and eng.opt.t == 4
        results[mode] = (losses, eng.reducer.flat_param.clone(), [t_.clone() for t_ in SAMPLER.log],
                         torch.get_rng_state(), grads)
        SAMPLER.log = None
        SAMPLER.drop_static()
    le, pe, de, re_, ge = results["eager"]
    assert abs(le[0] - le[1]) > 1e-6, "the two batches should differ"
    for mode in ("graph", "graph_nolookahead", "graph_deferred", "graph_deferred_split", "graph_deferred_early"):
        lg, pg, dg, rg, gg = results[mode]
        assert len(de) == len(dg) and all(torch.equal(a, b) for a, b in zip(de, dg)), mode
        assert torch.equal(re_, rg), "host RNG state diverged between eager and graph mode"
        # (after two or three AdamW steps at lr 1e-3 the losses carry the optimizer's amplification of the fp32
        # atomics' summation-order noise: 2.1e-4 seen between two runs of the SAME mode)
        assert all(abs(a - b) < 5e-4 * max(1.0, abs(a)) for a, b in zip(le, lg)), (mode, le, lg)
        # gradients of the 4th step, i.e. after three AdamW updates: rounding-level noise on near-zero gradients becomes
        # +-lr parameter steps, which the next backward sees (2e-3 typical, 7e-3 observed once for a deferred run)
        assert rel_err(gg, ge) < 1e-2, mode
        # AdamW turns rounding-level noise on near-zero gradients into +-lr steps, hence the looser bound here
        assert rel_err(pg, pe) < 4 * 3 * 1e-3, mode


@pytest.mark.parametrize("name", ["c1_paper", "c2_paper"])
def test_slab_carried_gradients_change_nothing(name):
    """kernels.LAZY (round 4): where a split-K product is followed by a kernel that can sum the slabs on load -- the FFN's input
    gradient into the LayerNorm backward in front of it, the next layer's q | k | v input gradient into the distilling tail's
    backward, the distilling convolution's product into the BatchNorm tail -- the slab-sum launch is skipped and an unwritten
    placeholder travels through autograd.  Same engine step with the paths on and off, paper-size GPS backbone (d_model 832:
    the small presets have no split-K products): the paths are really taken, every gradient agrees to the noise of the fp32
    atomics, and nothing is left unconsumed."""
    from routeformer_amd import kernels as K
    from routeformer_amd.engine import TrainEngine
    K.set_precision("bf16")
    out = {}
    was = K.LAZY_DX, K.LAZY_BN_FWD, K.LAZY_ATTN
    try:
        for lazy in (False, True):
            K.LAZY_DX = K.LAZY_BN_FWD = K.LAZY_ATTN = lazy  # (the attention form is off by default: no faster; covered here)
            model, cfg, sd, c = build_product_model(name, DEV)
            item = case_item(c)
            item_d = {"train": _to_dev(item["train"]), "target": _to_dev(item["target"])}
            eng = TrainEngine(model)
            model.train()
            torch.manual_seed(RSEED)
            n0 = K.LAZY_COUNT[0]
            res = eng._fwd_bwd(item_d, 10)
            torch.cuda.synchronize()
            assert not K.LAZY
            out[lazy] = (float(res["loss"].detach()), eng.reducer.flat_grad.clone(), K.LAZY_COUNT[0] - n0,
                         {n: b.clone() for n, b in model.named_buffers() if "running" in n})
    finally:
        K.LAZY_DX, K.LAZY_BN_FWD, K.LAZY_ATTN = was
    (l0, g0, n_off, b0), (l1, g1, n_on, b1) = out[False], out[True]
    print(f"[{name}] slab-carried tensors per step: {n_on}; loss {l0:.6f} / {l1:.6f}; gradient buffer rel diff {rel_err(g1, g0):.2e}")
    assert n_off == 0 and n_on >= 6, (n_off, n_on)
    assert abs(l0 - l1) <= 1e-6 * max(1.0, abs(l0))
    assert rel_err(g1, g0) < 1e-6
    for n in b0:  # BatchNorm running statistics of the distilling layers (written by the slab-summing forward)
        assert rel_err(b1[n], b0[n]) < 1e-6, n


def test_engine_sinks_match_autograd():
    """Gradient sinks (kernels accumulate straight into the flat buffer, packed QKV) give the same
    gradients as plain autograd accumulation on the same model and batch."""
    from routeformer_amd.engine import TrainEngine, train_step_losses, trainable_parameters
    model, cfg, sd, c = build_product_model("c2_small", DEV)
    item = case_item(c)
    item_d = {"train": _to_dev(item["train"]), "target": _to_dev(item["target"])}
    model.train()
    torch.manual_seed(5)
    train_step_losses(model, item_d, 10)["loss"].backward()
    ref = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    eng = TrainEngine(model)
    assert any("_packed" in m.__dict__ for m in model.modules()), "no attention layer got packed QKV views"
    torch.manual_seed(5)
    eng._fwd_bwd(item_d, 10)
    for n, p in model.named_parameters():
        if n in ref:
            assert rel_err(p.grad, ref[n]) < 2e-5, n
    assert set(ref) == {n for n, p in model.named_parameters() if "video_backbone" not in n}


def test_graphed_engine_fresh_tensors_stale_buffers_and_discount_keys():
    """The graphed engine owns every buffer its graphs read (engine.GraphedTrainEngine._stage_clips):
    (a) a loader that hands over freshly allocated device tensors every step (no ids) gets the same step as the
        eager engine, with a bounded number of captured graphs and no memory growth;
    (b) a loader that REFILLS its device buffer in place between ``step(next_item=...)`` and the next ``step`` does
        not train on stale trunk tokens: batches are told apart by ``item["id"]``, not by ``data_ptr``;
    (c) an epoch that is a key of ``discount_factor`` re-captures the loss arithmetic (gamma is a by-value kernel
        argument of the fused trajectory head) -- graphed == eager across it."""
    from routeformer_amd import synthetic
    from routeformer_amd.engine import GraphedTrainEngine, TrainEngine

    def batch(c, seed):
        it = synthetic.synth_item(c["B"], c["T"], c["P"], seed, c["H"], c["W"], streams=c["streams"], gaze=c["gaze"])
        return {"train": _to_dev(it["train"]), "target": _to_dev(it["target"])}

    schedule = [(9, 21), (9, 22), (10, 23), (11, 24), (12, 25), (12, 26)]  # (epoch, data seed); 12 is a discount key
    runs = {}
    for mode in ("eager", "graph"):
        model, cfg, sd, c = build_product_model("c2_small", DEV)
        model.configs.discount_factor = {0: 0.9, 12: 0.5}
        eng = TrainEngine(model, lr=1e-3) if mode == "eager" else GraphedTrainEngine(model, lr=1e-3)
        if mode == "graph":
            eng.capture(batch(c, 20), epoch=9)
        torch.manual_seed(7)
        losses, mem = [], []
        for epoch, seed in schedule:
            losses.append(float(eng.step(batch(c, seed), epoch=epoch)["loss"]))  # fresh tensors, no id, no look-ahead
            torch.cuda.synchronize()
            mem.append(torch.cuda.memory_allocated())
        runs[mode] = losses
        if mode == "graph":
            assert len(eng._graphs) <= 2 and eng._trunk_g is not None
            assert mem[-1] <= mem[2] + (8 << 20), ("graphs / pools grew with the number of steps", mem)
    # lr 1e-3 makes this a chaotic comparison after a few updates: rounding-level differences in the fp32 atomics become
    # +-lr parameter steps and, from the fifth step on, different ProbSparse selections (observed 1.2e-2 on the loss).
    # The first four steps are held tightly; the last two only need to show that the epoch-12 discount was re-captured
    # (a stale gamma would be off by tens of percent: see the eager check below).
    for i, (a, b) in enumerate(zip(runs["eager"], runs["graph"])):
        assert abs(a - b) < (5e-4 if i < 4 else 5e-2) * max(1.0, abs(a)), (i, runs)
    # gamma 0.9 -> 0.5 at epoch 12 moves the loss by far more than the tolerance: the eager run must show it too
    assert abs(runs["eager"][4] - runs["eager"][3]) > 1e-3

    # (b) in-place refill between the look-ahead call and the next step (eager reference first: the key-sample
    # source is process-wide and switches to its static mode when a graph is captured)
    from routeformer_amd.models.blocks import SAMPLER
    SAMPLER.drop_static()
    a, b, fresh = batch(c, 31), batch(c, 32), batch(c, 33)
    ref_model, *_ = build_product_model("c2_small", DEV)
    ref = TrainEngine(ref_model, lr=1e-3)
    torch.manual_seed(3)
    ref.step(a, epoch=10)
    want = float(ref.step(fresh, epoch=10)["loss"])
    model, cfg, sd, c = build_product_model("c2_small", DEV)
    eng = GraphedTrainEngine(model, lr=1e-3).capture(batch(c, 30), epoch=10)
    a["id"], b["id"] = "a", "b"
    torch.manual_seed(3)
    eng.step(a, epoch=10, next_item=b)
    for part in ("train", "target"):          # the loader overwrites b's buffers with another batch ...
        for n, v in fresh[part].items():
            b[part][n].copy_(v)
    b["id"] = "fresh"                          # ... and says so
    got = float(eng.step(b, epoch=10)["loss"])
    assert abs(got - want) < 5e-4 * max(1.0, abs(want)), (got, want)
    # and with the stale id the engine WOULD have used the tokens computed ahead for the old contents
    SAMPLER.drop_static()


@pytest.mark.parametrize("case_name,B", [("C2", 8), ("C4", 16), ("C5", 4)])
def test_full_size_train_step(case_name, B):
    """One full train step of BASELINE.json configs[1], [3], [4] at their stated per-GPU batch (8 / 16 / 4), full
    resolution, paper hyper-parameters, bf16 matrix-core mode (what bench.py times): loss / ADE / FDE finite, and
    the HIP-graph replayed step equals the eager step (loss, gradient buffer, updated parameters)."""
    from routeformer_amd import kernels as K, presets, synthetic
    from routeformer_amd.engine import GraphedTrainEngine, TrainEngine
    from routeformer_amd.models import Routeformer, RouteformerConfig
    from routeformer_amd.models.gps_backbone import GPSBackboneConfig, Informer
    from routeformer_amd.models.video_backbone import HRNet16Backbone, VideoBackboneConfig
    from routeformer_amd.models.blocks import SAMPLER
    K.set_precision("bf16")
    c = presets.case(case_name)
    assert c["B"] == B
    _, cfg = presets.build_configs(c, GPSBackboneConfig, RouteformerConfig, VideoBackboneConfig)
    item = synthetic.synth_item(B, c["T"], c["P"], 21, c["H"], c["W"], streams=c["streams"], gaze=c["gaze"])
    item = {"train": _to_dev(item["train"]), "target": _to_dev(item["target"])}
    out = {}
    for mode in ("eager", "graph"):
        model = Routeformer(cfg, gps_backbone=Informer, video_backbone=HRNet16Backbone)
        model.load_state_dict(synthetic.synth_state_dict(model.state_dict(), 7))
        model = model.to(DEV)
        eng = TrainEngine(model) if mode == "eager" else GraphedTrainEngine(model).capture(item, epoch=10)
        torch.manual_seed(5)
        res = eng.step(item, epoch=10)
        torch.cuda.synchronize()
        for k in ("loss", "traj_loss", "dense_loss", "ade", "fde"):
            assert torch.isfinite(res[k]).all(), (mode, k)
        assert res["future_gps"].shape == (B, c["P"], 2) and torch.isfinite(res["future_gps"]).all()
        out[mode] = (float(res["loss"]), eng.reducer.flat_grad.clone(), eng.reducer.flat_param.clone(),
                     res["future_gps"].clone())
        SAMPLER.drop_static()
        del eng, model
        torch.cuda.empty_cache()
    (le, ge, pe, fe), (lg, gg, pg, fg) = out["eager"], out["graph"]
    assert abs(le - lg) < 1e-4 * max(1.0, abs(le)), (le, lg)
    assert rel_err(fg, fe) < 1e-4
    assert rel_err(gg, ge) < 2e-3 and rel_err(pg, pe) < 1e-4   # fp32-atomic summation order only


@pytest.mark.parametrize("case_name,B", [("C2", 8), ("C4", 16), ("C5", 4)])
def test_full_size_batch_consistency(case_name, B):
    """BASELINE.json configs[1], [3], [4] at full resolution / horizon / paper hyper-parameters and their stated
    per-GPU batch: size-independent properties instead of a CPU oracle run --
    (i) every sample's trajectory is independent of what else is in the batch (eval mode: per-sample ops,
        BatchNorm on running stats, one shared key-sample table per call exactly like the reference), so
        forwarding sample i alone with the same seed reproduces row i of the batched output;
    (ii) same seed -> bitwise identical outputs; different seed -> different key samples, still finite."""
    from routeformer_amd import presets, synthetic
    from routeformer_amd.models import Routeformer, RouteformerConfig
    from routeformer_amd.models.gps_backbone import GPSBackboneConfig, Informer
    from routeformer_amd.models.video_backbone import HRNet16Backbone, VideoBackboneConfig
    c = presets.case(case_name)
    c["B"] = B
    _, cfg = presets.build_configs(c, GPSBackboneConfig, RouteformerConfig, VideoBackboneConfig)
    model = Routeformer(cfg, gps_backbone=Informer, video_backbone=HRNet16Backbone)
    model.load_state_dict(synthetic.synth_state_dict(model.state_dict(), 7))
    model = model.to(DEV).eval()
    batch = _to_dev(synthetic.synth_batch(B, c["T"], 21, c["H"], c["W"], streams=c["streams"], gaze=c["gaze"]))

    def run(bt, seed):
        torch.manual_seed(seed)
        with torch.no_grad():
            out = model(bt)
        return out[0] if isinstance(out, tuple) else out

    full = run(batch, 5)
    assert full.shape == (B, c["P"], 2) and torch.isfinite(full).all()
    assert torch.equal(full, run(batch, 5)), "same seed must reproduce bitwise"
    other = run(batch, 6)
    assert torch.isfinite(other).all() and not torch.equal(full, other)
    for i in (0, B - 1):
        single = run({k: v[i:i + 1] for k, v in batch.items()}, 5)
        assert rel_err(single[0], full[i]) < 1e-4, (case_name, i)


def test_uint8_clips_equal_fp16_clips():
    """SURVEY 8(f) #3 (video ingest): a batch whose clips are raw uint8 frames gives bit-identical predictions to
    the same batch converted the dataset's way (`astype(np.float16) / 255.0`, io/dataset.py:1506-1523)."""
    from routeformer_amd.models.blocks import SAMPLER
    model, cfg, sd, c = build_product_model("c2_small", DEV)
    model.eval()
    item = case_item(c)["train"]
    g = torch.Generator().manual_seed(3)
    raw = {k: torch.randint(0, 256, v.shape, generator=g, dtype=torch.uint8) for k, v in item.items() if k.endswith("_video")}
    outs = []
    for conv in (False, True):
        batch = {k: v.to(DEV) for k, v in item.items() if not k.endswith("_video")}
        for k, v in raw.items():
            batch[k] = (torch.from_numpy(v.numpy().astype(np.float16) / 255.0) if conv else v).to(DEV)
        torch.manual_seed(5)
        with torch.no_grad():
            outs.append(model(batch))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


# ------------------------------------------------------------------------------------------------
# nn.Dropout on the trainable path (the paper configuration must be able to train)
# ------------------------------------------------------------------------------------------------
def _grads_vs_summary(G, key, named, tol):
    names = [str(s) for s in G[key + "grad_names"]]
    stats = G[key + "grad_stats"]
    floor = 1e-3 * float(stats[:, 0].max())
    bad = []
    for n, (nrm, _) in zip(names, stats):
        g = named[n].grad
        got = 0.0 if g is None else float(g.double().norm())
        if abs(got - nrm) > tol * max(floor, nrm):
            bad.append((n, got, nrm))
    assert not bad, bad[:8]
    for f in G.files:
        if f.startswith(key + "grad::"):
            assert rel_err(named[f[len(key + "grad::"):]].grad, G[f]) < tol, f


@pytest.mark.parametrize("prec,tol", [("f32", 2e-4), ("bf16", 5e-2)])
def test_dropout_blocks_vs_reference(prec, tol):
    """The product's encoder / decoder / GPS blocks in TRAIN mode with dropout 0.1 against the reference's outputs and
    gradients (tests/golden/dropout.npz), with the reference's recorded keep-masks injected (``K.RNG.forced``) and its
    key samples replayed: every dropout site -- attention output, FFN hidden (dropped by the reference in its
    (B, d_ff, L) layout) and output, the FullAttention probabilities (in-kernel), the data embedding."""
    from conftest import masks
    from routeformer_amd import kernels as K, presets
    from routeformer_amd.models.blocks import SAMPLER, PerceiveDecoder, PerceiveEncoder
    from routeformer_amd.models.gps_backbone import GPSBackboneConfig, Informer, Transformer
    G = golden("dropout")
    K.set_precision(prec)
    P = 0.1

    def run(module, inputs, key, loss, oracle):
        """``oracle(sd, idx, drop)``: the CPU oracle on the same weights / samples / masks -- only to learn its top-u
        selections, which are imposed on the kernels (the selection is discontinuous: a bf16 rounding can flip it)."""
        module.train()
        sd = {"m." + k: v.detach().cpu() for k, v in module.state_dict().items()}
        src = O.IndexSource(draws(G, key))
        with torch.no_grad():
            oracle(sd, src, O.DropoutSource(masks(G, key)))
        SAMPLER.replay = draws(G, key)
        K.RNG.forced = [m.clone() for m in masks(G, key)]
        K.TOPS.forced = [t_.clone() for t_ in src.tops]
        try:
            y = module(*inputs)
            assert not K.RNG.forced and not SAMPLER.replay and not K.TOPS.forced, key
        finally:
            K.RNG.forced, SAMPLER.replay, K.TOPS.forced = None, None, None
        loss(y).backward()
        return y

    enc = _load(PerceiveEncoder(in_channels=240, out_channels=64, out_len=1, n_heads=8, layers=2, d_ff=256, dropout=P))
    x = t(G["enc.x"]).to(DEV).requires_grad_()
    y = run(enc, (x,), "enc.", lambda y: y.square().sum(),
            lambda sd, idx, drop: O.perceive_encoder(sd, "m", t(G["enc.x"]), 8, 1, idx, dropout=P, drop=drop))
    # bf16 mode: gradients through nearly-uniform softmax rows cancel 3-4 digits (see test_model_train_step_golden), so
    # only the outputs and the input gradient (Frobenius) are held to the bf16 tolerance there
    from conftest import fro_err
    exact = prec == "f32"
    assert rel_err(y, G["enc.y"]) < tol and (rel_err(x.grad, G["enc.dx"]) < 5 * tol if exact else fro_err(x.grad, G["enc.dx"]) < 0.15)
    if exact:
        _grads_vs_summary(G, "enc.", dict(enc.named_parameters()), 5 * tol)

    dec = _load(PerceiveDecoder(query_channels=64, value_channels=64, out_channels=64, out_len=40, dropout=P, d_ff=256,
                                n_heads=8, layers=2, mix=False))
    mem, qry = t(G["dec.mem"]).to(DEV).requires_grad_(), t(G["dec.qry"]).to(DEV).requires_grad_()
    yd = run(dec, (mem, qry), "dec.", lambda y: y.square().sum(),
             lambda sd, idx, drop: O.perceive_decoder(sd, "m", t(G["dec.mem"]), t(G["dec.qry"]), 8, 40, idx, dropout=P,
                                                      drop=drop))
    assert rel_err(yd, G["dec.y"]) < tol
    if exact:
        assert rel_err(mem.grad, G["dec.dmem"]) < 5 * tol and rel_err(qry.grad, G["dec.dqry"]) < 5 * tol
        _grads_vs_summary(G, "dec.", dict(dec.named_parameters()), 5 * tol)
    else:
        assert fro_err(mem.grad, G["dec.dmem"]) < 0.15 and fro_err(qry.grad, G["dec.dqry"]) < 0.15

    for tag, cls in (("inf", Informer), ("tf", Transformer)):
        gcfg = GPSBackboneConfig(seq_len=20, label_len=20, pred_len=10, **dict(presets.GPS_TINY, dropout=P))
        gcfg.output_attention, gcfg.smart_decoder, gcfg._enc_in, gcfg._c_out = False, True, 69, 66
        net = _load(cls(gcfg))
        xg = t(G[tag + ".x"]).to(DEV).requires_grad_()
        if tag == "inf":
            orc = lambda sd, idx, drop: O.informer(sd, "m", t(G["inf.x"]), pred_len=10, n_heads=gcfg.n_heads,  # noqa: E731
                                                   factor=gcfg.factor, activation=gcfg.activation, smart_decoder=True,
                                                   training=True, idx=idx, dropout=P, drop=drop)
        else:
            orc = lambda sd, idx, drop: O.transformer_gps(sd, "m", t(G["tf.x"]), pred_len=10, n_heads=gcfg.n_heads,  # noqa: E731
                                                          activation=gcfg.activation, dropout=P, drop=drop)
        yg = run(net, (xg,), tag + ".", lambda y: y.square().mean(), orc)
        assert rel_err(yg, G[tag + ".y"]) < tol, tag
        if exact:
            assert rel_err(xg.grad, G[tag + ".dx"]) < 5 * tol, tag
            _grads_vs_summary(G, tag + ".", dict(net.named_parameters()), 5 * tol)
        # (bf16 mode: the d_model-64 GPS blocks' input gradient is a sum of cancelling terms; only the output is held)


def _dropout_case():
    from routeformer_amd import presets
    from routeformer_amd.models import RouteformerConfig
    from routeformer_amd.models.gps_backbone import GPSBackboneConfig
    from routeformer_amd.models.video_backbone import VideoBackboneConfig
    c = presets.case("c2_small")
    c["rf"] = dict(c["rf"], feature_dropout=0.1, view_dropout=0.6, gaze_dropout=0.2)
    c["gps"] = dict(c["gps"], dropout=0.1)
    _, cfg = presets.build_configs(c, GPSBackboneConfig, RouteformerConfig, VideoBackboneConfig)
    return c, cfg


def _dropout_model(cfg):
    from routeformer_amd import synthetic
    from routeformer_amd.models import Routeformer
    from routeformer_amd.models.gps_backbone import Informer
    from routeformer_amd.models.video_backbone import HRNet16Backbone
    model = Routeformer(cfg, gps_backbone=Informer, video_backbone=HRNet16Backbone)
    sd = synthetic.synth_state_dict(model.state_dict(), 7)
    model.load_state_dict(sd)
    return model.to(DEV), sd


@pytest.mark.parametrize("kind", ["none", "view", "gaze"])
def test_dropout_train_step_vs_reference(kind):
    """A whole train step with view 0.6 / gaze 0.2 / feature 0.1 / GPS 0.1 dropout against the reference run: the same
    host seed gives the same view / gaze decisions and key samples (the masks never touch the host generator, as on a
    GPU run of the reference), the recorded masks are injected, the oracle's top-u selections imposed."""
    from conftest import masks
    from routeformer_amd import kernels as K
    from routeformer_amd.engine import train_step_losses
    from routeformer_amd.models.blocks import SAMPLER
    G = golden("dropout")
    c, cfg = _dropout_case()
    model, sd = _dropout_model(cfg)
    item = case_item(c)
    item_d = {"train": _to_dev(item["train"]), "target": _to_dev(item["target"])}
    seed = int(G["model.seeds"][["none", "view", "gaze"].index(kind)])
    key = f"model.{kind}."
    torch.manual_seed(seed)
    orc = O.OracleRouteformer(cfg, sd, training=True, drop=O.DropoutSource(masks(G, key)))
    with torch.no_grad():
        orc.train_step(item, 10)
    model.train()
    K.TOPS.forced = [t_.clone() for t_ in orc.idx.tops]
    K.RNG.forced = [m.clone() for m in masks(G, key)]
    SAMPLER.log = []
    torch.manual_seed(seed)
    try:
        res = train_step_losses(model, item_d, 10)
        assert not K.TOPS.forced and not K.RNG.forced
    finally:
        K.TOPS.forced, K.RNG.forced = None, None
    assert len(SAMPLER.log) == int(G[key + "n_draws"])
    assert all(torch.equal(a, b) for a, b in zip(SAMPLER.log, draws(G, key)))
    assert torch.equal(torch.rand(1), t(G[key + "rng_after"])), "host generator diverged from the reference"
    assert rel_err(res["future_gps"], G[key + "future_gps"]) < TOL_F32
    assert rel_err(res["target_vis"], G[key + "target_vis"]) < TOL_F32
    for k in ("loss", "traj_loss", "dense_loss", "ade", "fde"):
        ref = float(G[key + k])
        assert abs(float(res[k]) - ref) < 1e-3 * max(1.0, abs(ref)), (k, float(res[k]), ref)
    res["loss"].backward()
    _grads_vs_summary(G, key, dict(model.named_parameters()), 5e-3)


def test_graphed_engine_with_dropouts_matches_eager():
    """The paper run's dropouts (full_comparison.py:272-275) under HIP-graph replay: the host decisions (view / gaze
    dropout) are drawn before every replay in the reference's order and select one of the captured step variants;
    the device-side masks advance with a device-resident step counter.  Same host seed + same mask seed -> the
    graphed engine takes the same decisions and computes the same steps as the eager engine, and the optimizer skips
    the gaze branch's slots on the steps that dropped it (grad None -> AdamW skip in the reference)."""
    from routeformer_amd import kernels as K, synthetic
    from routeformer_amd.engine import GraphedTrainEngine, TrainEngine
    from routeformer_amd.models.blocks import SAMPLER
    c, cfg = _dropout_case()
    items = []
    for seed in (11, 12):
        it = synthetic.synth_item(c["B"], c["T"], c["P"], seed, c["H"], c["W"], streams=c["streams"], gaze=c["gaze"])
        items.append({"train": _to_dev(it["train"]), "target": _to_dev(it["target"]), "id": seed})
    runs = {}
    from routeformer_amd import engine as E
    for mode in ("eager", "graph", "graph_deferred"):
        model, sd = _dropout_model(cfg)
        was = E.EARLY_SUMSQ  # the deferred run also takes the clip norm's big term mid-backward (measurement switch, off by default)
        E.EARLY_SUMSQ = mode == "graph_deferred"
        try:
            eng = (TrainEngine(model, lr=1e-4) if mode == "eager" else
                   GraphedTrainEngine(model, lr=1e-4, defer_update=mode == "graph_deferred").capture(items[0], epoch=10))
        finally:
            E.EARLY_SUMSQ = was
        if mode != "eager":
            assert SAMPLER.n_variants == 6  # view: keep | drop left | drop right, x gaze: keep | drop
        if mode == "graph_deferred":  # the update is launched per segment: backbone | gaze slots | the rest
            assert len(eng._segments) >= 3 and sum(side for _, _, side in eng._segments) == 1
        K.RNG.manual_seed(11)   # restarts the device step counter: both engines draw the same masks from here on
        torch.manual_seed(5)
        gaze_w = dict(model.named_parameters())["gaze_encoder.projection.weight"]
        losses, decisions, gaze_moved = [], [], []
        for i in range(14):
            before = gaze_w.detach().clone()
            res = eng.step(items[i % 2], epoch=10, next_item=items[(i + 1) % 2] if mode != "eager" else None)
            losses.append(float(res["loss"].detach()))
            decisions.append(tuple(model.__dict__.get("_unused_prefixes", ())) if mode == "eager"
                             else tuple(eng._variant_unused[SAMPLER._variant]))
            gaze_moved.append(bool((gaze_w.detach() != before).any()))
            if mode == "graph_deferred":  # the clip norm's early partial sums saw every gradient of the backbone's range
                from routeformer_amd import _hip
                assert eng._early_sumsq
                for a, b, o, early in eng.opt._ss_plan:
                    if early:
                        k = int(_hip.lib().rf_sumsq_parts(b - a))
                        got = float(eng.opt.sumsq[o:o + k].double().sum())
                        want = float(eng.reducer.flat_grad[a:b].double().pow(2).sum())
                        assert want > 0 and abs(got - want) < 1e-5 * want, (i, got, want)
        torch.cuda.synchronize()
        if mode == "graph_deferred":
            before = gaze_w.detach().clone()
            eng.flush()  # the last step's update is still pending
            torch.cuda.synchronize()
            gaze_moved.append(bool((gaze_w.detach() != before).any()))
            assert eng.opt.t == 14
        runs[mode] = (losses, decisions, gaze_moved, eng.reducer.flat_param.clone(), torch.get_rng_state())
        if mode != "eager":
            assert len({k[1] for k in eng._graphs}) >= 3, "expected several decision variants in 14 steps"
        SAMPLER.drop_static()
    (le, de, me, pe, re_), (lg, dg, mg, pg, rg) = runs["eager"], runs["graph"]
    assert de == dg and torch.equal(re_, rg), "host decisions / generator diverged between eager and graph mode"
    assert any(de) and not all(de), "the schedule should contain steps with and without the gaze branch"
    # dropped gaze branch: its parameters do not move that step (no weight decay, no moment decay) -- in both engines
    assert me == [not d for d in de] and mg == me
    assert all(abs(a - b) < 2e-3 * max(1.0, abs(a)) for a, b in zip(le, lg)), (le, lg)
    assert rel_err(pg, pe) < 14 * 3e-4
    # deferred update (the step's clip + AdamW replayed at the head of the NEXT step, per segment with device-side
    # "pending" flags and per-segment update counts): same decisions, the gaze slots move one call later -- or not at all
    # for a step that dropped the branch --, same parameters after the final flush
    ld, dd, md, pd, rd = runs["graph_deferred"]
    assert dd == de and torch.equal(rd, re_)
    assert md == [False] + [not d for d in de], (md, de)
    assert all(abs(a - b) < 2e-3 * max(1.0, abs(a)) for a, b in zip(le, ld)), (le, ld)
    assert rel_err(pd, pe) < 14 * 3e-4


def test_fused_stack_matches_layerwise_train_step():
    """The fused per-sequence encoder stack (one launch for all layers of the frame / gaze / fusion encoders,
    csrc/seqlayer.hip) against the layer-by-layer kernels inside one whole train step, bf16 matrix-core mode: same
    host draws, the layer-by-layer run's top-u selections imposed on the fused run (q / k are rounded to bf16 inside
    the fused kernel, which can flip a near-tie) -- loss, trajectories and the whole gradient buffer agree to bf16
    rounding; and the fused path was really taken."""
    from conftest import fro_err
    from routeformer_amd import kernels as K
    from routeformer_amd.engine import TrainEngine
    K.set_precision("bf16")
    out, calls = {}, []
    real = K._seqstack_launch
    try:
        for fused in (False, True):
            K.SEQSTACK = fused
            model, cfg, sd, c = build_product_model("c2_small", DEV)
            item = case_item(c)
            item_d = {"train": _to_dev(item["train"]), "target": _to_dev(item["target"])}
            eng = TrainEngine(model)
            model.train()
            if fused:
                K.TOPS.forced = [t_.clone() for t_ in out[False][3]]
                K._seqstack_launch = lambda *a, **k: (calls.append(a[6:8]), real(*a, **k))[1]
            else:
                K.TOPS.record = []
            torch.manual_seed(5)
            res = eng._fwd_bwd(item_d, 10)
            torch.cuda.synchronize()
            if fused:
                assert not K.TOPS.forced
            out[fused] = (float(res["loss"].detach()), res["future_gps"].detach().clone(), eng.reducer.flat_grad.clone(),
                          K.TOPS.record)
            K.TOPS.record, K.TOPS.forced = None, None
    finally:
        K.SEQSTACK, K._seqstack_launch = True, real
        K.TOPS.record, K.TOPS.forced = None, None
    assert len(calls) >= 4, f"the fused stack was not used ({calls})"  # frame, gaze, fusion encoders x (input, target)
    (l0, f0, g0, _), (l1, f1, g1, _) = out[False], out[True]
    assert abs(l0 - l1) < 2e-2 * max(1.0, abs(l0)), (l0, l1)
    assert rel_err(f1, f0) < TOL_BF16
    # whole-model gradient through two different bf16 roundings of the stacks' forward (measured 0.09-0.11 depending on
    # the GEMM summation order; the backward kernels themselves agree to 2.5e-4 on identical forward saves -- see
    # test_fused_stack_backward_vs_layerwise for the well-conditioned comparison)
    assert fro_err(g1, g0) < 0.15, fro_err(g1, g0)


def test_token_cache_in_model_and_engine():
    """The backbone-feature cache (the reference's torchcache steady state) changes nothing but the work: eval outputs
    with the cache cold, warm and absent are bit-identical; the graphed engine with the cache skips the trunk for batch
    ids it has seen and still computes the eager engine's steps."""
    from routeformer_amd import synthetic
    from routeformer_amd.engine import GraphedTrainEngine, TrainEngine
    from routeformer_amd.models.blocks import SAMPLER
    from routeformer_amd.models.video_backbone import TokenCache
    model, cfg, sd, c = build_product_model("c2_small", DEV)
    model.eval()
    item = case_item(c)
    batch = _to_dev(item["train"])

    def run():
        torch.manual_seed(5)
        with torch.no_grad():
            return model(batch)

    ref = run()
    model.video_backbone.token_cache = TokenCache(64, DEV)
    cold, warm = run(), run()
    assert model.video_backbone.token_cache.hits > 0
    for a in (cold, warm):
        assert torch.equal(a[0], ref[0]) and torch.equal(a[1], ref[1])
    model.video_backbone.token_cache = None

    items = []
    for seed in (11, 12):
        it = synthetic.synth_item(c["B"], c["T"], c["P"], seed, c["H"], c["W"], streams=c["streams"], gaze=c["gaze"])
        items.append({"train": _to_dev(it["train"]), "target": _to_dev(it["target"]), "id": seed})
    losses = {}
    for mode in ("eager", "graph_cached"):
        m2, *_ = build_product_model("c2_small", DEV)
        if mode == "eager":
            eng = TrainEngine(m2, lr=1e-3)
        else:
            m2.video_backbone.token_cache = TokenCache(256, DEV)
            eng = GraphedTrainEngine(m2, lr=1e-3).capture(items[0], epoch=10)
        torch.manual_seed(9)
        losses[mode] = [float(eng.step(items[i % 2], epoch=10, next_item=items[(i + 1) % 2])["loss"].detach()) for i in range(5)]
        if mode != "eager":
            assert set(eng._cached_ids) == {11, 12} and eng._trunk_g is not None
            replays = []
            real = eng._trunk_g.replay
            eng._trunk_g.replay = lambda: (replays.append(1), real())[1]
            n_graphs = len(eng._graphs)
            eng.step(items[1], epoch=10, next_item=items[0])
            assert not replays and len(eng._graphs) == n_graphs, "a cached batch must not run the trunk"
            eng.verify_ids()  # honest ids: the content check passes
            # an id REUSED for other frames (ids restarting each epoch / a shuffling loader keyed by batch
            # index) is caught by the device-side content check, reported at the latest by the next step
            other = synthetic.synth_item(c["B"], c["T"], c["P"], 13, c["H"], c["W"], streams=c["streams"], gaze=c["gaze"])
            liar = {"train": _to_dev(other["train"]), "target": _to_dev(other["target"]), "id": 11}
            eng.step(liar, epoch=10)
            torch.cuda.synchronize()
            with pytest.raises(RuntimeError, match="differ from the ones cached"):
                eng.step(items[1], epoch=10)
            assert not eng._cached_ids  # forgotten: the next steps re-hash and re-learn
            eng.step(items[1], epoch=10)
            eng.verify_ids()
        SAMPLER.drop_static()
    assert all(abs(a - b) < 5e-4 * max(1.0, abs(a)) for a, b in zip(losses["eager"], losses["graph_cached"])), losses


@pytest.mark.parametrize("B,L", [(6, 65), (3, 160), (2, 97), (2, 320), (1, 97)])
def test_fused_stack_dropout_vs_oracle_and_layerwise(B, L):
    """(L = 65: the one-workgroup-per-sequence stack; L = 160 / 97 / 320: the row-tiled stack, csrc/enclayer.hip --
    cross_modal_transformer.py:288-301 x layers.)
    Dropout INSIDE the fused encoder stack (train mode, p = 0.2), forward AND backward against the CPU oracle's autograd:
    a first free run materialises the Philox masks the kernels draw (``K.RNG.record``; they depend on (seed, step, site,
    element) only) and the host draws; the oracle runs on those masks and draws, with autograd; the fused stack then runs
    again with the oracle's top-u selections imposed and its output, input gradient and EVERY parameter gradient are held
    to the oracle's (bf16 operand rounding is the only difference); finally the layer-by-layer kernels run on the same
    masks and selections (fused-vs-layerwise, as before)."""
    from conftest import fro_err
    from routeformer_amd import kernels as K, synthetic
    from routeformer_amd.engine import GradReducer
    from routeformer_amd.models.blocks import SAMPLER, PerceiveEncoder
    K.set_precision("bf16")
    P = 0.2
    g = torch.Generator().manual_seed(3)
    x_cpu = torch.randn(B, L, 240, generator=g)
    w_cpu = torch.randn(B, 1, 64, generator=g)
    out, orc = {}, {}
    try:
        for mode in ("fused_free", "fused", "layerwise"):
            K.SEQSTACK = mode != "layerwise"
            enc = _load(PerceiveEncoder(in_channels=240, out_channels=64, out_len=1, n_heads=8, layers=3, d_ff=256, dropout=P))
            enc.train()
            # gradient sinks + packed QKV views (what TrainEngine sets up): the fused path needs them for its backward
            layers = [m for m in enc.modules() if hasattr(m, "packing_groups")]
            red = GradReducer(list(enc.parameters()), groups=[g_ for m in layers for g_ in m.packing_groups()])
            for m in layers:
                gw, gb = m.packing_groups()
                vw, vb = red.packed_view(gw), red.packed_view(gb)
                m._packed = {"w": vw[0], "gw": vw[1], "b": vb[0], "gb": vb[1]}
            eng = type("E", (), {"reducer": red})()
            K.SINK.active = True
            K.RNG.manual_seed(77)
            K.RNG.begin_step(torch.device(DEV))
            eng.reducer.zero()
            x = x_cpu.to(DEV).requires_grad_()
            torch.manual_seed(11)
            if mode == "fused_free":
                K.RNG.record, K.TOPS.record = [], []
                SAMPLER.log = []
            else:
                if mode == "layerwise":
                    K.RNG.forced = [m.clone() for m in out["fused_free"]["masks"]]
                K.TOPS.forced = [t_.clone() for t_ in orc["tops"]]
            y = enc(x)
            (y * w_cpu.to(DEV)).sum().backward()
            K.flush_weight_grads()
            torch.cuda.synchronize()
            assert mode == "fused_free" or not K.TOPS.forced
            out[mode] = dict(y=y.detach().cpu(), dx=x.grad.detach().cpu(), grad=eng.reducer.flat_grad.clone().cpu(),
                             pgrad={n: p_.grad.detach().cpu().clone() for n, p_ in enc.named_parameters()},
                             masks=K.RNG.record, tops=K.TOPS.record, draws=SAMPLER.log,
                             sd={k: v.detach().cpu().clone() for k, v in enc.state_dict().items()})
            K.RNG.record, K.RNG.forced, K.TOPS.record, K.TOPS.forced, SAMPLER.log = None, None, None, None, None
            K.SINK.active = False
            if mode == "fused_free":
                f = out[mode]
                assert len(f["masks"]) == 9
                assert abs(float(torch.stack([m.float().mean() for m in f["masks"]]).mean()) - (1 - P)) < 0.01
                # the CPU oracle on the same masks (it drops the hidden activation in its (B, d_ff, L) layout) and draws
                sd = {"m." + k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(".pe")) for k, v in f["sd"].items()}
                masks = [m.cpu() if (i % 3) != 1 else m.cpu().transpose(1, 2) for i, m in enumerate(f["masks"])]
                src = O.IndexSource([d.clone() for d in f["draws"]])
                x_o = x_cpu.clone().requires_grad_()
                y_o = O.perceive_encoder(sd, "m", x_o, 8, 1, src, dropout=P, drop=O.DropoutSource(masks))
                (y_o * w_cpu).sum().backward()
                orc = dict(y=y_o.detach(), dx=x_o.grad.detach(), tops=src.tops,
                           pgrad={k[2:]: v.grad.detach() for k, v in sd.items() if v.requires_grad and v.grad is not None})
                flips = sum(int((a.cpu().long() != b).any(-1).sum()) for a, b in zip(f["tops"], src.tops))
                print(f"fused dropout (B {B}, L {L}): free-running selections differing from the oracle's: {flips} of "
                      f"{sum(t_.shape[0] * t_.shape[1] for t_ in src.tops)} (b, h) problems; free rel err {rel_err(f['y'], y_o):.2e}")
    finally:
        K.SEQSTACK = True
        K.SINK.active = False
        K.RNG.record, K.RNG.forced, K.TOPS.record, K.TOPS.forced, SAMPLER.log = None, None, None, None, None
    f, u = out["fused"], out["layerwise"]
    # (1) fused forward + backward vs the oracle's autograd, same masks, draws and selections
    e_y, e_dx = rel_err(f["y"], orc["y"]), fro_err(f["dx"], orc["dx"])
    rows = sorted(((fro_err(f["pgrad"][n], go), n) for n, go in orc["pgrad"].items()
                   if float(go.norm()) > 1e-3 * max(float(g_.norm()) for g_ in orc["pgrad"].values())), reverse=True)
    flat_o = torch.cat([orc["pgrad"][n].double().reshape(-1) for n in sorted(orc["pgrad"])])
    flat_f = torch.cat([f["pgrad"][n].double().reshape(-1) for n in sorted(orc["pgrad"])])
    whole = float(flat_f @ flat_o / (flat_f.norm() * flat_o.norm()))
    print(f"fused dropout (B {B}, L {L}) vs oracle autograd: y rel err {e_y:.2e}, dx fro err {e_dx:.2e}, whole-gradient cosine "
          f"{whole:.5f}, worst parameter gradients (fro): " + ", ".join(f"{n} {e:.2e}" for e, n in rows[:5]))
    # bounds: bf16 operand rounding through 3 layers (observed: y <= 5e-3, dx <= 2e-2, parameters <= 4e-2 except the
    # cancellation-dominated query / key projections, which are reported above and bounded through the whole gradient)
    assert e_y < TOL_BF16 and e_dx < 5e-2, (e_y, e_dx)
    assert whole > 0.995, whole
    soft = lambda n: ".query_projection." in n or ".key_projection." in n  # noqa: E731
    hard = [(e, n) for e, n in rows if not soft(n)]
    assert hard[0][0] < 8e-2, hard[:5]
    # (2) vs the layer-by-layer path on the same masks and selections
    assert rel_err(f["y"], u["y"]) < 3e-2
    assert fro_err(f["dx"], u["dx"]) < 0.1 and fro_err(f["grad"], u["grad"]) < 0.1, (fro_err(f["dx"], u["dx"]), fro_err(f["grad"], u["grad"]))


@pytest.mark.parametrize("B,L,S,act,P", [(8, 40, 40, "gelu", 0.0), (3, 21, 40, "gelu", 0.0), (2, 70, 33, "relu", 0.0),
                                         (8, 40, 40, "gelu", 0.2), (3, 33, 21, "relu", 0.1)])
def test_rowchain_decoder_vs_layerwise(B, L, S, act, P):
    """The d_model = 64 PerceiveDecoder as attention launches + row-local chains (csrc/rowchain.hip: out-projection + residual
    + LayerNorm [+ FFN + LayerNorm] + the next projection in ONE launch, forward and backward) against the layer-by-layer
    kernels, bf16 matrix-core mode, same host draws and the layer-by-layer run's selections imposed: output, input gradients
    (queries and memory) and the whole parameter-gradient buffer; and the chain path was really taken.  P > 0: train-mode
    dropout INSIDE the chains (and on the cross-attention probabilities) -- the Philox masks the chain run drew are materialised
    (``K.RNG.record``) and injected into the layer-by-layer run, so both apply the same masks at the same sites."""
    from conftest import fro_err
    from routeformer_amd import kernels as K
    from routeformer_amd.engine import GradReducer
    from routeformer_amd.models.blocks import SAMPLER, PerceiveDecoder
    K.set_precision("bf16")
    g = torch.Generator().manual_seed(5)
    q_cpu, m_cpu = torch.randn(B, L, 128, generator=g), torch.randn(B, S, 64, generator=g)
    w_cpu = torch.randn(B, L, 64, generator=g)
    out, calls = {}, []
    real = K._RowChain.apply
    try:
        for chain in ((True, False) if P > 0 else (False, True)):  # (with dropout the chain run goes first: it draws the masks)
            K.ROWCHAIN = chain
            dec = _load(PerceiveDecoder(query_channels=128, value_channels=64, out_channels=64, out_len=L, n_heads=8, layers=2,
                                        dropout=P, activation=act))
            dec.train()
            layers = [m for m in dec.modules() if hasattr(m, "packing_groups")]
            red = GradReducer(list(dec.parameters()), groups=[g_ for m in layers for g_ in m.packing_groups()])
            for m in layers:
                gw, gb = m.packing_groups()
                vw, vb = red.packed_view(gw), red.packed_view(gb)
                m._packed = {"w": vw[0], "gw": vw[1], "b": vb[0], "gb": vb[1]}
            K.SINK.active = True
            K.RNG.manual_seed(77)
            K.RNG.begin_step(torch.device(DEV))
            red.zero()
            q, mem = q_cpu.to(DEV).requires_grad_(), m_cpu.to(DEV).requires_grad_()
            torch.manual_seed(11)
            first = (not chain) if P == 0 else chain
            if first:
                K.TOPS.record = []
                if P > 0:
                    K.RNG.record = []
            else:
                K.TOPS.forced = [t_.clone() for t_ in out[not chain]["tops"]]
                if P > 0:
                    K.RNG.forced = [m.clone() for m in out[not chain]["masks"]]
            if chain:
                K._RowChain.apply = lambda *a: (calls.append(1), real(*a))[1]
            y = dec(mem, q)
            (y * w_cpu.to(DEV)).sum().backward()
            K.flush_weight_grads()
            torch.cuda.synchronize()
            out[chain] = dict(y=y.detach().cpu(), dq=q.grad.detach().cpu(), dm=mem.grad.detach().cpu(),
                              grad=red.flat_grad.clone().cpu(), tops=K.TOPS.record, masks=K.RNG.record, rng=torch.get_rng_state())
            assert not K.TOPS.forced and not K.RNG.forced, "imposed selections / masks were not all consumed"
            K.TOPS.record, K.TOPS.forced, K.RNG.record, K.RNG.forced = None, None, None, None
            K.SINK.active = False
    finally:
        K.ROWCHAIN = True
        K._RowChain.apply = real
        K.SINK.active = False
        K.TOPS.record, K.TOPS.forced, K.RNG.record, K.RNG.forced = None, None, None, None
    c, u = out[True], out[False]
    assert len(calls) == 4, "two chains per decoder layer expected"
    if P > 0:  # 2 layers x (self out, cross probabilities, cross out, hidden, conv2 out)
        assert len(c["masks"]) == 10 and abs(float(torch.stack([m.float().mean() for m in c["masks"]]).mean()) - (1 - P)) < 0.02
    assert torch.equal(c["rng"], u["rng"]), "host draws differ between the two paths"
    errs = {"y": rel_err(c["y"], u["y"]), **{k: fro_err(c[k], u[k]) for k in ("dq", "dm", "grad")}}
    print("rowchain vs layer-by-layer:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert errs["y"] < 1e-2, errs          # (observed: <= 2.5e-3 output, <= 3.4e-3 gradients)
    for k in ("dq", "dm", "grad"):
        assert errs[k] < 1.5e-2, errs


@pytest.mark.parametrize("shape", [(6, 65, 8, "gelu", 0.0), (5, 40, 3, "gelu", 0.0), (3, 80, 2, "relu", 0.0),
                                   (4, 17, 2, "gelu", 0.0), (6, 65, 3, "gelu", 0.2), (3, 40, 2, "relu", 0.1)])
def test_fused_stack_backward_vs_layerwise(shape):
    """The fused backward of the encoder stack (csrc/seqlayer_bwd.hip, one launch for every layer's data path) against
    the layer-by-layer backward kernels on the SAME saved tensors (same fused forward, same selections): dx and every
    parameter gradient.  Both round their GEMM operands to bf16; the layer-by-layer attention backward keeps fp32
    probabilities, hence the (small) tolerance.  Shapes: the frame encoder (L = 65, 8 layers), the gaze encoder's
    length (L = 40, the 3-row-tile kernel), the longest supported sequence, a ragged short one; GELU and ReLU; with
    dropout the fused backward regenerates the forward's Philox masks in the kernel, the layer-by-layer path through
    rf_dropout -- same (seed, step, site, element) -> same masks."""
    from conftest import fro_err
    from routeformer_amd import kernels as K
    from routeformer_amd.engine import GradReducer
    from routeformer_amd.models.blocks import SAMPLER, PerceiveEncoder
    K.set_precision("bf16")
    B, L, layers, act, drop_p = shape
    g = torch.Generator().manual_seed(3)
    x_cpu = torch.randn(B, L, 240, generator=g)
    w_cpu = torch.randn(B, 1, 64, generator=g)
    out, used = {}, []
    real = K._seqstack_bwd_launch
    try:
        for fused_bwd in (True, False):
            K.SEQSTACK_BWD = True  # (the transposed fragments are packed either way; the switch below picks the path)
            enc = _load(PerceiveEncoder(in_channels=240, out_channels=64, out_len=1, n_heads=8, layers=layers, d_ff=256,
                                        dropout=drop_p, activation=act))
            enc.train()
            K.RNG.manual_seed(77)
            K.RNG.begin_step(torch.device(DEV))
            mods = [m for m in enc.modules() if hasattr(m, "packing_groups")]
            red = GradReducer(list(enc.parameters()), groups=[g_ for m in mods for g_ in m.packing_groups()])
            for m in mods:
                gw, gb = m.packing_groups()
                vw, vb = red.packed_view(gw), red.packed_view(gb)
                m._packed = {"w": vw[0], "gw": vw[1], "b": vb[0], "gb": vb[1]}
            K.SINK.active = True
            red.zero()
            x = x_cpu.to(DEV).requires_grad_()
            torch.manual_seed(11)
            y = enc(x)
            K.SEQSTACK_BWD = fused_bwd
            K._seqstack_bwd_launch = lambda *a, **k: (used.append(fused_bwd), real(*a, **k))[1]
            (y * w_cpu.to(DEV)).sum().backward()
            K.flush_weight_grads()
            torch.cuda.synchronize()
            names = [n for n, _ in enc.named_parameters()]
            grads = {n: p._rf_grad.detach().cpu().clone() for n, p in enc.named_parameters()}
            out[fused_bwd] = dict(y=y.detach().cpu(), dx=x.grad.detach().cpu(), grads=grads, names=names)
            K.SINK.active = False
    finally:
        K.SEQSTACK_BWD, K._seqstack_bwd_launch = True, real
        K.SINK.active = False
    assert used and all(used), "the fused backward did not run"
    f, u = out[True], out[False]
    assert torch.equal(f["y"], u["y"])
    assert torch.isfinite(f["dx"]).all()
    e_dx = fro_err(f["dx"], u["dx"])
    # (key-projection biases have a mathematically zero gradient -- softmax ignores a per-query constant -- so both
    #  paths hold rounding noise there: errors are measured against the RMS gradient magnitude of the whole model)
    rms = float(torch.cat([v.flatten() for v in u["grads"].values()]).square().mean().sqrt())
    def err(n):
        a, b_ = f["grads"][n].double(), u["grads"][n].double()
        return float((a - b_).norm() / max(float(b_.norm()), rms * b_.numel() ** 0.5 * 0.05))
    worst = max((err(n), n) for n in f["names"])
    print(f"fused backward {shape}: dx fro err {e_dx:.2e}; worst parameter gradient {worst[0]:.2e} ({worst[1]})")
    assert e_dx < 2e-2, e_dx
    assert worst[0] < 3e-2, worst


@pytest.mark.parametrize("B,L,layers,act", [(3, 160, 3, "gelu"), (2, 320, 2, "gelu"), (5, 120, 2, "relu"), (1, 97, 2, "gelu")])
def test_tiled_stack_vs_layerwise(B, L, layers, act):
    """The row-tiled encoder stack for sequences beyond the fused stack's L <= 80 (csrc/enclayer.hip: per layer one
    attention launch + one row-tile launch, forward and backward; the fusion encoder's L = 160 / 320, a ragged L, a tile
    spanning two sequences) in one PerceiveEncoder against the layer-by-layer kernels (RF_TILED_STACK off): same host draws,
    the layer-by-layer run's selections imposed (q / k are split-bf16 in the tiled path, plain bf16-operand products in
    the other: a near-tie may resolve differently) -- forward output, input gradient, every parameter gradient.  A third
    run takes the tiled forward with the layer-by-layer backward kernels on ITS saved tensors: the two backward
    implementations on identical inputs."""
    from conftest import fro_err
    from routeformer_amd import kernels as K
    from routeformer_amd.engine import GradReducer
    from routeformer_amd.models.blocks import PerceiveEncoder
    K.set_precision("bf16")
    g = torch.Generator().manual_seed(B * 100 + L)
    x_cpu = torch.randn(B, L, 64, generator=g)
    w_cpu = torch.randn(B, 40, 64, generator=g)
    out, fwd_calls, lw_calls = {}, [], []
    real_fwd, real_lw = K._TiledStack.forward, K._stack_backward_layerwise
    K._stack_backward_layerwise = lambda *a, **k: (lw_calls.append(1), real_lw(*a, **k))[1]
    try:
        for mode in ("layerwise", "tiled", "tiled_fwd_only"):
            K.TILED_STACK = mode != "layerwise"
            K.TILED_STACK_BWD = mode == "tiled"
            enc = _load(PerceiveEncoder(in_channels=64, out_channels=64, out_len=40, n_heads=8, layers=layers, d_ff=256,
                                        dropout=0.0, activation=act))
            enc.train()
            mods = [m for m in enc.modules() if hasattr(m, "packing_groups")]
            red = GradReducer(list(enc.parameters()), groups=[g_ for m in mods for g_ in m.packing_groups()])
            for m in mods:
                gw, gb = m.packing_groups()
                vw, vb = red.packed_view(gw), red.packed_view(gb)
                m._packed = {"w": vw[0], "gw": vw[1], "b": vb[0], "gb": vb[1]}
            K.SINK.active = True
            red.zero()
            if mode == "layerwise":
                K.TOPS.record = []
            else:
                K.TOPS.forced = [t_.clone() for t_ in out["layerwise"]["tops"]]
                K._TiledStack.forward = staticmethod(lambda *a, **k: (fwd_calls.append(mode), real_fwd(*a, **k))[1])
            n_lw = len(lw_calls)
            x = x_cpu.to(DEV).requires_grad_()
            torch.manual_seed(11)
            y = enc(x)
            (y * w_cpu.to(DEV)).sum().backward()
            K.flush_weight_grads()
            torch.cuda.synchronize()
            if mode != "layerwise":
                assert not K.TOPS.forced
                assert (len(lw_calls) > n_lw) == (mode == "tiled_fwd_only"), "wrong backward path"
            grads = {n: p._rf_grad.detach().cpu().clone() for n, p in enc.named_parameters()}
            out[mode] = dict(y=y.detach().cpu(), dx=x.grad.detach().cpu(), grads=grads, tops=K.TOPS.record)
            K.TOPS.record, K.TOPS.forced = None, None
            K.SINK.active = False
    finally:
        K.TILED_STACK, K.TILED_STACK_BWD, K._TiledStack.forward, K._stack_backward_layerwise = True, True, real_fwd, real_lw
        K.TOPS.record, K.TOPS.forced = None, None
        K.SINK.active = False
    assert "tiled" in fwd_calls and "tiled_fwd_only" in fwd_calls, "the row-tiled stack was not taken"

    def compare(a, b_):
        rms = float(torch.cat([v.flatten() for v in b_["grads"].values()]).square().mean().sqrt())
        worst = max((float((a["grads"][n].double() - b_["grads"][n].double()).norm()
                           / max(float(b_["grads"][n].double().norm()), rms * b_["grads"][n].numel() ** 0.5 * 0.05)), n)
                    for n in b_["grads"])
        return rel_err(a["y"], b_["y"]), fro_err(a["dx"], b_["dx"]), worst

    e_y, e_dx, worst = compare(out["tiled"], out["layerwise"])
    print(f"tiled stack B={B} L={L} vs layer-by-layer: y rel err {e_y:.2e}, dx fro err {e_dx:.2e}, worst parameter gradient "
          f"{worst[0]:.2e} ({worst[1]})")
    # (ReLU stack: a handful of mask flips between the two bf16 roundings of the forward -- 4.9e-2 observed)
    assert e_y < TOL_BF16 and e_dx < 5e-2 and worst[0] < 8e-2, (e_y, e_dx, worst)
    e_y, e_dx, worst = compare(out["tiled"], out["tiled_fwd_only"])
    print(f"   row-tile backward vs layer-by-layer backward on the same saves: dx fro err {e_dx:.2e}, worst parameter gradient "
          f"{worst[0]:.2e} ({worst[1]})")
    assert e_y == 0.0 and e_dx < 2e-2 and worst[0] < 3e-2, (e_y, e_dx, worst)


def test_side_stream_branches_change_nothing():
    """Every sub-graph the engine moves to a side stream (target-side pass, gaze encoder, the GPS backbone decoder's
    encoder-independent block) computes what it computes on the main stream: one train step with all forks on
    (``RF_OVERLAP`` bits 0, 1, 3) against one with none, same host draws -- loss, trajectory and gradients agree up to the
    summation order of the fp32 atomics (LayerNorm / bias gradients)."""
    from conftest import fro_err
    from routeformer_amd import kernels as K
    from routeformer_amd.engine import TrainEngine
    K.set_precision("f32")
    saved = K.OVERLAP_MASK
    out = {}
    try:
        for mask in (0, 11):
            K.OVERLAP_MASK = mask
            model, cfg, sd, c = build_product_model("c2_small", DEV)
            item = case_item(c)
            item_d = {"train": _to_dev(item["train"]), "target": _to_dev(item["target"])}
            eng = TrainEngine(model)
            model.train()
            torch.manual_seed(5)
            res = eng._fwd_bwd(item_d, 10)
            torch.cuda.synchronize()
            out[mask] = (float(res["loss"].detach()), res["future_gps"].detach().clone(), eng.reducer.flat_grad.clone())
    finally:
        K.OVERLAP_MASK = saved
    (l0, f0, g0), (l1, f1, g1) = out[0], out[11]
    assert abs(l0 - l1) < 1e-5 * max(1.0, abs(l0)), (l0, l1)
    assert rel_err(f1, f0) < 1e-5
    assert fro_err(g1, g0) < 1e-4, fro_err(g1, g0)
