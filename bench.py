#!/usr/bin/env python3
"""bench.py -- samples/s of one full Routeformer train step on synthetic GEM-shaped batches.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = forward(input) + target-side feature forward + trajectory/dense losses + backward
(+ bucketed RCCL gradient all-reduce overlapped with backward when N > 1) + global-norm clip + AdamW,
i.e. the reference's ``training_step`` + optimizer step (experiments/full_comparison.py:470-532,
681-711, 829-830) for ONE Routeformer.  Workload = BASELINE.json configs[1] ("C2"): full model
(GPS + left/right scene video + front video + gaze), 224x224, 8 s history -> 6 s future, paper
hyper-parameters, batch 8 PER GPU (weak scaling: configs[2] is the same at N=8).
Inputs are resident in HBM before the timed region starts.

Prints ONE JSON line on rank 0 (metric/value/... + "roofline" for the dominant kernel class +
"cpu_baseline" = the CPU oracle timed on this box's host cores at N=1).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}  # dense peaks, same guide


def build(case_name, device, precision, dropout="none"):
    from routeformer_amd import kernels as K, presets, synthetic
    from routeformer_amd.models import Routeformer, RouteformerConfig
    from routeformer_amd.models.gps_backbone import GPSBackboneConfig, Informer
    from routeformer_amd.models.video_backbone import HRNet16Backbone, VideoBackboneConfig

    K.set_precision(precision)
    c = presets.case(case_name)
    if dropout == "paper":
        c["rf"] = dict(c["rf"], feature_dropout=0.05, view_dropout=0.6, gaze_dropout=0.2)
    _, cfg = presets.build_configs(c, GPSBackboneConfig, RouteformerConfig, VideoBackboneConfig)
    torch.manual_seed(0)
    model = Routeformer(cfg, gps_backbone=Informer, video_backbone=HRNet16Backbone if cfg.with_video else None)
    sd = synthetic.synth_state_dict(model.state_dict(), 7)  # random-init weights of that architecture
    model.load_state_dict(sd)
    return model.to(device), cfg, sd, c


def make_item(c, rank, device=None, case_id=2):
    from routeformer_amd import synthetic
    item = synthetic.synth_item(c["B"], c["T"], c["P"], 1000 * case_id + rank, c["H"], c["W"], streams=c["streams"],
                                gaze=c["gaze"])
    if device is not None:
        item = {k: {n: v.to(device) for n, v in d.items()} for k, d in item.items()}
    return item


def cpu_baseline(cfg, sd, c, steps=10, warm=3):
    """The CPU oracle (kind "port": our restatement of the reference, pinned to the reference's golden
    vectors) doing the SAME full train step on the host cores.  Bounded sample: `warm` untimed
    steps, then `steps` timed steps of the B=8 workload; the MEDIAN step time is reported."""
    from oracle import routeformer_oracle as O
    # the GPU box gives a 16-core share per GPU (cpu_count() reports the whole host): oversubscribing stalls
    cores = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    torch.set_num_threads(cores)
    item = make_item(c, 0)
    params = {k: v.clone().requires_grad_(v.is_floating_point() and "video_backbone" not in k and "running" not in k
                                          and not k.endswith(".pe")) for k, v in sd.items()}
    train = [p for p in params.values() if p.requires_grad]
    opt = torch.optim.AdamW(train, lr=cfg.lr, weight_decay=cfg.wd)
    times = []
    for i in range(warm + steps):
        t0 = time.perf_counter()
        torch.manual_seed(i)
        orc = O.OracleRouteformer(cfg, params, training=True)
        res = orc.train_step(item, epoch=10)
        opt.zero_grad(set_to_none=True)
        res["loss"].backward()
        torch.nn.utils.clip_grad_norm_(train, 2.5)
        opt.step()
        times.append(time.perf_counter() - t0)
        print(f"[cpu_baseline] step {i}{' (warm-up)' if i < warm else ''}: {times[-1]:.2f} s on {cores} threads",
              file=sys.stderr, flush=True)
    timed = sorted(times[warm:])
    dt = timed[len(timed) // 2] if len(timed) % 2 else 0.5 * (timed[len(timed) // 2 - 1] + timed[len(timed) // 2])
    return {"value": c["B"] / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"median of {steps} full train steps (after {warm} warm-up steps) of the same C2 batch-{c['B']} "
                      f"workload, fp32, torch CPU oracle, {dt:.2f} s/step (min {timed[0]:.2f}, max {timed[-1]:.2f})"}


def site_len_stack(site, cfg):
    """L_Q of a ProbSparse call site (L_Q, sample_k) if it belongs to a Perceive encoder / decoder stack (their factor is
    always 5, cross_modal_transformer.py:379; the paper's Informer uses 4), else -1."""
    import math
    L = site[0]
    return L if site[1] == min(5 * int(math.ceil(math.log(L))), L) and cfg.gps_backbone_config.factor != 5 else -1


def ade_vs_cpu_ref(model, cfg, items, precision, n=16, seeds=(1234, 4321), medium=True):
    """BASELINE.json's second metric, "ADE vs CPU ref": the L2 distance in METRES between the trajectories this model
    (HIP kernels, its current weights) and the CPU oracle (same weights, same seed -> same host-RNG key samples)
    predict, eval mode, over `n` samples of the bench batches x `seeds` -- a distribution, not two samples.
    Per arithmetic mode (fp32 MFMA / bf16 MFMA):
      free     ProbSparse top-u selections made by the kernels (what a user gets);
      imposed  the oracle's selections imposed (isolates the arithmetic from the discontinuous selection);
      flips    how often the kernels' selection differs from the oracle's when every call sees inputs that are still
               on the oracle's trajectory (kernels.TOPS.shadow: each call runs free first, then imposed), per
               attention site (L_Q x sample_k) -- the rate at which the selection leaves the reference's path.
    `rel` = max |error| / trajectory scale per sample (north_star: 1e-3 fp32, 1e-2 bf16).  Outside the timed region."""
    from oracle import routeformer_oracle as O
    from routeformer_amd import kernels as K
    keys = list(items[0]["train"].keys())
    pool = {k: torch.cat([it["train"][k] for it in items], dim=0)[:n] for k in keys}
    n = pool[keys[0]].shape[0]
    sd = {k: v.detach().float().cpu().clone() if v.is_floating_point() else v.detach().cpu().clone()
          for k, v in model.state_dict().items()}
    was_training = model.training
    model.eval()
    res = {"samples": n, "seeds": list(seeds), "unit": "m"}
    acc = {}
    t_cpu = t_med = 0.0
    try:
        for seed in seeds:
            src = O.IndexSource()
            torch.manual_seed(seed)
            t0 = time.perf_counter()
            with torch.no_grad():
                out = O.OracleRouteformer(cfg, sd, training=False, idx=src).forward({k: v.cpu() for k, v in pool.items()})
            t_cpu += time.perf_counter() - t0
            pos_o = (out[0] if isinstance(out, tuple) else out).double()
            scale = pos_o.abs().amax(dim=(1, 2)).clamp_min(1.0)  # per sample
            res["trajectory_scale_m"] = float(pos_o.abs().max())
            sites = [tuple(t.shape) for t in src.log]  # (L_Q, sample_k) of every ProbSparse call, reference order

            def count_flips(bucket, mine_list):
                """Selections differing from the fp32 oracle's, per attention site; ``bucket``: name -> [selections,
                flipped, rows, rows flipped]."""
                run = 0
                for ci, (site, mine, ref) in enumerate(zip(sites, mine_list, src.tops)):
                    run = run + 1 if (ci > 0 and sites[ci - 1] == site) else 0
                    diff = (mine.cpu().long().sort(dim=-1).values != ref.long().sort(dim=-1).values)
                    names = [f"L{site[0]}xk{site[1]}"]
                    # first layer of an encoder stack: the one call of a fused stack whose INPUT is teacher-forced
                    # (the free pass of a fused stack runs all its layers on its own selections)
                    if site[0] == site_len_stack(site, cfg) and run % cfg.encoder_layers == 0:
                        names.append(names[0] + ".first_layer")
                    for nm in names:
                        e = bucket.setdefault(nm, [0, 0, 0, 0])
                        e[0] += diff.any(dim=-1).numel(); e[1] += int(diff.any(dim=-1).sum())
                        e[2] += diff.numel(); e[3] += int(diff.sum())

            # the REFERENCE's own training precision -- torch.set_float32_matmul_precision("medium")
            # (full_comparison.py:48: bf16 operand rounding inside fp32 matmuls on a GPU) + cuDNN's TF32 convolutions -- as
            # an arithmetic mode of the CPU oracle (oracle.ARITH = "medium"), against the exact-fp32 oracle on the same
            # samples and seeds: free-running trajectories, and teacher-forced flip counts exactly as for the kernels
            if medium:
                try:
                    O.ARITH = "medium"
                    t0 = time.perf_counter()
                    for mode in ("free", "imposed"):
                        srm = O.IndexSource()
                        if mode == "imposed":
                            srm.forced = [t.clone() for t in src.tops]
                        torch.manual_seed(seed)
                        with torch.no_grad():
                            om = O.OracleRouteformer(cfg, sd, training=False, idx=srm).forward({k: v.cpu() for k, v in pool.items()})
                        pos = (om[0] if isinstance(om, tuple) else om).double()
                        d = (pos - pos_o).norm(dim=-1)
                        a = acc.setdefault(("medium_oracle", mode), {"ade": [], "max": [], "rel": []})
                        a["ade"] += d.mean(dim=1).tolist()
                        a["max"] += d.amax(dim=1).tolist()
                        a["rel"] += ((pos - pos_o).abs().amax(dim=(1, 2)) / scale).tolist()
                        if mode == "imposed":
                            count_flips(acc.setdefault(("medium_oracle", "flips"), {}), srm.tops)
                    t_med += time.perf_counter() - t0
                finally:
                    O.ARITH = None
            for prec in ("f32", "bf16"):
                K.set_precision(prec)
                for mode in ("free", "imposed"):
                    K.TOPS.forced = [t.clone() for t in src.tops] if mode == "imposed" else None
                    K.TOPS.shadow = [] if mode == "imposed" else None
                    torch.manual_seed(seed)
                    with torch.no_grad():
                        o = model(pool)
                    pos = (o[0] if isinstance(o, tuple) else o).double().cpu()
                    d = (pos - pos_o).norm(dim=-1)                                   # (n, P) metres
                    rel = (pos - pos_o).abs().amax(dim=(1, 2)) / scale               # (n,)
                    a = acc.setdefault((prec, mode), {"ade": [], "max": [], "rel": []})
                    a["ade"] += d.mean(dim=1).tolist()
                    a["max"] += d.amax(dim=1).tolist()
                    a["rel"] += rel.tolist()
                    if mode == "imposed":
                        shadow, K.TOPS.shadow = K.TOPS.shadow, None
                        count_flips(acc.setdefault((prec, "flips"), {}), shadow)
    finally:
        K.TOPS.forced, K.TOPS.shadow = None, None
        K.set_precision(precision)
        model.train(was_training)
    res["cpu_forward_s"] = round(t_cpu / len(seeds), 2)
    out_med = {}
    for (prec, mode), a in acc.items():
        dst, key = (out_med, mode) if prec == "medium_oracle" else (res, f"{prec}_{mode}")
        if mode == "flips":
            tot = [sum(e[i] for k, e in a.items() if not k.endswith(".first_layer")) for i in range(4)]
            first = [sum(e[i] for k, e in a.items() if k.endswith(".first_layer")) for i in range(2)]
            dst[key] = {"selections": tot[0], "flipped": tot[1], "rate": tot[1] / max(tot[0], 1),
                        "first_layer_rate": first[1] / max(first[0], 1), "rows": tot[2], "rows_flipped": tot[3],
                        "by_site": {k: {"selections": e[0], "flipped": e[1]} for k, e in a.items()}}
        else:
            r = sorted(a["rel"])
            dst[key] = {"ade": sum(a["ade"]) / len(a["ade"]), "max": max(a["max"]),
                        "rel_median": r[len(r) // 2], "rel_p90": r[int(0.9 * (len(r) - 1))], "rel_max": r[-1],
                        "within_tolerance": sum(x <= (1e-3 if prec == "f32" else 1e-2) for x in r) / len(r)}
    if out_med:
        out_med["what"] = ("the CPU oracle with the reference's GPU training precision -- bf16-rounded matmul operands "
                           "(set_float32_matmul_precision('medium'), full_comparison.py:48), TF32-rounded convolution operands "
                           "(cuDNN default), fp32 accumulation -- against the exact-fp32 CPU oracle, same samples and seeds: "
                           "free = its own selections, imposed = the fp32 oracle's, flips counted teacher-forced like the kernels'")
        out_med["cpu_forward_s"] = round(t_med / len(seeds), 2)
        res["medium_oracle_vs_f32_oracle"] = out_med
        # is the product's bf16 mode inside what the reference's own precision mode does?
        if "bf16_free" in res and "free" in out_med:
            res["bf16_free_vs_medium_oracle_free"] = {
                k: {"product_bf16": res["bf16_free"][k], "medium_oracle": out_med["free"][k]}
                for k in ("rel_median", "rel_p90", "rel_max", "within_tolerance")}
            res["bf16_free_vs_medium_oracle_free"]["flip_rate"] = {
                "product_bf16": res.get("bf16_flips", {}).get("rate"), "medium_oracle": out_med.get("flips", {}).get("rate")}
    return res


def batch_independence(model, item, precision, seed=4321):
    """Eval-mode `future_gps` of sample 0 of the bench batch vs the same sample forwarded alone with the same seed
    (per-sample ops, BatchNorm on running statistics, one shared key-sample table per call as in the reference):
    max |difference| relative to the trajectory scale, for the timed weights and the timed batch.  -> {"f32": ...,
    <timed precision>: ...}.  In exact-fp32 mode the two forwards agree to rounding (bound 1e-3 = north_star's fp32
    tolerance).  In bf16 mode a batch of 1 and a batch of 8 take different launch shapes (row-block height, skinny vs
    tiled GEMM, split-K depth): their fp32 accumulation orders differ at 1e-7, which the discontinuous top-u selection
    can turn into a different set of active queries -- reported, bounded by 5e-2 (see ade_vs_cpu_ref for the
    distribution of such selection effects)."""
    from routeformer_amd import kernels as K
    was_training = model.training
    model.eval()
    res = {}
    try:
        for prec in dict.fromkeys(("f32", precision)):
            K.set_precision(prec)
            outs = []
            for bt in (item["train"], {k: v[:1] for k, v in item["train"].items()}):
                torch.manual_seed(seed)
                with torch.no_grad():
                    o = model(bt)
                outs.append((o[0] if isinstance(o, tuple) else o)[0].double())
            res[prec] = float((outs[0] - outs[1]).abs().max() / max(1.0, float(outs[0].abs().max())))
    finally:
        K.set_precision(precision)
        model.train(was_training)
    return res


def self_launch(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start N rank processes of this script
    through torch.distributed.run (fresh children -- this parent has not touched the GPU and never does), relay
    rank 0's JSON line, return the children's status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:
        if ln.lstrip().startswith("{") and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    return rc if rc != 0 else (0 if line is not None else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--case", default="C2")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay (N=1)")
    ap.add_argument("--cpu-steps", type=int, default=10)
    ap.add_argument("--no-ade", action="store_true", help="skip the ADE-vs-CPU-reference leg (N=1, rank 0)")
    ap.add_argument("--ade-samples", type=int, default=16, help="samples (x 2 seeds) of the ADE-vs-CPU-reference leg")
    ap.add_argument("--trunk-cache", action="store_true",
                    help="attach the HBM-resident backbone-feature cache (the reference's @torchcache steady state: the "
                         "frozen trunk is skipped for frames it has seen) -- a second bench line, not the headline")
    ap.add_argument("--dropout", default="none", choices=["none", "paper"],
                    help="paper: the reference run's dropouts (full_comparison.py:272-275: feature 0.05, view 0.6, "
                         "gaze 0.2) instead of the parity configuration's zeros -- a second bench line, not the headline")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    backend = os.environ.get("RF_DIST_BACKEND", "nccl")  # "gloo": single-GPU rehearsal of the N>1 path
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # RF_REHEARSE_COLLECTIVES=1 at N=1: a one-rank RCCL group and the engine's whole N>1 code path (bucketed
    # all-reduce launches, two-graph step) on a single GPU -- a rehearsal, never the reported N=1 line
    rehearse = world == 1 and os.environ.get("RF_REHEARSE_COLLECTIVES") == "1"
    multi = world > 1 or rehearse
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        # RCCL prints a version banner on STDOUT when it builds its communicator; stdout is reserved for the one
        # JSON line, so file descriptor 1 points at stderr until the communicator exists
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    from routeformer_amd import kernels as K
    from routeformer_amd.engine import GraphedTrainEngine, TrainEngine

    def note(msg):
        print(f"[bench r{rank}] {msg}", file=sys.stderr, flush=True)

    model, cfg, sd, c = build(args.case, device, args.precision, args.dropout)
    if args.trunk_cache and cfg.with_video:
        from routeformer_amd.models.video_backbone import TokenCache
        model.video_backbone.token_cache = TokenCache(4096, device)
    # two different synthetic batches, used alternately: the engine's look-ahead (conv trunk of the NEXT
    # batch under the current step) then always works on data it has not seen in this step
    items = [make_item(c, rank, device), make_item(c, rank + 500, device)]
    for i, it in enumerate(items):
        it["id"] = i  # explicit batch ids: the engine's look-ahead never keys on tensor addresses
    item = items[0]
    note("model + 2 synthetic batches resident in HBM")
    use_graph = not args.no_graph
    # RF_DEFER_UPDATE=1: step k's clip + AdamW replayed at the head of step k+1's graph, its GPS-backbone share on a
    # side stream under the camera/gaze/fusion encoders (engine.GraphedTrainEngine).  Measured SLOWER (10.30 vs 10.07
    # ms/step: the streaming update slows the latency-bound encoder kernels by more than it saves), so it is off.
    # clip + AdamW of step k at the head of step k + 1's replay, the GPS backbone's share on a side stream underneath
    # the camera / gaze / fusion encoders (same arithmetic, same order; the flush() calls below keep exactly K updates
    # inside the timed region).  Measured A/B at N = 1: 7.00 / 6.95 -> 6.84 / 6.80 ms.  With gaze dropout the update is launched per
    # segment, each with its own device-side "pending" flag and update count.  Not available with the sharded gradient exchange.
    want = os.environ.get("RF_DEFER_UPDATE", "auto")
    defer = (want == "1") or (want == "auto" and os.environ.get("RF_DP_MODE", "allreduce") == "allreduce")
    engine = GraphedTrainEngine(model, defer_update=defer) if use_graph else TrainEngine(model)
    if use_graph:
        # capture once up front; fall back step by step: two-graph step (N > 1) -> one graph -> eager launches.
        # The decision is COLLECTIVE (a rank that failed alone would otherwise issue a different sequence of
        # collectives than its peers: engine construction broadcasts the parameters): every rank reports, the MAX
        # is all-reduced, and all ranks switch mode together.
        from routeformer_amd.models.blocks import SAMPLER

        def any_rank_failed(failed: bool) -> bool:
            if not multi:
                return failed
            flag = torch.tensor([1.0 if failed else 0.0], device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            return bool(flag.item() > 0)

        for attempt in ("as configured", "single graph"):
            failed = False
            try:
                engine.capture(item, epoch=10)
            except Exception as exc:  # noqa: BLE001
                failed = True
                print(f"[bench r{rank}] HIP-graph capture failed ({attempt}; {type(exc).__name__}: {exc})", file=sys.stderr,
                      flush=True)
            if not any_rank_failed(failed):
                break
            SAMPLER.drop_static()
            if attempt == "as configured" and engine.split:
                engine = GraphedTrainEngine(model, defer_update=defer)
                engine.split = False
            else:
                print(f"[bench r{rank}] using eager launches", file=sys.stderr, flush=True)
                use_graph = False
                engine = TrainEngine(model)
                break

    def sync():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    if use_graph and args.dropout != "none":
        # the host dropout decisions select among graph variants: hold all of them before the clock starts (a training
        # run meets every variant within its first few hundred steps; a 10-step window would time the captures)
        note(f"{engine.precapture()} graph variants captured")
    note(f"engine ready ({'hipGraph' if use_graph else 'eager'}), warm-up")
    for i in range(args.warmup):
        engine.step(items[i % 2], epoch=10, next_item=items[(i + 1) % 2])
    flush = getattr(engine, "flush", lambda: None)
    flush()  # a deferred update of the last warm-up step is applied outside the timed region ...
    sync()
    note("timed region")
    w0 = args.warmup
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == args.steps - 1 and not use_graph:
            K.PROFILE.enable()  # HIP events around every kernel-class launch of the last timed step
        res = engine.step(items[(w0 + i) % 2], epoch=10, next_item=items[(w0 + i + 1) % 2])
    flush()  # ... and the one of the last timed step inside it: exactly K forward/backward passes and K updates
    sync()
    elapsed = time.perf_counter() - t0
    K.PROFILE.disable()
    if use_graph:
        # Graph replay has no per-launch host hook: time the kernel classes live on the same stream in one
        # extra eager pass right after the timed region (same shapes, same data, HIP events per launch).
        from routeformer_amd.models.blocks import SAMPLER
        SAMPLER.rewind_static()
        K.PROFILE.enable()
        engine._eager_fwd_bwd(engine._static_item, 10)
        K.PROFILE.disable()
        SAMPLER.drop_static()  # the legs below run the model outside the engine: ordinary host draws again
    rank_ms = None
    if multi:
        mine = torch.tensor([elapsed], device=device, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [float(t) for t in every]
        elapsed = max(per_rank)  # the MAX over ranks is what counts
        rank_ms = {"min": min(per_rank) / args.steps * 1e3, "max": max(per_rank) / args.steps * 1e3,
                   "per_rank": [round(t / args.steps * 1e3, 4) for t in per_rank]}
    assert torch.isfinite(res["loss"]).item(), "loss is not finite"
    for k in ("ade", "fde"):
        assert torch.isfinite(res[k]).all().item(), f"{k} is not finite"
    rccl_ranks = dist.get_world_size() if multi else 1
    # a line for N GPUs is only printed when N ranks actually took part in the RCCL exchange.  The gloo
    # rehearsal of tests/test_gpu_dp.py (RF_DIST_BACKEND=gloo: two ranks sharing one GPU) says so in the line ("rehearsal").
    ranks_on_devices = 1
    if world > 1:
        devs = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
        dist.all_gather(devs, torch.tensor([local], dtype=torch.int64, device=device))
        ranks_on_devices = len({int(d) for d in devs})
    rehearsal_line = world > 1 and (backend != "nccl" or ranks_on_devices != world)
    if world > 1 and rccl_ranks != world:
        if rank == 0:
            print(f"bench.py: --gpus {world} but the process group has {rccl_ranks} ranks: no line printed", file=sys.stderr)
        dist.destroy_process_group()
        sys.exit(3)
    if rehearsal_line and os.environ.get("RF_DIST_BACKEND") is None:
        if rank == 0:
            print(f"bench.py: --gpus {world} ranks share {ranks_on_devices} device(s) (backend {backend}): not a {world}-GPU "
                  f"measurement, no line printed (set RF_DIST_BACKEND explicitly for a rehearsal)", file=sys.stderr)
        dist.destroy_process_group()
        sys.exit(3)
    # more than "finite": the timed model must treat the samples of its batch independently (eval forward of sample
    # 0 inside the batch == the same sample alone, same seed; fp32 mode, bound 1e-3 = north_star's fp32 tolerance)
    indep = batch_independence(model, item, args.precision)
    assert indep["f32"] < 1e-3, f"sample 0 of the bench batch depends on its batch mates: rel diff {indep['f32']:.3e}"
    assert indep[args.precision] < 5e-2, f"batch dependence in the timed mode: rel diff {indep[args.precision]:.3e}"

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        total_samples = c["B"] * world * args.steps
        prof = K.PROFILE.summary()
        roof, top = None, []
        if prof:
            # dominant kernel = largest per-step total among the kernel symbols, measured live with HIP events on the
            # launch stream that bracket each re-issued launch of that symbol (K.PROFILE.refine, rf_kernel_timer_arm)
            ranked = sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])
            # candidates: the five largest single-kernel symbols by eager event time (tags with a "+" are calls
            # that launch two kernels); each is re-timed launch by launch with start/stop events bracketing exactly
            # the dispatch, and the largest refined total wins -- the raw eager event times include the Python launch
            # path between kernels
            best = None
            for cand, st in [kv for kv in ranked if "+" not in kv[0]][:14]:
                fine = K.PROFILE.refine(cand)
                if fine is None:  # no replayable launches recorded for this symbol: its eager figure is not comparable
                    continue
                n_l, us, fl, by = fine
                ai = fl / max(by, 1)
                mf = ai > (MFMA_PEAK_TFLOPS[args.precision] * 1e12 / (HBM_PEAK_GBS * 1e9))
                top.append({"kernel": cand, "launches_per_step": n_l, "avg_us": round(us / n_l, 2),
                            "total_us_per_step": round(us, 1), "bound": "mfma" if mf else "hbm",
                            "achieved_gbs": round(by / (us * 1e-6) / 1e9, 1), "achieved_tflops": round(fl / (us * 1e-6) / 1e12, 2),
                            "frac": round((fl / (us * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS[args.precision]) if mf
                                          else (by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS), 4)})
                if best is None or fine[1] > best[1][1]:
                    best = (cand, fine)
            if best is None:
                cand, st = ranked[0]
                best = (cand, (st["launches"], st["total_ms"] * 1e3, st["flops"], st["bytes"]))
            name, (launches, total_us, flops, nbytes) = best
            avg_s = total_us / launches * 1e-6
            gbs = nbytes / launches / avg_s / 1e9
            tfl = flops / launches / avg_s / 1e12
            peak_tf = MFMA_PEAK_TFLOPS[args.precision]
            # regime: arithmetic intensity of the algorithmic work vs the ridge point
            mfma_bound = (flops / max(nbytes, 1)) > (peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9))
            # HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE run
            # separately, gfx950 x2 read correction applied; bench.py itself cannot collect counters)
            traffic, traffic_src = None, None
            try:
                for rnd in ("r04", "r03", "r02", "r01"):  # the newest committed PMC pass that knows this kernel
                    path = os.path.join(ROOT, "profiles", rnd, "pmc_traffic.json")
                    if os.path.exists(path):
                        with open(path) as fh:
                            kern = json.load(fh)["kernels"]
                        traffic = kern.get(name, {}).get("hbm_bytes_per_launch")
                        if traffic is None and name == "adamw_clip_kernel<true>" and "adamw_clip_kernel<false>" in kern:
                            # the PMC passes run the eager engine, whose update is ONE launch of the by-value variant over
                            # the whole buffer; the replayed step covers the same buffer with `launches` slices of the
                            # device-scalar variant (same loop body): per launch = the whole-buffer figure / launches
                            traffic = kern["adamw_clip_kernel<false>"]["hbm_bytes_per_launch"] / launches
                        if traffic is not None:
                            traffic_src = f"profiles/{rnd}/pmc_traffic.json"
                            break
            except (OSError, ValueError, KeyError):
                pass
            roof = {"kernel": name + " (librf_hip.so, anonymous namespace)", "bound": "mfma" if mfma_bound else "hbm",
                    "achieved": tfl if mfma_bound else gbs, "peak": peak_tf if mfma_bound else HBM_PEAK_GBS,
                    "unit": "TFLOP/s" if mfma_bound else "GB/s",
                    "frac": (tfl / peak_tf) if mfma_bound else (gbs / HBM_PEAK_GBS), "traffic": traffic,
                    "traffic_unit": f"HBM bytes per launch (PMC, {traffic_src})",
                    "algorithmic_bytes_per_launch": nbytes / launches, "launches_per_step": launches, "avg_us": avg_s * 1e6,
                    "flop_per_byte": flops / max(nbytes, 1), "alt_tflops": tfl, "alt_gbs": gbs,
                    "eager_event_ms_by_kernel": {k: round(v["total_ms"], 3) for k, v in ranked[:12]}}
        # the fused per-sequence encoder stack (csrc/seqlayer.hip) against the bf16 matrix-core
        # peak, per launch shape: the one kernel family of the step whose operands stay on chip
        fused = []
        for tag in [k for k in prof if k.startswith("seq_stack_")]:  # forward and backward
            fine = K.PROFILE.refine(tag)
            if fine is None:
                continue
            n_l, us, fl, by = fine
            fused.append({"kernel": tag, "bound": "mfma", "launches_per_step": n_l, "avg_us": us / n_l,
                          "achieved": fl / (us * 1e-6) / 1e12, "peak": MFMA_PEAK_TFLOPS["bf16"], "unit": "TFLOP/s",
                          "frac": fl / (us * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS["bf16"],
                          "algorithmic_gbs": by / (us * 1e-6) / 1e9, "flop_per_byte": fl / max(by, 1)})
        out = {
            "metric": "samples/sec (train step) on synthetic GEM batch",
            "value": total_samples / elapsed, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"{args.case}: full Routeformer (GPS+left/right/front video+gaze), "
                                   f"{c['H']}x{c['W']}, T={c['T']}->P={c['P']}, paper hyper-params, "
                                   f"batch {c['B']}/GPU, random-init weights, frozen HRNet-16 encoder",
                       "global_batch": c["B"] * world, "parallelism": f"dp{world}",
                       "trunk": ("HBM token cache: every frame of the two alternating batches is resident after the warm-up, "
                                 "the frozen conv trunk is skipped (the reference's torchcache steady state)"
                                 if args.trunk_cache else "runs every step (uncached: the headline configuration)"),
                       "dropout": ({"feature": cfg.feature_dropout, "view": cfg.view_dropout, "gaze": cfg.gaze_dropout,
                                    "gps_backbone": cfg.gps_backbone_config.dropout}
                                   if args.dropout != "none" else "0 (parity configuration, SURVEY 8(d))"),
                       "step": "fwd + target-feature fwd + losses + bwd + grad all-reduce + clip + AdamW",
                       "launch": ("hipGraph replay of fwd+bwd" + (" (+ the previous step's clip/AdamW at its head)" if defer else ""))
                       if use_graph else "eager launches"},
            "loss": float(res["loss"].detach()), "rccl_ranks": rccl_ranks,
            **({"rehearsal": f"{world} ranks over {backend} on {ranks_on_devices} device(s): NOT a {world}-GPU measurement"}
               if rehearsal_line else {}), "batch_independence_rel": indep["f32"], "batch_independence_rel_timed_mode": indep[args.precision],
            "roofline": roof, "roofline_top": top or None, "roofline_fused_encoder_stack": fused or None,
        }
        if multi:
            out["config"]["gradient_exchange"] = {
                "mode": os.environ.get("RF_DP_MODE", "allreduce"),
                "transport": "rf_comm_* (own RCCL communicator + communication stream)" if os.environ.get("RF_DP_COMM") == "rf"
                else f"torch.distributed process group ({backend})",
                "two_graph_step": bool(getattr(engine, "split", False)), "deferred_update": bool(defer)}
            out["ms_per_step_by_rank"] = rank_ms
        if rehearse:
            out["config"]["rehearsal"] = "one-rank RCCL group, N>1 code path (RF_REHEARSE_COLLECTIVES=1)"
        if world == 1 and not rehearse and not args.no_ade and cfg.with_video:
            out["ade_vs_cpu_ref"] = ade_vs_cpu_ref(model, cfg, items, args.precision, n=args.ade_samples)
        if world == 1 and not args.no_cpu_baseline and not rehearse:
            out["cpu_baseline"] = cpu_baseline(cfg, sd, c, args.cpu_steps)
        print(json.dumps(out))
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
