"""Training-step engine: the reference's step recipe + MI355X-native data parallelism.

* ``train_step_losses``  -- the Routeformer branch of ``ParallelTrainer.training_step``
  (``experiments/full_comparison.py:476-521``): forward, target-side feature pass, trajectory /
  dense SmoothL1 losses with the epoch-gated dense weight, ADE / FDE.
* ``GradReducer``        -- one process per GPU; all trainable gradients live in ONE flat fp32 buffer,
  cut into buckets in reverse-forward order; each bucket is all-reduced (RCCL over xGMI through
  ``torch.distributed``) as soon as autograd has produced its last gradient, on the communicator's own
  HIP stream, overlapping the rest of backward.  Replaces Lightning's ``DDPStrategy(nccl)``
  (``full_comparison.py:794,832``).  No data-path collective other than this one exchange.
* ``FusedAdamW``         -- global-norm clip (2.5) + AdamW over the flat buffers in two launches
  (``full_comparison.py:694-702,829-830``).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from routeformer_amd.losses import FutureDiscountedLoss
from routeformer_amd.score import ade, fde


# Weight-gradient groups that fill up mid-backward go to a side stream (-0.04 .. -0.09 ms at C2).  Round 3 switched this
# off after a host crash inside hipGraphLaunch that only showed with it; the cause was elsewhere (graph execs destroyed
# under PyTorch's HIP runtime, see _new_graph below) -- the extra branch merely changed which freed object the launch hit.
WGRAD_SIDE = os.environ.get("RF_WGRAD_SIDE", "1") == "1"
# Deferred update, one process: RF_EARLY_SUMSQ=1 sums the GPS backbone's share of the clip norm mid-backward on a side stream
# (TrainEngine._backward_gps_first) instead of in the whole-buffer pass at the head of the next step (68 us of HBM time in
# front of the encoders' update).  Measured neutral to slightly slower (5.35 vs 5.39 ms over four alternating pairs, 5.41 vs
# 5.49 with the backbone's weight gradients flushed at its input): the sum contends with the backward chain for the time it
# saves at the head.  Off by default; kept as a measurement switch (tests run it explicitly).
EARLY_SUMSQ = os.environ.get("RF_EARLY_SUMSQ", "0") == "1"
# Deferred update: RF_UPDATE_AFTER_EMBED=1 forks the GPS backbone's streaming share behind the camera-token embedding instead of at
# the head of the step.  The kernel trace suggested it (the embedding's im2col + product take 255 + 129 us next to the update, 25 + 34
# alone) -- but the profiler's timeline is not the replay's: measured 5.250 / 5.272 / 5.274 -> 5.416 / 5.408 / 5.490 ms.  Off.
UPDATE_AFTER_EMBED = os.environ.get("RF_UPDATE_AFTER_EMBED", "0") == "1"
EARLY_SUMSQ_FLUSH = os.environ.get("RF_EARLY_SUMSQ_FLUSH", "0") == "1"  # flush the backbone's queued weight gradients at its input


# ---- captured graphs are never destroyed ------------------------------------------------------------------------
# The HIP runtime PyTorch 2.10+rocm7.0 ships (libamdhip64 7.0.2, the one this process runs on whatever /opt/rocm holds)
# leaves OTHER graph execs with dangling parallel-stream pointers when a multi-branch hipGraphExec is destroyed: a later
# hipGraphLaunch -- of a new exec or of one that was alive all along -- walks freed hip::Stream objects (GraphExec::Run ->
# Graph::UpdateStreams, fault at libamdhip64.so+0xaee41) and segfaults once the allocator has reused them.  Stand-alone
# repro without PyTorch: tools/probes/graph_width_probe.hip (crashes within tens of rounds under that runtime as soon as
# older execs are destroyed, never when none is, never under ROCm 7.2's runtime); this was round 3's "host segmentation
# fault inside hipGraphLaunch" (DESIGN section 5b).  So: every torch.cuda.CUDAGraph this engine creates gets one reference
# that is never given back (no destructor, not even at interpreter exit), and what a dropped graph held of the engine's
# memory pool is handed back by hand (``_retire``).
_KEPT_GRAPHS: list = []


def _new_graph() -> "torch.cuda.CUDAGraph":
    import ctypes
    g = torch.cuda.CUDAGraph()
    _KEPT_GRAPHS.append(g)
    ctypes.pythonapi.Py_IncRef(ctypes.py_object(g))
    return g


def _retire(g, device) -> None:
    """The engine drops graph ``g`` (re-capture after a recipe change, engine teardown): the exec stays alive (see above),
    its use count on the memory pool is released so that the pool's blocks return to the allocator once every graph that
    shared the pool has been retired.  A retired graph must never be replayed again."""
    for one in (g if isinstance(g, tuple) else (g,)):
        if one is None or getattr(one, "_rf_retired", False):
            continue
        one._rf_retired = True
        try:
            torch._C._cuda_releasePool(device.index if device.index is not None else torch.cuda.current_device(), one.pool())
        except Exception:  # (a graph whose capture never began has no pool)
            pass


def _capture_kw() -> dict:
    """Stream-capture options.  In a multi-rank job the process group's watchdog thread polls its events with
    hipEventQuery at arbitrary moments; under the default "global" capture mode such a call from ANOTHER thread
    invalidates a capture in progress.  "thread_local" keeps the checks for this thread (the one capturing)."""
    if dist.is_available() and dist.is_initialized():
        return {"capture_error_mode": "thread_local"}
    return {}


def train_step_losses(model, item, epoch: int = 0, trajectory_loss: Optional[FutureDiscountedLoss] = None,
                      dense_loss: Optional[FutureDiscountedLoss] = None,
                      tokens_ready: bool = False) -> Dict[str, torch.Tensor]:
    """item = {"train": Data, "target": Data}.  Returns loss terms, metrics and predictions.
    ``tokens_ready``: the conv-trunk tokens of this item were installed already (pipelined engine)."""
    cfg = model.configs
    tl = trajectory_loss or FutureDiscountedLoss(cfg.discount_factor, cfg.epsilon, loss_function="smooth_l1")
    dl = dense_loss or FutureDiscountedLoss(cfg.discount_factor, cfg.visual_epsilon, loss_function="smooth_l1")
    tl.current_epoch = dl.current_epoch = epoch
    target_gps = item["target"]["gps"].to(torch.float32)
    res: Dict[str, torch.Tensor] = {}
    if cfg.dense_prediction:
        from routeformer_amd import kernels as K
        overlap = K.OVERLAP and (K.OVERLAP_MASK & 1) and target_gps.is_cuda
        if K.OVERLAP and not tokens_ready and hasattr(model, "prefetch_video_tokens"):
            # frozen conv trunk: one pass over the history AND target frames (336 images at B=8)
            model.prefetch_video_tokens([item["train"], item["target"]])
        if overlap:  # fork point: the target-side pass must not wait for the input forward
            side = K.fork_side_stream("target")
        fused = (K.OVERLAP and target_gps.is_cuda and hasattr(model, "forward_raw") and model.training
                 and not cfg.autoregressive and tl.loss_function == "smooth_l1" and dl.loss_function == "smooth_l1"
                 and item["train"]["gps"].dtype == torch.float32)
        if fused:
            raw, last_gps = model.forward_raw(item["train"])
        else:
            future_gps, future_vis = model(item["train"])
        # host order stays "input forward, then target pass" (reference draw order, full_comparison.py:481-482);
        # on the device the two are independent, so the target pass gets its own stream
        with torch.no_grad():  # the reference detaches this branch (full_comparison.py:495)
            if overlap:
                with torch.cuda.stream(side):
                    _, target_vis = model.preprocess_batch(item["target"], training=False)
                torch.cuda.current_stream().wait_stream(side)
            else:
                _, target_vis = model.preprocess_batch(item["target"], training=False)
        if fused:  # postprocess + both losses + metrics: one launch each way
            E = cfg.image_embedding_size
            g_t, g_d = tl.discount(), dl.discount()
            assert g_t == g_d, "fused head assumes one discount table for both losses (as in the reference driver)"
            target_vis = target_vis[:, : raw.shape[1]]  # (read in place by the fused head: batch stride)
            loss, traj, dense, a_, f_, future_gps = K.traj_head(
                raw, last_gps, target_gps, target_vis, g_t, cfg.dense_loss_ratio, epoch >= 10,
                cfg.motion_std if cfg.normalize_motion else 1.0, cfg.motion_mean if cfg.normalize_motion else 0.0)
            res.update(dense_loss=dense, future_vis=raw[:, :, 2:2 + E], target_vis=target_vis, loss=loss, traj_loss=traj,
                       future_gps=future_gps, ade=a_, fde=f_)
            if hasattr(model, "clear_video_tokens") and not tokens_ready:
                model.clear_video_tokens()
            return res
        target_vis = target_vis[:, : future_vis.shape[1]]
        step = cfg.autoregressive_step_size
        if cfg.autoregressive:
            future_gps, target_gps = future_gps[:, :step], target_gps[:, :step]
        traj = tl(future_gps, target_gps)
        if cfg.autoregressive:
            traj = traj * (cfg.gps_backbone_config.pred_len / step)
            future_vis, target_vis = future_vis[:, :step], target_vis[:, :step]
        dense = dl(future_vis, target_vis)
        if epoch < 10:  # dense loss switched on after 10 epochs
            weight = 0
        else:
            weight = (cfg.dense_loss_ratio * traj / torch.clamp(dense, min=1e-6)).detach()
        loss = traj + weight * dense
        res.update(dense_loss=dense, future_vis=future_vis, target_vis=target_vis)
    else:
        future_gps = model(item["train"])
        traj = tl(future_gps, target_gps)
        loss = traj
    res.update(loss=loss, traj_loss=traj, future_gps=future_gps, ade=ade(future_gps, target_gps),
               fde=fde(future_gps, target_gps))
    if hasattr(model, "clear_video_tokens") and not tokens_ready:
        model.clear_video_tokens()
    return res


def eval_step(model, item, trajectory_loss: Optional[FutureDiscountedLoss] = None, epoch: int = 0, passes: int = 5,
              seed: int = 12345):
    """The reference's evaluation protocol (``ParallelTrainer._eval_step``, full_comparison.py:654-679):
    reseed the host RNG, average the predicted trajectory over ``passes`` forward passes (each draws fresh
    ProbSparse key samples), then per-sample loss / ADE / FDE.  Returns (losses, ades, fdes, mean
    trajectory), the first three of shape (B,)."""
    cfg = model.configs
    tl = trajectory_loss or FutureDiscountedLoss(cfg.discount_factor, cfg.epsilon, loss_function="smooth_l1")
    tl.current_epoch = epoch
    torch.manual_seed(seed)
    target_gps = item["target"]["gps"]
    runs = []
    with torch.no_grad():
        for _ in range(passes):
            out = model(item["train"])
            runs.append(out[0] if cfg.dense_prediction else out)
        future_gps = torch.stack(runs).mean(dim=0)
        losses, ades, fdes = [], [], []
        for i in range(future_gps.shape[0]):
            f, t = future_gps[i:i + 1], target_gps[i:i + 1]
            losses.append(tl(f, t)); ades.append(ade(f, t)); fdes.append(fde(f, t))
    torch.seed()
    return torch.stack(losses), torch.stack(ades), torch.stack(fdes), future_gps


def agree_unused(flag: bool, group) -> bool:
    """Data-parallel agreement on "this step did not use the gaze branch": true only if EVERY rank dropped it.
    DDP semantics of the reference (find_unused_parameters=True, full_comparison.py:794): a parameter one rank used
    gets the averaged gradient on all ranks and is updated everywhere; one that NO rank used keeps ``grad=None`` and
    its AdamW update is skipped everywhere.  ``group``: a host-side (gloo) process group -- the flag is known on the
    host before the step is launched, so the exchange costs no device synchronisation; None = single process."""
    if group is None:
        return flag
    t = torch.tensor([1 if flag else 0], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t))


def trainable_parameters(model) -> List[torch.nn.Parameter]:
    """Everything but the frozen video backbone (full_comparison.py:689-691), registration order."""
    return [p for n, p in model.named_parameters() if "video_backbone" not in n and p.requires_grad]


class GradReducer:
    """Flat gradient buffer + bucketed, overlapped all-reduce (mean over ranks).

    Device-agnostic (the 2-rank gloo test drives it on the CPU); on the GPU the process group is RCCL.
    Parameters are laid out in REVERSE registration order so buckets fill in roughly the order
    backward produces gradients (GPS backbone first, frame encoder last)."""

    def __init__(self, params: List[torch.nn.Parameter], bucket_mb: float = 32.0, group=None, groups=(), mode=None,
                 lead=None, coalesce=None):
        """``groups``: lists of parameters that must sit back to back, in the given order (e.g. the Q, K, V
        projection weights of one attention layer, so they can be used as ONE packed matrix).

        ``mode`` (default: env RF_DP_MODE or "allreduce") -- how the gradient exchange is done:
          "allreduce"   bucketed ``all_reduce`` (RCCL ring / tree), every rank then updates every parameter;
          "direct"      direct reduce-scatter: one ``all_to_all`` per region sends chunk j of the gradients straight to
                        rank j over its own xGMI link (all 7 links busy at once instead of a ring hop per link), rank j
                        sums the W chunks in rank order (fp32), updates ITS chunk of the parameters (sharded AdamW) and an
                        in-place ``all_gather`` returns the updated chunks to everyone;
          "direct_bf16" the same with bf16 on the wire (half the reduce-scatter bytes), accumulation in fp32.
        Every element is reduced by exactly one rank in a fixed order and the parameters are gathered, so replicas
        are bit-identical in all modes.  ``lead``: ids of the parameters forming the leading region (the GPS backbone,
        whose gradients are final first): regions are exchanged independently, each padded to W chunks.
        ``coalesce`` (default: env RF_DP_COALESCE != "0"): neighbouring all-reduce buckets that are ready together leave
        as one collective; off = one collective per bucket."""
        import os as _os
        self.mode = mode or _os.environ.get("RF_DP_MODE", "allreduce")
        assert self.mode in ("allreduce", "direct", "direct_bf16"), self.mode
        self.coalesce = (_os.environ.get("RF_DP_COALESCE", "1") != "0") if coalesce is None else bool(coalesce)
        order = list(reversed(params))
        member = {id(p): g for g in groups for p in g}
        seen, laid = set(), []
        for p in order:  # keep reverse-registration order, but emit a whole group at its first member
            if id(p) in seen:
                continue
            for q in member.get(id(p), [p]):
                seen.add(id(q))
                laid.append(q)
        self.params = laid
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        # RF_REHEARSE_COLLECTIVES=1: run the exchange step even in a one-rank group (tools/rccl_rehearsal.py: the
        # whole N>1 code path -- RCCL launches, stream ordering, split graphs -- on a single GPU)
        import os
        self.exchange = self.world > 1 or (os.environ.get("RF_REHEARSE_COLLECTIVES") == "1" and dist.is_initialized())
        dev = self.params[0].device
        ALIGN = 64  # floats: every parameter starts on a 256-B boundary (16-B vector loads in the GEMMs)
        # layout pass: offsets, buckets, regions (direct modes: every region is padded to W equal chunks)
        sharded = self.mode != "allreduce" and self.exchange
        W = max(1, self.world)
        quantum = ALIGN * W if sharded else ALIGN
        lead = set(lead or ())
        cap = max(1, int(bucket_mb * (1 << 20) / 4))
        self.buckets: List[tuple] = []  # (start, end)
        self._bucket_of: Dict[int, int] = {}
        self.offset: Dict[int, int] = {}
        self.regions: List[tuple] = []
        off, b_start, r_start = 0, 0, 0
        for i, p in enumerate(self.params):
            if sharded and i > 0 and (id(p) in lead) != (id(self.params[i - 1]) in lead) and off > r_start:
                off = -(-off // quantum) * quantum  # region boundary: close bucket and region
                if off > b_start:
                    self.buckets.append((b_start, off))
                    b_start = off
                self.regions.append((r_start, off))
                r_start = off
            self.offset[id(p)] = off
            self._bucket_of[id(p)] = len(self.buckets)
            off += -(-p.numel() // ALIGN) * ALIGN
            if off - b_start >= cap:
                self.buckets.append((b_start, off))
                b_start = off
        total = -(-off // quantum) * quantum
        if total > b_start:
            self.buckets.append((b_start, total))
        self.regions.append((r_start, total))
        self.flat_grad = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_param = torch.zeros(total, device=dev, dtype=torch.float32)
        for p in self.params:
            o, n = self.offset[id(p)], p.numel()
            self.flat_param[o:o + n].copy_(p.detach().reshape(-1))
            p.data = self.flat_param[o:o + n].view_as(p)
            p.grad = self.flat_grad[o:o + n].view_as(p)
            p._rf_grad = p.grad  # gradient sink picked up by routeformer_amd.kernels
        self._region_done = [False] * len(self.regions)
        self._region_work: Dict[int, tuple] = {}
        self._members = [0] * len(self.buckets)
        for p in self.params:
            self._members[self._bucket_of[id(p)]] += 1
        self._pending = list(self._members)
        self._launched = [False] * len(self.buckets)
        self._works: List = []
        self._starts = sorted((self.offset[id(p)], id(p)) for p in self.params)
        self.hooks_enabled = True  # False: no in-backward launches (HIP-graph capture); finish() sends all
        # RF_DP_COMM=rf: the all-reduce mode talks to RCCL through rf_comm_* (own communicator + communication stream)
        self.rfcomm = None
        if self.exchange and not self.sharded and _os.environ.get("RF_DP_COMM", "pg") == "rf" and dev.type == "cuda":
            from routeformer_amd.comm import RfComm
            self.rfcomm = RfComm(dist.get_rank(group), self.world, group=group)
        if self.exchange and not self.sharded:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._on_grad)

    def on_sink_write(self, view):
        """A kernel accumulated straight into ``view`` (a slice of flat_grad): same bookkeeping as the
        autograd hook, for every parameter slot the view covers (packed Q/K/V views cover three)."""
        import bisect
        if not self.hooks_enabled or self.sharded:
            return
        lo = (view.data_ptr() - self.flat_grad.data_ptr()) // 4
        hi = lo + view.numel()
        i = bisect.bisect_left(self._starts, (lo, 0))
        while i < len(self._starts) and self._starts[i][0] < hi:
            b = self._bucket_of[self._starts[i][1]]
            self._pending[b] -= 1
            if self._pending[b] == 0 and not self._launched[b]:
                self._launch(b)
            i += 1

    def packed_view(self, group):
        """(param view, grad view) spanning a contiguous group, or None if padding separates its members."""
        start = self.offset[id(group[0])]
        off = start
        for p in group:
            if self.offset[id(p)] != off:
                return None
            off += p.numel()
        shape = (sum(p.shape[0] for p in group),) + tuple(group[0].shape[1:])
        return self.flat_param[start:off].view(shape), self.flat_grad[start:off].view(shape)

    # -- per-step protocol: zero() -> backward -> finish() -----------------------------------------
    def zero(self):
        self.flat_grad.zero_()
        self.begin_step()
        from routeformer_amd import kernels as K
        K.WGRAD.begin_step()

    def begin_step(self):
        """Reset the per-step bucket bookkeeping (a HIP-graph replay zeroes the buffer itself, on the device)."""
        self._pending = list(self._members)
        self._launched = [False] * len(self.buckets)
        self._works = []
        self._region_done = [False] * len(self.regions)
        self._region_work = {}

    # -- direct reduce-scatter / all-gather (modes "direct", "direct_bf16") ------------------------------------------
    @property
    def sharded(self) -> bool:
        return self.mode != "allreduce" and self.exchange

    def _rank(self) -> int:
        return dist.get_rank(self.group)

    def local_chunks(self):
        """[lo, hi) slices of the flat buffers this rank owns, one per region (the whole buffer when not sharded)."""
        if not self.sharded:
            return [(0, self.flat_param.numel())]
        r = self._rank()
        return [(lo + r * ((hi - lo) // self.world), lo + (r + 1) * ((hi - lo) // self.world)) for lo, hi in self.regions]

    def _host_staged(self, t):
        """gloo moves host memory: device tensors are staged through the host for it (tests / rehearsal only)."""
        return t.is_cuda and dist.get_backend(self.group) == "gloo"

    def _exchange_region(self, ri: int):
        """Start the direct reduce-scatter of region ``ri``: chunk j of this rank's gradients goes to rank j."""
        lo, hi = self.regions[ri]
        g = self.flat_grad[lo:hi]
        wire = g.to(torch.bfloat16) if self.mode == "direct_bf16" else g
        if self._host_staged(wire):
            send = wire.cpu()
            recv = torch.empty_like(send)
            work = dist.all_to_all_single(recv, send, group=self.group, async_op=True)
        else:
            recv = torch.empty_like(wire)
            work = dist.all_to_all_single(recv, wire, group=self.group, async_op=True)
        self._region_work[ri] = (work, recv, wire)
        self._region_done[ri] = True

    def _finish_region(self, ri: int):
        """Sum the W received chunks in rank order (fp32) into this rank's chunk of the flat gradient buffer."""
        work, recv, _ = self._region_work.pop(ri)
        work.wait()
        lo, hi = self.regions[ri]
        c = (hi - lo) // self.world
        total = recv.view(self.world, c).to(torch.float32).sum(dim=0)
        a, b = self.local_chunks()[ri]
        self.flat_grad[a:b].copy_(total, non_blocking=True)

    def gather_params(self):
        """After the sharded update: every rank's chunk of the parameters goes to all ranks (in place)."""
        if not self.sharded:
            return
        for (lo, hi), (a, b) in zip(self.regions, self.local_chunks()):
            full, mine = self.flat_param[lo:hi], self.flat_param[a:b]
            if self._host_staged(full):
                out = torch.empty(hi - lo, dtype=full.dtype)
                dist.all_gather_into_tensor(out, mine.cpu(), group=self.group)
                full.copy_(out)
            else:
                dist.all_gather_into_tensor(full, mine, group=self.group)

    def all_gather_floats(self, local: torch.Tensor) -> torch.Tensor:
        """Concatenation over ranks of a small fp32 vector (the per-chunk partial sums of the gradient norm)."""
        if self._host_staged(local):
            out = torch.empty(self.world * local.numel(), dtype=local.dtype)
            dist.all_gather_into_tensor(out, local.cpu(), group=self.group)
            return out.to(local.device)
        out = torch.empty(self.world * local.numel(), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=self.group)
        return out

    def _all_reduce(self, view):
        """SUM all-reduce of a slice of the flat gradient buffer, asynchronous, ordered after everything enqueued so far
        on the current stream: through ``rf_comm_allreduce_bucket`` (RF_DP_COMM=rf: our own RCCL communicator and
        communication stream, csrc/comm.hip) or through the process group (ProcessGroupNCCL = RCCL on its own stream)."""
        if self.rfcomm is not None:
            self.rfcomm.allreduce_bucket(view)
            self._works.append(None)
        else:
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _launch(self, b: int):
        s, e = self.buckets[b]
        self._all_reduce(self.flat_grad[s:e])
        self._launched[b] = True

    def _on_grad(self, p):
        if not self.hooks_enabled:
            return
        b = self._bucket_of[id(p)]
        self._pending[b] -= 1
        if self._pending[b] == 0 and not self._launched[b]:
            self._launch(b)

    def launch_complete_prefix(self, names_of: Dict[int, str], prefix: str) -> int:
        """All-reduce, now, every bucket made up ONLY of parameters whose name starts with ``prefix`` (their
        gradients are final).  Returns how many were launched."""
        if not self.exchange:
            return 0
        if self.sharded:  # regions are the unit of exchange: those made up only of ``prefix`` parameters go now
            n = 0
            for ri, (lo, hi) in enumerate(self.regions):
                inside = [p for p in self.params if lo <= self.offset[id(p)] < hi]
                if not self._region_done[ri] and inside and all(names_of[id(p)].startswith(prefix) for p in inside):
                    self._exchange_region(ri)
                    n += 1
            return n
        return self._launch_runs([b for b in range(len(self.buckets))
                                  if not self._launched[b] and self._bucket_prefix_ok(b, names_of, prefix)])

    def _launch_runs(self, ready: List[int]) -> int:
        """All-reduce the given buckets, every run of neighbours as ONE collective over their joint slice of the flat
        buffer: when many buckets become final at the same moment (end of a replayed graph) there is nothing to
        overlap bucket by bucket, and xGMI's point-to-point rings are per-link bound -- fewer, larger messages."""
        i, n = 0, 0
        while i < len(ready):
            j = i
            while self.coalesce and j + 1 < len(ready) and ready[j + 1] == ready[j] + 1:
                j += 1
            s, e = self.buckets[ready[i]][0], self.buckets[ready[j]][1]
            self._all_reduce(self.flat_grad[s:e])
            for b in ready[i:j + 1]:
                self._launched[b] = True
            n += j + 1 - i
            i = j + 1
        return n

    def _bucket_prefix_ok(self, b, names_of, prefix):
        key = (b, prefix)
        hit = self.__dict__.setdefault("_prefix_cache", {}).get(key)
        if hit is None:
            hit = all(names_of[id(p)].startswith(prefix) for p in self.params if self._bucket_of[id(p)] == b)
            self._prefix_cache[key] = hit
        return hit

    def finish(self):
        """Flush buckets whose parameters got no gradient this step (e.g. gaze branch dropped), then
        make the compute stream wait for all reductions.  Gradients hold the SUM over ranks; the
        1/world factor is folded into the optimizer kernel (``grad_scale``)."""
        if self.sharded:
            for ri in range(len(self.regions)):
                if not self._region_done[ri]:
                    self._exchange_region(ri)
            for ri in sorted(self._region_work):
                self._finish_region(ri)
        elif self.exchange:
            self._launch_runs([b for b in range(len(self.buckets)) if not self._launched[b]])
            for w in self._works:
                if w is not None:
                    w.wait()
            if self.rfcomm is not None:
                self.rfcomm.wait()  # the current (compute) stream waits on the device; the host does not
        return 1.0 / self.world

    def broadcast_parameters(self, src: int = 0):
        if self.exchange and self.rfcomm is not None:
            self.rfcomm.broadcast(self.flat_param, root=src)
            self.rfcomm.wait()
        elif self.exchange:
            dist.broadcast(self.flat_param, src=src, group=self.group)


class LagMap:
    """Skipped-update count per element of the flat buffers, piecewise constant: ``cuts[i] <= x < cuts[i + 1]`` has count
    ``vals[i]`` (the last piece runs to the end).  torch.optim.AdamW keeps ``state['step']`` per parameter and advances it
    only when the parameter has a gradient; a range skipped by one step (a dropped gaze branch) and another range skipped
    by another step may overlap, and their counts then ADD on the overlap (the former per-range dict took the
    maximum over enclosing ranges, right only while a single skip set exists)."""

    def __init__(self, cuts=(0,), vals=(0,)):
        self.cuts, self.vals = list(cuts), list(vals)

    def _split(self, x: int) -> int:
        import bisect
        i = bisect.bisect_right(self.cuts, x) - 1
        if i < 0:
            self.cuts.insert(0, x)
            self.vals.insert(0, 0)
            return 0
        if self.cuts[i] != x:
            self.cuts.insert(i + 1, x)
            self.vals.insert(i + 1, self.vals[i])
            return i + 1
        return i

    def add(self, lo: int, hi: int, n: int = 1):
        if hi <= lo:
            return
        i, j = self._split(int(lo)), self._split(int(hi))
        for k in range(i, j):
            self.vals[k] += n

    def inner_cuts(self, lo: int, hi: int):
        return [c for c in self.cuts if lo < c < hi]

    def lag(self, a: int, b: int) -> int:
        """The count of [a, b), which must lie inside one piece (the callers cut at ``inner_cuts`` first)."""
        import bisect
        i = bisect.bisect_right(self.cuts, a) - 1
        v = self.vals[i] if i >= 0 else 0
        j = bisect.bisect_left(self.cuts, b) - 1
        if any(self.vals[k] != v for k in range(max(i, 0), j + 1)):
            raise ValueError(f"[{a}, {b}) spans ranges with different update counts: cut the segment plan there")
        return v

    def as_dict(self):
        """{(lo, hi): count} of the pieces with a non-zero count (neighbours with equal counts merged)."""
        out, n = {}, len(self.cuts)
        k = 0
        while k < n:
            if self.vals[k] == 0 or k == n - 1:
                k += 1
                continue
            e = k
            while e + 1 < n - 1 and self.vals[e + 1] == self.vals[k]:
                e += 1
            out[(self.cuts[k], self.cuts[e + 1])] = self.vals[k]
            k = e + 1
        return out


class FusedAdamW:
    """AdamW + global-norm clipping over flat buffers; two kernel launches per step."""

    def __init__(self, flat_param: torch.Tensor, flat_grad: torch.Tensor, lr=1e-5, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=1e-4, max_grad_norm: float = 2.5):
        self.p, self.g = flat_param, flat_grad
        self.m = torch.zeros_like(flat_param)
        self.v = torch.zeros_like(flat_param)
        from routeformer_amd import _hip
        self.parts = int(_hip.lib().rf_sumsq_parts(flat_param.numel()))
        self.sumsq = torch.zeros(self.parts, device=flat_param.device, dtype=torch.float32)  # per-workgroup partials
        self._ss_plan = None  # split_sumsq: [(lo, hi, first partial, early)]
        self.betas, self.eps, self.wd, self.max_norm = betas, eps, weight_decay, max_grad_norm
        self.param_groups = [{"lr": lr}]  # what an LR scheduler drives (optimizers.LinearWarmupCosineAnnealingLR)
        self.t = 0
        self._lag = LagMap()  # per element: how many of the ``t`` updates skipped it (per-parameter step counters)

    def state_dict(self) -> dict:
        """Moments, update count AND the per-range skipped-update counts (a resumed run must not restart the
        bias corrections of a range that sat out some steps)."""
        return {"t": self.t, "m": self.m, "v": self.v, "lag_cuts": list(self._lag.cuts), "lag_vals": list(self._lag.vals),
                "lr": self.lr}

    def load_state_dict(self, sd: dict):
        self.t = int(sd["t"])
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self._lag = LagMap(sd["lag_cuts"], sd["lag_vals"])
        self.lr = float(sd["lr"])

    @property
    def lr(self) -> float:
        return self.param_groups[0]["lr"]

    @lr.setter
    def lr(self, value: float):
        self.param_groups[0]["lr"] = value

    def hyper(self, grad_scale: float, pending: bool = True, t: Optional[int] = None) -> List[float]:
        """The scalar arguments of update number ``t`` (default ``self.t``) as rf_adamw_clip_dev reads them from device
        memory."""
        t = max(self.t if t is None else t, 1)
        return [1.0 if pending else 0.0, self.max_norm, self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                1.0 - self.betas[0] ** t, (1.0 - self.betas[1] ** t) ** 0.5, grad_scale]

    def split_sumsq(self, lo: int, hi: int):
        """The partial sums of squares of the clip norm in up to three launches -- [0, lo), [lo, hi), [hi, n) -- each into its
        own run of ``self.sumsq``: the middle range ("early": the GPS backbone, 95 % of the bytes) can then be summed as soon
        as its gradients are complete, underneath the rest of the backward pass, and only the small remainder ("late") sits
        in front of the update.  The update kernel adds ALL partials in one fixed order, whichever launch wrote them."""
        from routeformer_amd import _hip
        n, o, plan = self.p.numel(), 0, []
        for a, b, early in ((0, lo, False), (lo, hi, True), (hi, n, False)):
            if b > a:
                k = int(_hip.lib().rf_sumsq_parts(b - a))
                plan.append((a, b, o, early))
                o += k
        self._ss_plan, self.parts = plan, o
        self.sumsq = torch.zeros(o, device=self.p.device, dtype=torch.float32)

    def launch_sumsq(self, which: str = "all"):
        """``which``: "all", or with ``split_sumsq`` in force "early" / "late" (see there)."""
        from routeformer_amd import _hip, kernels as K
        if self._ss_plan is None:
            assert which == "all"
            _hip.check(_hip.lib().rf_sumsq(self.g.data_ptr(), self.p.numel(), self.sumsq.data_ptr(), K._stream()), "rf_sumsq")
            return
        for a, b, o, early in self._ss_plan:
            if which == "all" or (which == "early") == early:
                _hip.check(_hip.lib().rf_sumsq(self.g.data_ptr() + 4 * a, b - a, self.sumsq.data_ptr() + 4 * o, K._stream()),
                           "rf_sumsq")

    def launch_update_dev(self, lo: int, hi: int, hyper_dev: torch.Tensor, max_blocks: int = 0):
        """AdamW over the slice [lo, hi) of the flat buffers on the current stream, scalars from ``hyper_dev``
        (the clip coefficient still comes from the norm of the WHOLE gradient buffer: launch_sumsq first).
        ``max_blocks`` > 0 throttles the launch (a side-stream update with slack)."""
        from routeformer_amd import _hip, kernels as K
        if hi <= lo:
            return
        o = 4 * lo
        ev = K.PROFILE.begin() if K.PROFILE.on else None
        args = (self.p.data_ptr() + o, self.g.data_ptr() + o, self.m.data_ptr() + o, self.v.data_ptr() + o, hi - lo,
                self.sumsq.data_ptr(), self.parts, hyper_dev.data_ptr())
        _hip.check(_hip.lib().rf_adamw_clip_dev(*args, max_blocks, K._stream()), "rf_adamw_clip_dev")
        if ev is not None:
            # algorithmic bytes: p, m, v read + written, g read (28 B per parameter).  The timing replay (bench.py, after
            # the timed region) streams the same bytes with lr = wd = 0: parameters stay, only the moments decay
            probe = hyper_dev.clone()
            probe[0], probe[2], probe[6] = 1.0, 0.0, 0.0
            K.PROFILE.end("adamw_clip_kernel<true>", ev, 12.0 * (hi - lo), 28.0 * (hi - lo),
                          replay=lambda a=args[:-1], h=probe, kp=(self.p, self.g, self.m, self.v): _hip.lib().rf_adamw_clip_dev(
                              *a, h.data_ptr(), max_blocks, K._stream()))
        K.WEIGHTS_EPOCH += 1

    def _segments(self, lo: int, hi: int, skip):
        """[(a, b, t)] covering [lo, hi) minus the ``skip`` ranges, cut at the boundaries of every range that was skipped
        in an EARLIER step: torch.optim.AdamW keeps ``state['step']`` per parameter and advances it only when the
        parameter has a gradient (full_comparison.py:694-702 + routeformer.py:299-310: a dropped gaze branch leaves
        ``.grad`` None), so the bias corrections of such a range use its own update count ``t - lag``."""
        cuts = sorted(set([lo, hi] + [x for r in skip for x in r if lo < x < hi] + self._lag.inner_cuts(lo, hi)))
        out = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            if any(x <= a and b <= y for x, y in skip):
                continue
            out.append((a, b, self.t - self._lag.lag(a, b)))
        return out

    def _note_skipped(self, skip):
        for r in skip:
            self._lag.add(int(r[0]), int(r[1]))

    def step_sharded(self, reducer: "GradReducer", grad_scale: float, skip=()):
        """Modes "direct" / "direct_bf16": this rank holds the rank-summed gradients of ITS chunk of every region; it
        clips with the GLOBAL norm (per-chunk partial sums of squares, all-gathered: every rank adds the same W x P
        partials in the same order, so the coefficient is bit-identical everywhere), updates its chunks, and the
        in-place all-gather of the parameters follows (``reducer.gather_params``)."""
        from routeformer_amd import _hip, kernels as K
        self.t += 1
        self._note_skipped(skip)
        chunks = reducer.local_chunks()
        parts = [int(_hip.lib().rf_sumsq_parts(b - a)) for a, b in chunks]
        local = torch.zeros(sum(parts), device=self.p.device, dtype=torch.float32)
        o = 0
        for (a, b), n_p in zip(chunks, parts):
            _hip.check(_hip.lib().rf_sumsq(self.g.data_ptr() + 4 * a, b - a, local.data_ptr() + 4 * o, K._stream()), "rf_sumsq")
            o += n_p
        everyone = reducer.all_gather_floats(local)
        for a, b in chunks:
            for sa, sb, t in self._segments(a, b, skip):
                q = 4 * sa
                _hip.check(_hip.lib().rf_adamw_clip(self.p.data_ptr() + q, self.g.data_ptr() + q, self.m.data_ptr() + q,
                                                    self.v.data_ptr() + q, sb - sa, everyone.data_ptr(), everyone.numel(),
                                                    self.max_norm, self.lr, self.betas[0], self.betas[1], self.eps,
                                                    self.wd, t, grad_scale, K._stream()), "rf_adamw_clip")
        reducer.gather_params()
        K.WEIGHTS_EPOCH += 1

    def step(self, grad_scale: float = 1.0, skip=()):
        """``skip``: sorted, disjoint [lo, hi) ranges of the flat buffers that took no part in this step (their
        ``.grad`` would be None in the reference, whose AdamW then leaves the parameter, its moments AND its step
        counter untouched -- no weight decay either; they contribute nothing to the clip norm: their gradient slots
        are zero)."""
        from routeformer_amd import _hip, kernels as K
        self.t += 1
        self._note_skipped(skip)
        n = self.p.numel()
        self.launch_sumsq()
        for a, b, t in self._segments(0, n, skip):
            o = 4 * a
            _hip.check(_hip.lib().rf_adamw_clip(self.p.data_ptr() + o, self.g.data_ptr() + o, self.m.data_ptr() + o,
                                                self.v.data_ptr() + o, b - a, self.sumsq.data_ptr(), self.parts,
                                                self.max_norm, self.lr, self.betas[0], self.betas[1], self.eps,
                                                self.wd, t, grad_scale, K._stream()), "rf_adamw_clip")
        K.WEIGHTS_EPOCH += 1  # parameters were rewritten in place: cached bf16 copies are stale


class TrainEngine:
    """One full train step = forward(input) + target-feature forward + losses + backward (+ overlapped
    gradient all-reduce) + clip + AdamW -- the unit ``bench.py`` times."""

    def __init__(self, model, lr=None, weight_decay=None, max_grad_norm: float = 2.5, bucket_mb: float = 32.0,
                 overlap: bool = True):
        self.model = model
        self.overlap = overlap  # independent sub-graphs of the step on separate HIP streams
        # weight gradients queued during backward and flushed as grouped launches (kernels._WgradQueue)
        self.group_wgrad = __import__("os").environ.get("RF_GROUP_WGRAD", "1") != "0"
        from routeformer_amd import kernels as K
        # side streams, open-fork bookkeeping and the deferred weight-gradient queue belong to THIS engine: nothing of
        # another engine's capture (streams, queued operands, "slot written" notes) can reach into this one's
        self._streams = K.SideStreams()
        self._wgrad = K._WgradQueue()
        self._saved_scope = None
        self._early_sumsq = False  # (GraphedTrainEngine with a deferred update: see _backward_gps_first)
        cfg = model.configs
        layers = [m for m in model.modules() if hasattr(m, "packing_groups")]
        groups = [g for m in layers for g in m.packing_groups()]
        lead = {id(p) for n, p in model.named_parameters() if n.startswith("gps_backbone.")}
        self.reducer = GradReducer(trainable_parameters(model), bucket_mb, groups=groups, lead=lead)
        # Kernels write parameter gradients through sinks and through deferred grouped launches, and autograd's
        # post-accumulate hooks also fire for parameters whose Function returned None -- "this slot is final" is
        # only known at the end of backward (or at the stage boundary of the two-graph step), so buckets are
        # reduced there, not from per-parameter hooks.  (A 2-rank GPU test caught hook-driven launches going out
        # before queued weight gradients had been written.)
        self.reducer.hooks_enabled = False
        for m in layers:  # hand each attention layer its packed [Wq;Wk;Wv] / [bq;bk;bv] views
            gw, gb = m.packing_groups()
            vw, vb = self.reducer.packed_view(gw), self.reducer.packed_view(gb)
            if vw is not None and vb is not None:
                m._packed = {"w": vw[0], "gw": vw[1], "b": vb[0], "gb": vb[1]}
        self.reducer.broadcast_parameters(0)
        self.opt = FusedAdamW(self.reducer.flat_param, self.reducer.flat_grad,
                              lr=cfg.lr if lr is None else lr,
                              weight_decay=cfg.wd if weight_decay is None else weight_decay,
                              max_grad_norm=max_grad_norm)
        self.tl = FutureDiscountedLoss(cfg.discount_factor, cfg.epsilon, loss_function="smooth_l1")
        self.dl = FutureDiscountedLoss(cfg.discount_factor, cfg.visual_epsilon, loss_function="smooth_l1")
        # nn.Dropout anywhere on the trainable path: the device-side mask generator advances once per step
        self._device_dropout = (getattr(cfg, "feature_dropout", 0.0) > 0
                                or getattr(cfg.gps_backbone_config, "dropout", 0.0) > 0)
        self._names = {id(p): n for n, p in model.named_parameters()}
        # ranks make their own view / gaze dropout draws (the reference seeds nothing per rank): which optimizer slots
        # a step may skip is agreed over a host-side group (agree_unused)
        self._flag_group = None
        if self.reducer.exchange and self.reducer.world > 1 and getattr(cfg, "gaze_dropout", 0.0) > 0:
            self._flag_group = dist.new_group(backend="gloo")
        self._skip_cache = {}

    def _begin_step_kernels(self):
        """First launches of a step: zero the flat gradient buffer (GraphedTrainEngine may add the deferred update)."""
        self._advance_rng()
        self._repack_fused()
        self.reducer.zero()

    def _repack_fused(self):
        """bf16 fragment copies of the fused encoder stacks' weights (blocks.FusedStack): re-packed from the fp32
        masters at the head of every step -- one launch per stack, part of the captured graph, before the
        target-side pass forks off."""
        if not self.reducer.flat_param.is_cuda:
            return
        from routeformer_amd import kernels as K
        stacks = self.__dict__.get("_fused_stacks")
        if stacks is None:
            stacks = self._fused_stacks = [st for m in self.model.modules() if hasattr(m, "fused_stack")
                                           for st in [m.fused_stack()] if st is not None]
        if K.SEQSTACK and K.get_precision() == "bf16" and stacks:
            plan = self.__dict__.get("_pack_plan")
            if plan is None or plan[0] != K.SEQSTACK_BWD:
                ents = []
                for st in stacks:
                    st.refresh(force=True, collect=ents)  # (allocates the blobs, keys them; the launch is the plan's)
                plan = self._pack_plan = (K.SEQSTACK_BWD, K.PackPlan(ents, self.reducer.flat_param.device))
            for st in stacks:
                st._key = None  # the blobs are refreshed by the plan, not by FusedStack.refresh: never trust its key
            plan[1].launch()

    def _advance_rng(self):
        if self._device_dropout and self.reducer.flat_param.is_cuda:
            from routeformer_amd import kernels as K
            K.RNG.begin_step(self.reducer.flat_param.device)  # step += 1 on the device (captured: once per replay)

    def _skip_ranges(self, prefixes):
        """[lo, hi) slices of the flat buffers holding the parameters whose names start with one of ``prefixes``
        (merged, sorted): the slots a step did not touch (gaze branch dropped, routeformer.py:299-310)."""
        if not prefixes or not agree_unused(True, self._flag_group):
            if not prefixes and self._flag_group is not None:
                agree_unused(False, self._flag_group)  # every rank takes part in the exchange every step
            return ()
        return self._prefix_ranges(prefixes)

    def _prefix_ranges(self, prefixes):
        """The merged [lo, hi) slices of the parameters whose names start with one of ``prefixes`` (no rank agreement)."""
        hit = self._skip_cache.get(tuple(prefixes))
        if hit is not None:
            return hit
        r = self.reducer
        spans = sorted((r.offset[id(p)], r.offset[id(p)] + p.numel()) for p in r.params
                       if self._names[id(p)].startswith(tuple(prefixes)))
        merged = []
        for a, b in spans:
            a_al = a  # slots are padded to 64 floats: padding between two skipped neighbours is skipped too
            if merged and a_al - merged[-1][1] < 64:
                merged[-1][1] = b
            else:
                merged.append([a_al, b])
        self._skip_cache[tuple(prefixes)] = tuple((a, b) for a, b in merged)
        return self._skip_cache[tuple(prefixes)]

    def _loss_seed(self, loss):
        """d loss / d loss = 1 as a persistent tensor (``backward()`` would fill a fresh one: a launch between the forward and
        the backward pass, where nothing overlaps it)."""
        seed = self.__dict__.get("_seed_one")
        if seed is None or seed.device != loss.device or seed.dtype != loss.dtype or seed.shape != loss.shape:
            seed = self._seed_one = torch.ones_like(loss)
        return seed

    def _scope_in(self):
        from routeformer_amd import kernels as K
        self._saved_scope = (K.STREAMS, K.WGRAD)
        K.STREAMS, K.WGRAD = self._streams, self._wgrad

    def _scope_out(self):
        from routeformer_amd import kernels as K
        if self._saved_scope is not None:
            K.STREAMS, K.WGRAD = self._saved_scope
            self._saved_scope = None

    def _fwd_bwd(self, item, epoch, tokens_ready: bool = False):
        from routeformer_amd import kernels as K
        self._scope_in()
        K.OVERLAP = self.overlap
        K.SINK.active = True  # kernels accumulate parameter gradients straight into the flat buffer
        K.SINK.on_write = None
        K.WGRAD.active = self.group_wgrad
        K.WGRAD.side_early = self.overlap and WGRAD_SIDE  # (single-graph / eager path only: no collective waits on these)
        try:
            self._begin_step_kernels()
            self.model._keep_gps_input = self._early_sumsq
            try:
                res = train_step_losses(self.model, item, epoch, self.tl, self.dl, tokens_ready=tokens_ready)
            finally:
                self.model._keep_gps_input = False
            cut = self.model.__dict__.pop("_gps_input", None)
            if self._early_sumsq:
                self._backward_gps_first(res["loss"], cut)
            else:
                res["loss"].backward(gradient=self._loss_seed(res["loss"]))
            K.flush_weight_grads()  # queued dW / db launches, each on the stream its operands were produced on
            if self.overlap:
                K.join_side_streams()
            self._check_lazy()
        finally:
            K.LAZY.clear()
            K.SINK.active, K.SINK.on_write, K.OVERLAP, K.WGRAD.active = False, None, False, False
            K.WGRAD.side_early = False
            self._scope_out()
            self.model.__dict__.pop("_before_gps_backbone", None)
            K.STEP_HOOKS.pop("after_frame_embedding", None)
        return res

    @staticmethod
    def _check_lazy():
        """Every gradient that left its producer as split-K slabs (kernels.LAZY) must have met a consumer that sums them."""
        from routeformer_amd import kernels as K
        if K.LAZY:
            n = len(K.LAZY)
            K.LAZY.clear()
            raise RuntimeError(f"{n} slab-carried gradient(s) were never consumed: a placeholder tensor reached an autograd node "
                               "that does not look kernels.LAZY up (ffn_add_layer_norm(sole_consumer=True) on a shared tensor?)")

    def _backward_gps_first(self, loss, cut):
        """Backward pass with the clip norm's big term taken early (single process, deferred update): the GPS backbone --
        95 % of the trainable bytes, differentiated FIRST -- down to its input ``cut``, its queued weight gradients flushed
        (side stream), then the sum of squares of ITS range of the gradient buffer on that side stream, underneath the rest
        of the backward pass.  The next step's update then waits for a sum over the remaining 5 % only (``FusedAdamW.
        split_sumsq``; the whole-buffer pass was 68 us of HBM time at the head of every step, in front of the camera /
        gaze / fusion encoders' update)."""
        from routeformer_amd import kernels as K
        seed = self._loss_seed(loss)
        if cut is None or not cut.requires_grad:  # GPS-only model: nothing upstream of the backbone
            loss.backward(gradient=seed)
            K.flush_weight_grads()
            self.opt.launch_sumsq("early")
            return
        bp = self.__dict__.get("_gps_params")
        if bp is None:
            bp = self._gps_params = [p for n, p in self.model.named_parameters() if n.startswith("gps_backbone.") and p.requires_grad]
        grads = torch.autograd.grad(loss, [cut] + bp, grad_outputs=seed, allow_unused=True)
        for p, g in zip(bp, grads[1:]):  # (the few backbone parameters no kernel writes through a sink)
            if g is not None:
                p.grad.add_(g)
        cur = torch.cuda.current_stream()

        def early(st):  # st: the stream the last of the backbone's weight-gradient groups was launched on
            if self.overlap and st.cuda_stream == cur.cuda_stream:
                st = K.fork_side_stream("wgrad", origin=st)
            # the backbone's gradient sinks were written on the main stream and on every branch of this step: order the sum
            # behind all of them (a replayed graph starts a node as soon as its captured dependencies allow)
            if st.cuda_stream != cur.cuda_stream:
                st.wait_stream(cur)
            for other in list(K.STREAMS.forked.values()):
                if other.cuda_stream != st.cuda_stream:
                    st.wait_stream(other)
            with torch.cuda.stream(st):
                self.opt.launch_sumsq("early")

        if EARLY_SUMSQ_FLUSH:
            K.WGRAD.flush(side=K.WGRAD.side_early)
        slots = self.__dict__.get("_gps_slots")
        if slots is None:
            slots = self._gps_slots = {p._rf_grad.data_ptr() for p in bp if getattr(p, "_rf_grad", None) is not None}
        K.WGRAD.when_launched(slots, early)
        cut.backward(grads[0])
        if K.WGRAD._watch is not None:  # (still queued: the final flush launches them -- and calls ``early``)
            K.flush_weight_grads()

    # -- the same step in two stages (GraphedTrainEngine with N > 1): stage 1 = forward + the backward of the
    #    GPS backbone, which owns 95 % of the trainable bytes and is differentiated FIRST; stage 2 = the rest.
    #    The all-reduce of the backbone's buckets then runs underneath stage 2. ---------------------------------
    def _enter(self):
        from routeformer_amd import kernels as K
        self._scope_in()
        K.OVERLAP = self.overlap
        K.SINK.active = True
        K.SINK.on_write = None
        K.WGRAD.active = self.group_wgrad

    def _leave(self):
        from routeformer_amd import kernels as K
        K.SINK.active, K.SINK.on_write, K.OVERLAP, K.WGRAD.active = False, None, False, False
        self._scope_out()
        self.model.__dict__.pop("_before_gps_backbone", None)
        K.STEP_HOOKS.pop("after_frame_embedding", None)

    def _stage1(self, item, epoch, tokens_ready: bool = False):
        from routeformer_amd import kernels as K
        self._begin_step_kernels()
        self.model._keep_gps_input = True
        try:
            res = train_step_losses(self.model, item, epoch, self.tl, self.dl, tokens_ready=tokens_ready)
        finally:
            self.model._keep_gps_input = False
        cut = self.model.__dict__.pop("_gps_input", None)
        if cut is None or not cut.requires_grad:  # GPS-only model: nothing upstream of the backbone
            res["loss"].backward()
            K.flush_weight_grads()
            if self.overlap:
                K.join_side_streams()
            if self._early_sumsq:
                self.opt.launch_sumsq("early")
            return res, None
        # gradients of the cut AND of the backbone parameters that autograd itself accumulates (the few that no
        # kernel writes through a sink, e.g. the time-feature embedding): they would otherwise be skipped
        bp = [p for n, p in self.model.named_parameters() if n.startswith("gps_backbone.") and p.requires_grad]
        grads = torch.autograd.grad(res["loss"], [cut] + bp, allow_unused=True)
        for p, g in zip(bp, grads[1:]):
            if g is not None:
                p.grad.add_(g)
        K.flush_weight_grads()
        forked = {}
        self._check_lazy()  # (the slab-carried gradients all live inside the backbone)
        if self.overlap:
            forked = K.STREAMS.join()  # (the backbone's own side branch: stage 1 may be the end of a captured graph)
        if self._early_sumsq:  # (one process rehearsing the two-graph step: the backbone's gradients are complete here)
            self.opt.launch_sumsq("early")
        return res, (cut, grads[0], forked)

    def _stage2(self, carry):
        from routeformer_amd import kernels as K
        if carry is not None:
            cut, dcut, forked = carry
            # the rest of backward runs on the streams stage 1 forked (autograd replays each node on its forward's stream)
            # and writes gradient sinks there: the final join has to cover them again
            K.STREAMS.forked.update(forked)
            cut.backward(dcut)
            K.flush_weight_grads()
        if self.overlap:
            K.join_side_streams()

    def step(self, item, epoch: int = 0, next_item=None):
        self.model.train()
        res = self._fwd_bwd(item, epoch)
        scale = self.reducer.finish()
        self._update(scale, self._skip_ranges(self.model.__dict__.get("_unused_prefixes", ())))
        return res

    def _update(self, scale: float, skip=()):
        """Clip + AdamW: every rank on the whole buffer, or (direct modes) on its own chunks + parameter all-gather."""
        if self.reducer.sharded:
            self.opt.step_sharded(self.reducer, scale, skip)
        else:
            self.opt.step(scale, skip=skip)


class GraphedTrainEngine(TrainEngine):
    """The same step with forward + backward captured once in a HIP graph and replayed.

    A step of this model is ~2-4k small launches; replaying them from a graph removes the Python /
    launch-path cost between kernels.  What stays outside the graph: (i) the host-RNG draws of the
    ProbSparse key samples -- made before every replay in the reference's order, one async copy
    (``IndexSampler`` static mode); (ii) for N > 1 the bucketed gradient all-reduce, issued back to back
    on the communicator's stream right after the replay (the replayed backward cannot call hooks; at
    301 MB over 7 xGMI links this is a few % of the step); (iii) the clip + AdamW launches, whose scalar
    arguments (bias correction) change every step.

    Software pipelining across steps: the frozen conv trunk has no gradient and does not depend on the
    weights being trained, so ``step(item, next_item=...)`` runs the trunk pass of the NEXT batch (its own
    small graph, on a side stream) underneath this batch's transformer forward / backward / optimizer --
    large, chip-filling conv launches next to latency-bound small ones.  Every step still executes exactly
    one trunk pass and one full forward/backward/update.

    Requires a step whose control flow does not depend on random draws (view / gaze dropout 0) and fixed
    batch shapes."""

    def __init__(self, model, defer_update: bool = False, **kw):
        super().__init__(model, **kw)
        # defer_update: the clip + AdamW of step k is replayed at the START of step k+1's graph, and the 95 % of it
        # that belongs to the GPS backbone runs on a side stream underneath step k+1's camera / gaze / fusion encoders
        # (which do not read those parameters); the backbone's forward waits for it.  Same arithmetic in the same
        # order -- the bandwidth-bound update just no longer sits alone on the critical path.  ``flush()`` applies a
        # still-pending update (call it before reading parameters or saving a checkpoint).
        self.defer_update = defer_update
        self._pending = None          # per-segment scalars of the update the next replay has to apply
        self._segments = None         # [(lo, hi, on the side stream)]: fixed launch plan of a deferred update
        self._hyper = self._hyper_pinned = None
        self._gps_range = None
        self.graph = None
        self._static_item = None
        self._out = None
        self._trunk_g = None
        self._graphs = {}
        self._pool = None
        self._ready_id = None
        self._ready_checked = None
        self._cached_ids = {}     # batch id -> (cache slots, content keys) of its frames
        self._uncached_ids = set()  # ids whose frames did not fit into the cache
        self._id_bad = self._id_bad_host = self._id_checked = None
        self._clip_stage = self._clip_idx = None
        self._recipe = None
        self._tstream = None
        # N > 1: capture the step as two graphs so that the gradient all-reduce of the GPS backbone overlaps the
        # rest of the backward pass (RF_SPLIT_BWD=1 forces it at N = 1, =0 disables it)
        env = __import__("os").environ.get("RF_SPLIT_BWD")
        self.split = self.reducer.exchange if env is None else env == "1"
        self._names = {id(p): n for n, p in model.named_parameters()}
        c = model.configs
        if c.motion_noise > 0:
            raise ValueError("GraphedTrainEngine: motion_noise > 0 (torch.randn_like on the inputs) is not supported")
        if defer_update and self.reducer.sharded:
            raise ValueError("defer_update replays the whole-buffer update; use RF_DP_MODE=allreduce with it")

    def _eager_fwd_bwd(self, item, epoch):
        return self._fwd_bwd(item, epoch)

    # -- deferred update ---------------------------------------------------------------------------
    def _backbone_range(self):
        """[lo, hi) of the flat buffers holding exactly the GPS backbone's parameters, or None."""
        r = self.reducer
        mine = [p for p in r.params if self._names[id(p)].startswith("gps_backbone.")]
        if not mine:
            return None
        lo = min(r.offset[id(p)] for p in mine)
        hi = max(r.offset[id(p)] + p.numel() for p in mine)
        inside = [p for p in r.params if lo <= r.offset[id(p)] < hi]
        if len(inside) != len(mine):
            return None
        hi = min([r.offset[id(p)] for p in r.params if r.offset[id(p)] >= hi] + [r.flat_param.numel()])  # + padding
        return lo, hi

    def _begin_step_kernels(self):
        """First launches of a step.  Plain engine: zero the gradient buffer.  Deferred update: the pending clip +
        AdamW (scalars from ``self._hyper``, a no-op while nothing is pending), the backbone's share on a side stream
        that the backbone forward joins (``model._before_gps_backbone``), each slice zeroed once it has been used."""
        from routeformer_amd import kernels as K
        self._advance_rng()
        if not self.defer_update:
            self._repack_fused()
            self.reducer.zero()
            return
        r, opt = self.reducer, self.opt
        opt.launch_sumsq("late" if self._early_sumsq else "all")  # ("early" = the backbone's range: _backward_gps_first)
        use_side = self._gps_range is not None and self.overlap and __import__("os").environ.get("RF_DEFER_SIDE", "1") != "0"
        cur = torch.cuda.current_stream()

        def run(segments):  # one launch per segment (its own scalars), one zero per contiguous run of segments
            runs = []
            for i, (a, b, _) in segments:
                opt.launch_update_dev(a, b, self._hyper[i])
                if runs and runs[-1][1] == a:
                    runs[-1][1] = b
                else:
                    runs.append([a, b])
            for a, b in runs:
                r.flat_grad[a:b].zero_()

        seg = list(enumerate(self._segments))
        # the encoders' 5 % first and ALONE, then the fork: launched next to the backbone's 2-GB streaming update the
        # small slices were starved by it (rocprofv3 trace, profiles/r03/step_trace_before.txt: 394 us for 98 MB of
        # traffic, both launches ending together) -- and the camera / gaze / fusion encoders wait for exactly them
        run([x for x in seg if not (use_side and x[1][2])])
        # the encoders' slots are final: pack their bf16 fragments BEFORE the streaming update is let loose (next to it
        # the 16-us pack kernel ran for as long as the update did -- kernel trace, profiles/r03/overlap_experiments.txt)
        self._repack_fused()
        if use_side:
            box = {}

            def launch():  # fork HERE (wherever the caller is on the main stream) and stream the backbone's update
                if "side" not in box:
                    box["side"] = K.fork_side_stream("update", origin=torch.cuda.current_stream())
                    with torch.cuda.stream(box["side"]):
                        run([x for x in seg if x[1][2]])

            def join():
                launch()  # (a model without a camera-token embedding never reached the hook below)
                K.STEP_HOOKS.pop("after_frame_embedding", None)
                torch.cuda.current_stream().wait_stream(box["side"])

            if UPDATE_AFTER_EMBED and getattr(self.model, "with_video", False):
                K.STEP_HOOKS["after_frame_embedding"] = launch  # (measurement switch, see UPDATE_AFTER_EMBED)
            else:
                launch()
            self.model.__dict__["_before_gps_backbone"] = join
        r.begin_step()
        K.WGRAD.begin_step()

    def _plan_segments(self):
        """The fixed segmentation of the flat buffers a deferred update is launched in: cut at the GPS backbone's range
        (its share runs on the side stream) and at every range a step may leave untouched (a dropped gaze branch:
        routeformer.py:299-310) -- such a segment is switched off for that update by its own "pending" flag on the
        device, and carries its own update count (torch.optim.AdamW's per-parameter ``state['step']``)."""
        n = self.reducer.flat_param.numel()
        cuts = {0, n}
        rng = self._gps_range
        if rng is not None:
            cuts.update(rng)
        if self.model.configs.gaze_dropout > 0 and getattr(self.model, "with_gaze", False):
            for a, b in self._prefix_ranges(("gaze_encoder.", "gaze_video_decoder.")):
                cuts.update((a, b))
        cuts = sorted(cuts)
        return [(a, b, rng is not None and rng[0] <= a and b <= rng[1]) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]

    def _pending_rows(self, scale: float, skip):
        """Per segment the scalars of update ``opt.t`` (call after ``opt.t += 1`` and ``opt._note_skipped(skip)``)."""
        opt, rows = self.opt, []
        for a, b, _ in self._segments:
            skipped = any(x <= a and b <= y for x, y in skip)
            lag = opt._lag.lag(a, b)
            rows.append(opt.hyper(scale, pending=not skipped, t=opt.t - lag))
        return rows

    def _set_hyper(self):
        """Ship the pending update's scalars (or "nothing pending") ahead of the replay."""
        rows = self._pending if self._pending is not None else [self.opt.hyper(1.0, pending=False)] * len(self._segments)
        self._ship_hyper(rows)

    def _ship_hyper(self, rows):
        ev = self.__dict__.get("_hyper_copied")
        if ev is not None:
            ev.synchronize()  # the previous copy must have left the pinned buffer before it is rewritten
        vals = torch.tensor(rows, dtype=torch.float32)
        self._hyper_pinned[:, :vals.shape[1]].copy_(vals)
        self._hyper.copy_(self._hyper_pinned, non_blocking=True)
        self._hyper_copied = torch.cuda.Event()
        self._hyper_copied.record()

    def flush(self):
        """Apply a still-pending optimizer update now (eagerly, on the current stream)."""
        if self._pending is not None:
            rows, self._pending = self._pending, None
            self.opt.launch_sumsq()
            self._ship_hyper(rows)
            for i, (a, b, _) in enumerate(self._segments):
                self.opt.launch_update_dev(a, b, self._hyper[i])
            self.reducer.flat_grad.zero_()
        return self

    # -- conv-trunk side ---------------------------------------------------------------------------
    # The trunk graphs never read a caller's tensors: the frames a pass needs (the sub-sampled ones, 8 of 40 per
    # clip) are gathered into staging buffers the engine owns, and every graph is captured on those.  A loader
    # may therefore hand over freshly allocated device tensors every step, or refill its own buffers as soon as
    # ``step`` returns: the number of captured graphs stays at three (trunk alone, step, step + look-ahead trunk)
    # and nothing is keyed by ``data_ptr``.  The copy is ~100 MB per step at C2 (0.03 ms).
    def _alloc_clip_stage(self, item):
        clips, _ = self.model.video_clips([item["train"], item["target"]])
        self._clip_stage = [torch.empty((v.shape[0], idx.numel()) + tuple(v.shape[2:]), device=v.device, dtype=v.dtype)
                            for v, idx in clips]
        self._clip_idx = [idx.to(v.device) for v, idx in clips]

    def _stage_clips(self, item):
        """Gather the frames of ``item`` the trunk reads into the engine's staging buffers (current stream)."""
        clips, _ = self.model.video_clips([item["train"], item["target"]])
        assert len(clips) == len(self._clip_stage), "the batch has a different set of camera streams than the captured one"
        from routeformer_amd import _hip
        todo = []
        for (v, _), dst, idx in zip(clips, self._clip_stage, self._clip_idx):
            if v.dtype != dst.dtype or v.shape[0] != dst.shape[0] or v.shape[2:] != dst.shape[2:]:
                raise ValueError(f"clip {tuple(v.shape)} {v.dtype} does not match the captured step "
                                 f"({tuple(dst.shape)} {dst.dtype}): capture() again for a new batch shape")
            if v.is_cuda and v.is_contiguous() and dst.is_contiguous() and idx.dtype == torch.int64:
                todo.append((v, dst, idx))
            else:
                torch.index_select(v, 1, idx, out=dst)
        for s0 in range(0, len(todo), _hip.GATHER_MAX):  # one launch for all camera streams (was six index_select launches)
            chunk = todo[s0:s0 + _hip.GATHER_MAX]
            arr = (_hip.GatherEntry * len(chunk))()
            for e, (v, dst, idx) in zip(arr, chunk):
                e.src, e.dst, e.idx = v.data_ptr(), dst.data_ptr(), idx.data_ptr()
                e.B, e.T, e.F, e.pad = v.shape[0], v.shape[1], dst.shape[1], 0
                e.frame_bytes = v[0, 0].numel() * v.element_size()
            _hip.check(_hip.lib().rf_gather_frames(arr, len(chunk), torch.cuda.current_stream().cuda_stream), "rf_gather_frames")

    def _staged(self):
        return [(t, None) for t in self._clip_stage]  # frame index None = every staged frame

    def _encode(self, clips, out=None):
        """The trunk itself (never the token cache: the engine consults that per batch id, outside its graphs)."""
        vb = self.model.video_backbone
        return getattr(vb, "_encode_clips_uncached", vb.encode_clips)(clips, out=out)

    # -- backbone-feature cache (token_cache.TokenCache on the video backbone) --------------------------------------------
    def _cache(self):
        return getattr(self.model.video_backbone, "token_cache", None) if self._pipelined else None

    def _cached_tokens_into_next(self, item) -> bool:
        """Tokens of a batch the cache already holds (known by its explicit id) -> ``_tok_next``; no trunk pass, no
        host synchronisation.  An id is a promise that the frames are the ones seen under it before; the promise is
        CHECKED: the incoming frames are hashed again on the device and compared with the keys remembered for the id,
        the verdict is read back asynchronously and a reused id raises at the start of the next step (``verify_ids()``
        forces the check)."""
        cache, iid = self._cache(), item.get("id")
        hit = self._cached_ids.get(iid) if (cache is not None and iid is not None) else None
        if hit is None:
            return False
        if not self._check_id_content(item):
            return False  # cannot be hashed in place: take the trunk path
        cache.gather(hit[0], self._tok_next)
        return True

    @staticmethod
    def _clip_ptrs(item):
        # addresses AND version counters: a loader that refills fixed staging buffers in place keeps the addresses; the
        # in-place write bumps ``_version`` -- a changed version forces the device-side content hash again
        return tuple((v.data_ptr(), v._version) for part in ("train", "target") for v in item[part].values() if v.dim() == 5)

    def _check_id_content(self, item) -> bool:
        """Device-side check that the frames of ``item`` are the ones remembered under its id (no host synchronisation;
        ``verify_ids`` reads the verdict).  False when the id is unknown or the clips cannot be hashed in place."""
        cache, iid = self._cache(), item.get("id")
        hit = self._cached_ids.get(iid) if (cache is not None and iid is not None) else None
        if hit is None:
            return False
        clips, _ = self.model.video_clips([item["train"], item["target"]])
        if not all(v.is_cuda and v.is_contiguous() for v, _ in clips):
            return False
        keys = hit[1]
        if self._id_bad is None:
            self._id_bad = torch.zeros(1, dtype=torch.bool, device=keys.device)
            self._id_bad_host = torch.zeros(1, dtype=torch.bool).pin_memory()
        self._id_bad |= (cache.keys_of(clips) != keys).any()
        self._id_bad_host.copy_(self._id_bad, non_blocking=True)
        self._id_checked = (torch.cuda.Event(), iid)
        self._id_checked[0].record()
        return True

    def verify_ids(self, block: bool = True):
        """Raise if a batch id was reused for different frames (see ``_cached_tokens_into_next``).  ``block=False``: only
        look at a verdict that has already arrived."""
        pend = self.__dict__.get("_id_checked")
        if pend is None:
            return
        if block:
            pend[0].synchronize()
        elif not pend[0].query():
            return
        self._id_checked = None
        if bool(self._id_bad_host[0]):
            self._cached_ids.clear()
            self._id_bad.zero_()
            raise RuntimeError(f"GraphedTrainEngine: batch id {pend[1]!r} (or an earlier one) came with frames that differ from "
                               "the ones cached under it -- an id must identify the frame content for as long as the "
                               "engine lives (forget_ids() after changing the loader); the affected step trained on "
                               "the cached tokens")

    def forget_ids(self):
        """Drop the id -> cache-slot memory (call when the loader / dataset / id scheme changes)."""
        self._cached_ids.clear()
        self._uncached_ids.clear()
        self._ready_id = None

    def _remember_tokens(self, item, keys):
        """After a trunk pass over the staged frames of ``item``: store its tokens (content-keyed), and remember the
        slots AND keys under the batch id (one device synchronisation, the first time a batch is seen; an id whose
        frames did not fit is remembered too, so a cache smaller than the dataset costs no further synchronisations)."""
        cache, iid = self._cache(), item.get("id")
        if cache is None or keys is None or iid is None or iid in self._uncached_ids:
            return
        slots, _ = cache.lookup(keys, count_misses=False)
        final = cache.insert(keys, slots, self._tok_next)
        if bool((final >= 0).all()):
            self._cached_ids[iid] = (final, keys)
        else:
            if not self._uncached_ids:
                import warnings
                warnings.warn("TokenCache is full: batches that do not fit keep running the conv trunk every step")
            self._uncached_ids.add(iid)

    def _drop_graphs(self, keep_trunk: bool = False):
        """Forget the captured graphs (never destroyed: ``_retire``)."""
        dev = self.reducer.flat_param.device
        for g, _ in self.__dict__.get("_graphs", {}).values():
            _retire(g, dev)
        self._graphs = {}
        self.graph = self._out = None
        if not keep_trunk:
            _retire(self._trunk_g, dev)
            self._trunk_g = None

    def __del__(self):
        try:
            if self.reducer.flat_param.is_cuda:
                self._drop_graphs()
        except Exception:
            pass

    def _trunk_graph(self):
        """Graph of one trunk pass over the staged frames into ``self._tok_next`` (cold start / no look-ahead)."""
        if self._trunk_g is None:
            self._encode(self._staged(), out=self._tok_next)  # warm (weight folding, caches)
            torch.cuda.synchronize()
            g = _new_graph()
            with torch.cuda.graph(g, pool=self._pool, **_capture_kw()):
                self._encode(self._staged(), out=self._tok_next)
            self._trunk_g = g
        return self._trunk_g

    def _recipe_for(self, epoch: int):
        """What the captured loss arithmetic depends on besides the tensors: the dense-loss switch (epoch >= 10,
        full_comparison.py:500-505) and the discount in force at ``epoch`` (losses/future_discounted_mse.py:71-74;
        rf_traj_head takes gamma by value, so it is baked into the graph)."""
        self.tl.current_epoch = self.dl.current_epoch = epoch
        return (epoch >= 10, float(self.tl.discount()), float(self.dl.discount()))

    def capture(self, item, epoch: int = 0, warmup: int = 2):
        from routeformer_amd import kernels as K
        from routeformer_amd.models.blocks import SAMPLER
        self.reducer.hooks_enabled = False  # no collective inside the captured region
        self.model.train()
        SAMPLER.drop_static()  # a previous capture in this process (another engine / shape) planned its own draws
        dev = self.reducer.flat_param.device
        if self.defer_update:
            self.flush()
            from routeformer_amd._hip import lib as _lib  # noqa: F401  (fail loudly without the extension)
            self._gps_range = self._backbone_range()
            self._segments = self._plan_segments()
            self._early_sumsq = (EARLY_SUMSQ and self._gps_range is not None and self.overlap and not self.reducer.exchange
                                 and not self.reducer.sharded)
            if self._early_sumsq:
                self.opt.split_sumsq(*self._gps_range)
            # one row of scalars per segment; all zero = "nothing pending" during the warm-up passes
            self._hyper = torch.zeros(len(self._segments), 16, device=dev, dtype=torch.float32)
            self._hyper_pinned = torch.zeros(len(self._segments), 16, dtype=torch.float32).pin_memory()
        self._pipelined = self.overlap and bool(self.model.video_clips([item["train"], item["target"]])[0])
        # static inputs of the main graph: private copies of every tensor.  Pipelined engine: the video tensors are
        # only consulted for shapes / cache keys inside the main graph (the trunk reads the staging buffers), so
        # they are kept by reference; otherwise the captured trunk reads them and they are copied like gps / gaze.
        self._static_item = {part: {n: (v if (v.dim() == 5 and self._pipelined) else v.clone()) for n, v in item[part].items()}
                             for part in ("train", "target")}
        self._drop_graphs()
        # ONE memory pool for every graph of this engine (trunk, all decision variants, both stages of the split step):
        # they replay one at a time and each is self-contained (static inputs and persistent buffers live outside the
        # pool, a variant's outputs stay allocated), so their intermediates may share memory -- 12 variants cost the
        # memory of one, and a re-capture after a recipe change reuses what the retired graphs gave back
        self._pool = torch.cuda.graph_pool_handle()
        # plan: which host draws does one step make?  One plan per variant of the host dropout decisions (view /
        # gaze dropout, routeformer.py:301,405-410): depth-first over the outcomes, each variant run once eagerly
        # with its decisions imposed (decisions beyond the imposed prefix default to "keep")
        plans, unused, stack = [], [], [[]]
        side = self._streams.get("warmup")
        cap_ = getattr(torch.cuda.graph, "default_capture_stream", None)
        K.debug_line(f"[engine {id(self):x}] warmup stream {side.cuda_stream:x}, current {torch.cuda.current_stream().cuda_stream:x}, "
                     f"capture stream {cap_.cuda_stream if cap_ is not None else 0:x}")
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            while stack:
                forced = stack.pop()
                SAMPLER.plan, SAMPLER.forcing = [], list(forced)
                try:
                    self._eager_fwd_bwd(self._static_item, epoch)
                    plan = SAMPLER.plan
                finally:
                    SAMPLER.plan, SAMPLER.forcing = None, None
                plans.append(plan)
                unused.append(tuple(self.model.__dict__.get("_unused_prefixes", ())))
                outcomes = [op[2] for op in plan if op[0] == "bern"]
                for i in range(len(forced), len(outcomes)):
                    stack.append(outcomes[:i] + [True])
            for _ in range(max(0, warmup - len(plans))):
                SAMPLER.forcing = []
                try:
                    self._eager_fwd_bwd(self._static_item, epoch)
                finally:
                    SAMPLER.forcing = None
        torch.cuda.current_stream().wait_stream(side)
        SAMPLER.make_static(plans, dev)
        self._variant_unused = unused
        SAMPLER.refill_static()
        SAMPLER.select_static(0)
        torch.cuda.synchronize()
        if self._pipelined:
            clips, keys = self.model.video_clips([self._static_item["train"], self._static_item["target"]])
            n = sum(v.shape[0] * idx.numel() for v, idx in clips)
            self._tok_cur = torch.empty(n, 65, 240, device=dev, dtype=torch.float32)
            self._tok_next = torch.empty_like(self._tok_cur)
            self._tstream = self._streams.get("trunk")
            K.debug_line(f"[engine {id(self):x}] trunk stream {self._tstream.cuda_stream:x}")
            self._alloc_clip_stage(item)
            self._stage_clips(item)
            self._trunk_graph().replay()
            self._tok_cur.copy_(self._tok_next)
            self.model.set_video_tokens(self._tok_cur, clips, keys)
            torch.cuda.synchronize()
        self._epoch = epoch
        self._recipe = self._recipe_for(epoch)
        self.graph, self._out = self._main_graph(False, 0)
        self._ready_id = None
        return self

    def _main_graph(self, lookahead: bool, variant: int = 0):
        """Graph of forward + backward on the static inputs.  With ``lookahead`` the conv-trunk pass over the
        staged frames of the NEXT batch is a parallel branch of the SAME graph (forked stream, writes
        ``_tok_next``): two separate graphs launched on two streams do not overlap on ROCm 7.2 (measured:
        14.2 + 5.4 = 19.5 ms), branches of one graph do."""
        la = bool(lookahead and self._pipelined)
        key = (la, variant)
        hit = self._graphs.get(key)
        if hit is not None:
            return hit[0], hit[1]
        clips = None
        if la:
            clips = self._staged()
            self._encode(clips)  # warm caches outside capture (scratch output)
        if self._pipelined:
            c0, k0 = self.model.video_clips([self._static_item["train"], self._static_item["target"]])
            self.model.set_video_tokens(self._tok_cur, c0, k0)
        torch.cuda.synchronize()
        from routeformer_amd.models.blocks import SAMPLER
        now = SAMPLER._variant
        SAMPLER.select_static(variant)  # the captured pass takes this variant's decisions and key-sample slots
        g = _new_graph()
        dot = os.environ.get("RF_GRAPH_DOT")  # diagnosis: keep the hipGraph_t and print it (hipGraphDebugDotPrint)
        if dot:
            g.enable_debug_mode()
        from routeformer_amd import kernels as K
        K.debug_line(f"[engine {id(self):x}] capture variant {variant} lookahead {la} split {self.split}")
        if not self.split:
            with torch.cuda.graph(g, pool=self._pool, **_capture_kw()):
                cur = torch.cuda.current_stream()

                if la:
                    # (the branch forks at the HEAD of the step: forking it after the camera streams' frame encoder, so that
                    #  it runs underneath the latency-bound middle of the step only, was measured at 8.1 instead of 6.8 ms;
                    #  stream priorities -- range (0, -1) on this stack -- make no difference under graph replay:
                    #  profiles/r03/overlap_experiments.txt)
                    self._tstream.wait_stream(cur)
                    with torch.cuda.stream(self._tstream):
                        self._encode(clips, out=self._tok_next)
                out = self._fwd_bwd(self._static_item, self._epoch, tokens_ready=self._pipelined)
                if la:
                    cur.wait_stream(self._tstream)
        else:
            # two graphs over one memory pool, replayed back to back: the host can start the all-reduce of the GPS
            # backbone's gradient buckets between them (see step())
            g2 = _new_graph()
            self._enter()
            try:
                with torch.cuda.graph(g, pool=self._pool, **_capture_kw()):
                    cur = torch.cuda.current_stream()
                    if la:
                        self._tstream.wait_stream(cur)
                        with torch.cuda.stream(self._tstream):
                            self._encode(clips, out=self._tok_next)
                    out, carry = self._stage1(self._static_item, self._epoch, tokens_ready=self._pipelined)
                    if la:
                        cur.wait_stream(self._tstream)
                with torch.cuda.graph(g2, pool=self._pool, **_capture_kw()):
                    self._stage2(carry)
            finally:
                self._leave()
            del carry
            g = (g, g2)
        self.model.clear_video_tokens()
        SAMPLER.select_static(now)
        self._graphs[key] = (g, out)
        if dot and not isinstance(g, tuple):
            os.makedirs(dot, exist_ok=True)
            g.debug_dump(os.path.join(dot, f"graph_{id(self):x}_v{variant}_la{int(la)}.dot"))
        return g, out

    def _multi_copy(self, pairs):
        """dst <- src for every pair with one launch (rf_gather_frames with one "frame" per tensor) instead of one blit
        kernel each; pairs that do not qualify (host tensors, dtype / layout changes) take ``copy_``."""
        from routeformer_amd import _hip
        todo = []
        for src, dst in pairs:
            if (src.is_cuda and dst.is_cuda and src.dtype == dst.dtype and src.shape == dst.shape and src.is_contiguous()
                    and dst.is_contiguous() and src.numel() > 0):
                todo.append((src, dst))
            else:
                dst.copy_(src, non_blocking=True)
        if not todo:
            return
        zero = self.__dict__.get("_zero_idx")
        if zero is None:
            zero = self._zero_idx = torch.zeros(1, dtype=torch.int64, device=todo[0][1].device)
        for s0 in range(0, len(todo), _hip.GATHER_MAX):
            chunk = todo[s0:s0 + _hip.GATHER_MAX]
            arr = (_hip.GatherEntry * len(chunk))()
            for e, (src, dst) in zip(arr, chunk):
                e.src, e.dst, e.idx, e.B, e.T, e.F, e.pad = src.data_ptr(), dst.data_ptr(), zero.data_ptr(), 1, 1, 1, 0
                e.frame_bytes = src.numel() * src.element_size()
            _hip.check(_hip.lib().rf_gather_frames(arr, len(chunk), torch.cuda.current_stream().cuda_stream), "rf_gather_frames")

    def precapture(self, lookahead: bool = True) -> int:
        """Capture the main graph of EVERY variant of the host dropout decisions now (they are otherwise captured on
        first use, i.e. in the middle of training: ~50 ms each).  -> number of graphs held."""
        from routeformer_amd.models.blocks import SAMPLER
        assert self.graph is not None, "capture() first"
        for variant in range(max(1, SAMPLER.n_variants)):
            self._main_graph(False, variant)
            if lookahead and self._pipelined:
                self._main_graph(True, variant)
        torch.cuda.synchronize()
        return len(self._graphs)

    def step(self, item, epoch: int = 0, next_item=None):
        """One train step on ``item``; ``next_item`` (optional) = the batch of the following step, whose
        conv-trunk pass runs underneath this step (a parallel branch of the replayed graph).

        Batches are told apart by an explicit ``item["id"]`` (any hashable), never by tensor addresses: the trunk
        tokens computed ahead for ``next_item`` are used by the following call only if its ``item`` carries the same
        id; without ids every step runs its own trunk pass first (correct, just not pipelined)."""
        from routeformer_amd.models.blocks import SAMPLER
        self.verify_ids(block=False)
        if self.graph is None:
            self.capture(item, epoch)
        recipe = self._recipe_for(epoch)
        if recipe != self._recipe:
            # dense-loss switch (epoch 10) or a new discount (a key of ``discount_factor``): the loss arithmetic in
            # the captured graphs is stale -- capture the main graphs again on the same static buffers
            torch.cuda.synchronize()
            self._drop_graphs(keep_trunk=True)
            self._epoch, self._recipe = epoch, recipe
            self.graph, self._out = self._main_graph(False, 0)
        # host side of the step first: the reference's draws in its order (key samples, view / gaze dropout decisions)
        # into the static buffer; the decisions pick the graph variant to replay (captured on first use)
        variant = SAMPLER.refill_static()
        # device copies this step needs before the replay (the batch's non-video tensors into the static buffers, the
        # look-ahead tokens into place): ONE launch instead of a blit kernel each
        copies = []
        for part in ("train", "target"):
            for n, v in item[part].items():
                dst = self._static_item[part][n]
                if v.dim() == 5 and self._pipelined:
                    continue  # the main graph reads trunk tokens, not clips
                if v.data_ptr() != dst.data_ptr():
                    copies.append((v, dst))
        pending = None  # (item, keys) whose freshly computed tokens go into the cache after the replay
        if not self._pipelined:
            g, out = self._main_graph(False, variant)
        else:
            iid = item.get("id")
            cache = self._cache()
            if self._ready_id is not None and iid is not None and iid == self._ready_id:
                # tokens were prepared under this id by the previous call: by its look-ahead trunk pass (cache attached:
                # are these the frames remembered for the id?) or from the cache (already checked when they were gathered)
                if self._ready_checked != self._clip_ptrs(item):
                    self._check_id_content(item)
            else:
                if not self._cached_tokens_into_next(item):  # cold start / no look-ahead: run this batch's trunk now
                    self._stage_clips(item)
                    self._trunk_graph().replay()
                    if cache is not None:
                        self._remember_tokens(item, cache.keys_of(self._staged()))
            copies.append((self._tok_next, self._tok_cur))
            self._multi_copy(copies)  # (before anything below refills _tok_next for the NEXT batch)
            copies = []
            self._ready_id = None
            # (a next_item without an id could not be recognised by the following call: its trunk pass would be wasted)
            lookahead = next_item is not None and next_item.get("id") is not None
            self._ready_checked = None
            if lookahead and self._cached_tokens_into_next(next_item):
                lookahead = False  # tokens of the next batch are already in _tok_next: no trunk branch this step
                self._ready_checked = self._clip_ptrs(next_item)  # these very tensors have just been hashed
            elif lookahead:
                self._stage_clips(next_item)
                if cache is not None:
                    pending = (next_item, cache.keys_of(self._staged()))
            g, out = self._main_graph(lookahead, variant)
        SAMPLER.select_static(variant)
        self._multi_copy(copies)
        self.reducer.begin_step()  # the replay does not run the Python bookkeeping of zero()
        if self.defer_update:
            self._set_hyper()  # scalars of the update this replay starts with (or "nothing pending")
        if os.environ.get("RF_ENGINE_DEBUG"):
            from routeformer_amd import kernels as K
            K.debug_line(f"[engine {id(self):x}] replay variant {variant} lookahead "
                         f"{isinstance(g, tuple) or (g is self._graphs.get((True, variant), (None,))[0])}")
        if isinstance(g, tuple):
            g[0].replay()
            # 95 % of the gradient bytes (the GPS backbone's) are final here: reduce them underneath stage 2
            self.reducer.launch_complete_prefix(self._names, "gps_backbone.")
            g[1].replay()
        else:
            g.replay()
        if self._pipelined and next_item is not None:
            self._ready_id = next_item.get("id")
            if pending is not None:
                self._remember_tokens(*pending)
        scale = self.reducer.finish()
        if self.defer_update:
            skip = self._skip_ranges(self._variant_unused[variant])
            self.opt.t += 1
            self.opt._note_skipped(skip)
            self._pending = self._pending_rows(scale, skip)  # applied at the start of the next replay (or by flush())
        else:
            self._update(scale, self._skip_ranges(self._variant_unused[variant]))
        return out
