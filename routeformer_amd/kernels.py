"""Autograd glue: ``torch.autograd.Function``s whose forward AND backward are librf_hip.so calls.

Host-side responsibilities only: allocate outputs/workspaces with torch's caching allocator, pass
raw pointers + the current HIP stream over the C ABI, wire gradients.  No arithmetic happens here
and nothing falls back to eager PyTorch: a missing library or a CPU tensor raises.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch

from . import _hip
from ._hip import check, ptr

ACT = {None: 0, "none": 0, "relu": 1, "gelu": 2, "elu": 3}

_PRECISION = 0  # 0 = exact fp32 MFMA, 1 = bf16 MFMA inputs / fp32 accumulate


def set_precision(name: str):
    """``"f32"`` (parity mode) or ``"bf16"`` (matrix-core inputs rounded to bf16)."""
    global _PRECISION
    _PRECISION = {"f32": 0, "fp32": 0, "bf16": 1}[name]


def get_precision() -> str:
    return "bf16" if _PRECISION else "f32"


def _stream():
    return torch.cuda.current_stream().cuda_stream


def debug_line(msg: str):
    """RF_ENGINE_DEBUG=1: one line to RF_DEBUG_FILE (appended, unbuffered -- survives a crash and pytest's capture) or stderr."""
    if os.environ.get("RF_ENGINE_DEBUG"):
        path = os.environ.get("RF_DEBUG_FILE")
        if path:
            fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_APPEND, 0o644)
            os.write(fd, (msg + "\n").encode())
            os.close(fd)
        else:
            import sys
            print(msg, file=sys.stderr, flush=True)


# Independent sub-graphs of one step (the gaze-token encoder vs the video frame encoders; the target-side
# feature pass vs the input forward) are latency-bound chains of small kernels: running them on separate
# HIP streams lets the otherwise idle CUs overlap them.  Off by default (strict reference call order for
# the test hooks); the training engine switches it on.  Host-RNG draw order is unaffected.
OVERLAP = False
OVERLAP_MASK = int(os.environ.get("RF_OVERLAP", "11"))  # bit0 target pass, bit1 gaze encoder, bit3 decoder self-attention block beside the encoder


def _distinct_stream(taken):
    """A torch stream whose HIP handle differs from every handle in ``taken`` and from the current stream.
    ``torch.cuda.Stream()`` hands out a pool of 32 handles round-robin: in a long-lived process (a test session that
    builds a dozen engines) a "new" stream can BE the capture stream or another side stream -- a fork onto it then
    serialises silently, and ``on_side_stream`` reports the main stream as a side stream."""
    cur = torch.cuda.current_stream().cuda_stream
    for _ in range(64):
        st = torch.cuda.Stream()
        if st.cuda_stream != cur and st.cuda_stream not in taken:
            return st
    raise RuntimeError("no distinct HIP stream left in torch's stream pool")


class SideStreams:
    """The side streams of ONE owner (a training engine; the module-level default serves direct calls from tests and
    tools) plus the bookkeeping of an open fork: which of them were forked since the last join, and the tensors their
    kernels read (kept referenced until the join: no early reuse by the caching allocator).

    Engine-scoped on purpose (round 4).  The round-3 form was one process-global dict whose join waited on EVERY
    stream ever created -- including streams no fork of the current capture had touched (another engine's "update" /
    "wgrad" stream, the "gaze" stream in a variant that dropped the gaze branch): inside a stream capture that records an
    event on a stream that is NOT capturing and makes the capturing stream wait on it.  The runtime tolerates it
    (tools/probes/capture_probe.hip, mode bit 2), but it ties every capture to the history of the process; a join now
    names exactly the streams its own forks opened."""

    def __init__(self):
        self.streams = {}   # (key, device) -> torch.cuda.Stream
        self.forked = {}    # HIP handle -> stream, forked since the last join
        self.keepalive = []

    def handles(self):
        return {st.cuda_stream for st in self.streams.values()}

    def get(self, key: str):
        dev = torch.cuda.current_device()
        st = self.streams.get((key, dev))
        if st is None:
            taken = self.handles()
            cap = getattr(torch.cuda.graph, "default_capture_stream", None)
            if cap is not None:
                taken.add(cap.cuda_stream)
            st = self.streams[(key, dev)] = _distinct_stream(taken)
            debug_line(f"[streams {id(self):x}] {key} -> {st.cuda_stream:x} (current {torch.cuda.current_stream().cuda_stream:x})")
        return st

    def fork(self, key: str, origin=None, keep=None):
        """Stream ``key`` made to wait for everything queued on ``origin`` (default: the current stream); the join
        that ends the step (or the stage) waits for it."""
        st = self.get(key)
        st.wait_stream(origin if origin is not None else torch.cuda.current_stream())
        self.forked[st.cuda_stream] = st
        if keep is not None:
            self.keepalive.append(keep)
        return st

    def contains(self, stream) -> bool:
        h = stream.cuda_stream
        return any(st.cuda_stream == h for st in self.streams.values())

    def join(self, into=None):
        """``into`` (default: the current stream) waits for every stream forked since the last join -- and for no other.
        Inside a stream capture only streams that are themselves capturing are waited for: a forked stream that is not
        (the second graph of the split step inherits the first one's fork set, and autograd pulls only some of those
        streams into it) holds no work of this capture.  Returns the joined fork set (handle -> stream)."""
        cur = into if into is not None else torch.cuda.current_stream()
        todo = dict(self.forked)
        capturing = _is_capturing(cur)
        for h, st in todo.items():
            if h != cur.cuda_stream and (not capturing or _is_capturing(st)):
                cur.wait_stream(st)
        self.forked.clear()
        self.keepalive.clear()
        return todo


def _is_capturing(stream) -> bool:
    with torch.cuda.stream(stream):
        return torch.cuda.is_current_stream_capturing()


STREAMS = SideStreams()


def side_stream(key: str):
    return STREAMS.get(key)


def fork_side_stream(key: str, origin=None, keep=None):
    return STREAMS.fork(key, origin, keep)


def on_side_stream() -> bool:
    """True when the current stream is itself one of the side streams: the fork sites do not nest.  (Round 1 blamed a
    crash on nested forks; tools/probes/capture_probe.hip mode bit 4 captures and replays them without trouble -- that
    crash was most likely the destroyed-graph defect of DESIGN section 5b.  The sites stay un-nested because a branch of
    a branch has nothing left to overlap with.)"""
    return STREAMS.contains(torch.cuda.current_stream())


def join_side_streams():
    """Make the current stream wait for everything queued on the side streams forked since the last join.  Needed after
    backward: gradient sinks written by side-stream kernels bypass autograd's own leaf-stream synchronisation."""
    STREAMS.join()


class _Profiler:
    """HIP-event timing of kernel classes on the launch stream (bench.py's live roofline numbers).
    Off by default: a single attribute test per launch."""

    def __init__(self):
        self.on = False
        self.events = []

    def enable(self):
        self.on, self.events = True, []

    def disable(self):
        self.on = False

    def begin(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, tag, start, flops, nbytes, replay=None):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.events.append((tag, start, e, flops, nbytes, replay))

    def refine(self, tag, reps: int = 5):
        """Per-launch EXECUTION time of kernel ``tag``: the step's launches of it are re-issued (same operands, same
        shapes) as one uninterrupted stream of work -- a few untimed rounds first, so the timed ones run at load
        clocks like the step itself (an isolated launch after an idle gap reads ~30 % longer) -- each timed launch
        carrying a start / stop event pair that brackets exactly its dispatch (rf_kernel_timer_arm,
        hipExtLaunchKernelGGL): the duration rocprofv3's kernel trace reports, without dispatch turnaround or host
        launch path.  For a call that launches two kernels the first one is timed.
        Returns (launches, total_us_per_step, flops, bytes) or None."""
        import ctypes
        todo = [(fl, by, rp) for t, _, _, fl, by, rp in self.events if t == tag and rp is not None]
        if not todo or len(todo) * reps > 4096:
            return None
        lib = _hip.lib()
        for _ in range(int(os.environ.get("RF_REFINE_WARM", "3"))):  # clocks up, caches as in the step
            for _, _, rp in todo:
                rp()
        for _ in range(reps):
            for _, _, rp in todo:
                check(lib.rf_kernel_timer_arm(), "rf_kernel_timer_arm")
                rp()
        buf = (ctypes.c_float * (len(todo) * reps))()
        n = lib.rf_kernel_timer_collect(buf, len(buf))
        vals = [buf[i] for i in range(n)]
        if n != len(buf) or any(v < 0 for v in vals):
            return None
        total_us = sum(vals) / reps
        if os.environ.get("RF_REFINE_DEBUG"):
            import sys
            per = [sum(vals[i::len(todo)]) / reps for i in range(len(todo))]
            print(f"[refine] {tag}: " + " ".join(f"{v:.1f}" for v in per), file=sys.stderr)
        return len(todo), total_us, sum(fl for fl, _, _ in todo), sum(by for _, by, _ in todo)

    def summary(self):
        """tag -> {launches, total_ms, flops, bytes} (algorithmic flops / bytes summed over launches)."""
        if not self.events:
            return {}
        torch.cuda.synchronize()
        out = {}
        for tag, s, e, fl, by, _ in self.events:
            d = out.setdefault(tag, {"launches": 0, "total_ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["total_ms"] += s.elapsed_time(e)
            d["flops"] += fl
            d["bytes"] += by
        return out


PROFILE = _Profiler()


class _Rng:
    """Device-side state of the dropout-mask generator (csrc/philox.h): two uint64 {seed, step} per device.

    A keep-bit is a pure function of (seed, step, site, element index): ``site`` numbers the dropout calls of one
    step in host program order (reset by ``begin_step``), ``step`` is advanced on the DEVICE (``rf_rng_advance``,
    the first launch of every training step -- so a HIP-graph replay of the step draws fresh masks), and a
    backward kernel regenerates the mask of its forward from the site number it kept.
    Test hooks: ``forced`` = keep-masks consumed in call order instead of the generator (parity against the
    reference's recorded masks); ``record`` = list receiving the keep-mask of every call (materialised by
    rf_dropout's mask_out), e.g. to hand the product's Philox masks to the CPU oracle."""

    def __init__(self):
        self._state = {}
        self.site = 0
        self.seed_value: Optional[int] = None
        self.forced: Optional[list] = None
        self.record: Optional[list] = None

    def state(self, device) -> torch.Tensor:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        st = self._state.get(idx)
        if st is None:
            st = self._state[idx] = torch.zeros(2, dtype=torch.int64, device=torch.device("cuda", idx))
            seed = self.seed_value if self.seed_value is not None else torch.initial_seed()
            check(_hip.lib().rf_rng_seed(ptr(st), int(seed) & 0x7FFFFFFFFFFFFFFF, 0, _stream()), "rf_rng_seed")
        return st

    def manual_seed(self, seed: int):
        """Seed the mask generator (default: ``torch.initial_seed()`` at first use); restarts the step counter."""
        self.seed_value = int(seed)
        for st in self._state.values():
            with torch.cuda.device(st.device):
                check(_hip.lib().rf_rng_seed(ptr(st), self.seed_value & 0x7FFFFFFFFFFFFFFF, 0, _stream()), "rf_rng_seed")
        self.site = 0

    def begin_step(self, device=None):
        """First launch of a training step: step += 1 on the device, site numbering restarts."""
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        check(_hip.lib().rf_rng_advance(ptr(self.state(dev)), _stream()), "rf_rng_advance")
        self.site = 0

    def next_site(self) -> int:
        s = self.site
        self.site += 1
        return s

    def take_forced(self, shape, device):
        """Next injected keep-mask as a uint8 device tensor of logical ``shape`` (the reference drops the FFN's hidden
        activation in its (B, d_ff, L) Conv1d layout: such a mask is transposed to (B, L, d_ff))."""
        if self.forced is None:
            return None
        m = self.forced.pop(0)
        if tuple(m.shape) != tuple(shape):
            assert m.dim() == 3 and tuple(m.transpose(1, 2).shape) == tuple(shape), (tuple(m.shape), tuple(shape))
            m = m.transpose(1, 2)
        return m.to(device=device, dtype=torch.uint8).contiguous()

    def merge_forced(self, n_calls: int, sites_per_call: int):
        """``n_calls`` reference encoder calls run as one batched call: [call][site] masks -> per-site batch-concatenated."""
        if self.forced is None or n_calls == 1:
            return
        k = n_calls * sites_per_call
        head, rest = self.forced[:k], self.forced[k:]
        self.forced = [torch.cat([head[c * sites_per_call + j] for c in range(n_calls)], dim=0)
                       for j in range(sites_per_call)] + rest

    def materialise(self, site: int, shape, p: float, device, mask=None) -> torch.Tensor:
        """The keep-mask (bool) the generator yields for ``site`` in the current step (or the injected one)."""
        n = 1
        for d in shape:
            n *= int(d)
        out = torch.empty(n + 3, dtype=torch.uint8, device=device)[:n]
        check(_hip.lib().rf_dropout(None, None, n, p, ptr(self.state(device)), site, ptr(mask), ptr(out), _stream()),
              "rf_dropout(mask_out)")
        return out.view(tuple(shape)).bool()


RNG = _Rng()


def _drop_launch(x, y, p: float, site: int, mask):
    """y = x * keep / (1 - p) over a contiguous tensor (x is y: in place); forward and backward alike."""
    check(_hip.lib().rf_dropout(ptr(x), ptr(y), x.numel(), p, ptr(RNG.state(x.device)), site, ptr(mask), None, _stream()),
          "rf_dropout")


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, site, mask):
        _req(x, "dropout.x")
        x = x.contiguous()
        y = torch.empty_like(x)
        _drop_launch(x, y, p, site, mask)
        ctx.cfg = (p, site, mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, site, mask = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        _drop_launch(dy, dx, p, site, mask)
        return dx, None, None, None


def dropout(x, p: float, training: bool = True):
    """nn.Dropout(p)(x) in train mode with a device-side Philox mask (never stored; see ``_Rng``)."""
    if p <= 0.0 or not training:
        return x
    mask = RNG.take_forced(x.shape, x.device)
    site = RNG.next_site()
    if RNG.record is not None:
        RNG.record.append(RNG.materialise(site, x.shape, p, x.device, mask))
    return _Dropout.apply(x, p, site, mask)


def _req(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise _hip.HipLibraryError(
            f"{what}: routeformer_amd runs on the GPU only (got a {t.device} tensor); there is no CPU path")
    if t.dtype != torch.float32:
        raise TypeError(f"{what}: expected float32, got {t.dtype}")


def _splits(tiles: int, depth: int) -> int:
    """Split-K factor for skinny-output GEMMs (weight gradients): aim for >= 512 workgroups while
    keeping >= 128 of reduction depth per slice."""
    return max(1, min(64, depth // 128, -(-512 // max(tiles, 1))))


_TILE_COUNTERS = {}
# In-launch split-K reduction (last-arriving workgroup sums the slabs behind an agent-scope release/acquire
# hand-off) is implemented and tested, but measured SLOWER than the separate reduction launch at these slab
# sizes (416 vs 460 samples/s: every workgroup pays the L2 write-back of the release fence), so it is off.
IN_LAUNCH_SPLITK_REDUCE = __import__("os").environ.get("RF_SPLITK_INLAUNCH", "0") == "1"


def _tile_counters(device):
    """Split-K arrival counters: one persistent zeroed buffer per (device, stream) -- launches on one
    stream are ordered, and every launch leaves its counters at zero again."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    buf = _TILE_COUNTERS.get(key)
    if buf is None:
        buf = _TILE_COUNTERS[key] = torch.zeros(4096, device=device, dtype=torch.int32)
    return buf


def _auto_split(M: int, N: int, K: int) -> int:
    """Split-K for launches that would leave most of the 256 CUs idle (small-M Informer GEMMs).  Measured on the
    step's shapes (tools/splitk_sweep.py): a reduction depth >= 1024 wants ~512 workgroups (the slices are long
    enough to pay for the extra reduction pass), a shallower one only enough slices to reach ~256, and in
    either case >= 128 of depth per slice."""
    tm = -(-M // (128 if M >= 4096 and N >= 64 else 64))
    tiles = tm * -(-N // 64)
    if tiles >= 256 or K < 256:
        return 1
    want = -(-_SPLIT_WG_DEEP // tiles) if K >= 1024 else _SPLIT_WG // tiles
    return max(1, min(16, K // 128, want))


_SPLIT_WG = int(os.environ.get("RF_SPLIT_WG", "256"))            # workgroups a shallow (K < 1024) product is split towards
_SPLIT_WG_DEEP = int(os.environ.get("RF_SPLIT_WG_DEEP", "512"))  # ... and a deep one


# Skinny GEMM (csrc/gemm_skinny.hip) for the small-M linear layers of the GPS backbone: RF_SKINNY=0 turns it off,
# RF_SKINNY_MAX_M bounds the row count it takes (larger M re-reads the activation per 16-column tile: the tiled kernel wins)
# (measured, tools/skinny_sweep.py under graph replay: K <= 1024 wins up to ~192 rows -- 6-10 us against 10.5-17.5 us
#  for the tiled kernel + slab sum; K > 1024 needs slices + a slab sum itself and wins up to ~64 rows)
SKINNY_GEMM = os.environ.get("RF_SKINNY", "1") != "0"
SKINNY_MAX_M = int(os.environ.get("RF_SKINNY_MAX_M", "192"))
SKINNY_MAX_M_DEEP = int(os.environ.get("RF_SKINNY_MAX_M_DEEP", "64"))


def gemm(A, lda_m, lda_k, B, ldb_k, ldb_n, C, ldc, M, N, K, *, bias=None, residual=None, ldr=0,
         res_rows=0, res_before_act=0, act=0, preact=None, ldp=0, dact_src=None, ldd=0, dact=0,
         splitk=0, atomic=False, a_rowsum=None):
    if (SKINNY_GEMM and _PRECISION == 1 and not atomic and splitk == 0 and a_rowsum is None
            and M <= (SKINNY_MAX_M if K <= 1024 else SKINNY_MAX_M_DEEP)):
        z = _hip.lib().rf_gemm_skinny_split(ptr(A), lda_m, lda_k, ptr(B), ldb_k, ldb_n, M, N, K)
        if z > 0:
            ws = torch.empty(z * M * N, device=C.device, dtype=torch.float32) if z > 1 else None
            ev = PROFILE.begin() if PROFILE.on else None
            args = (ptr(A), lda_m, lda_k, ptr(B), ldb_k, ldb_n, ptr(C), ldc, M, N, K, ptr(bias), ptr(residual), ldr, res_rows,
                    res_before_act, act, ptr(preact), ldp, ptr(dact_src), ldd, dact, ptr(ws))
            check(_hip.lib().rf_gemm_skinny(*args, _stream()), "rf_gemm_skinny")
            if ev is not None:
                rtm = 1 if M <= 16 else (2 if M <= 32 else 4)
                tag = f"gemm_skinny_kernel<{rtm}, {2 if (M > 64 and N >= 1024) else 1}, {0 if ldb_k == 1 else 1}>"
                if z > 1:
                    tag += " + skinny_reduce_kernel"
                keep = (A, B, C, bias, residual, preact, dact_src, ws)
                PROFILE.end(tag, ev, 2.0 * M * N * K, 4.0 * (M * K + K * N + M * N),
                            replay=lambda a=args, k=keep: _hip.lib().rf_gemm_skinny(*a, _stream()))
            return
    if splitk == 0:
        splitk = _auto_split(M, N, K)
    ws = cnt = None
    if splitk > 1 and not atomic:
        ws = torch.empty(splitk * M * N, device=C.device, dtype=torch.float32)
        if IN_LAUNCH_SPLITK_REDUCE:
            cnt = _tile_counters(C.device)
    ev = PROFILE.begin() if PROFILE.on else None
    args = (ptr(A), lda_m, lda_k, ptr(B), ldb_k, ldb_n, ptr(C), ldc, M, N, K, ptr(bias), ptr(residual), ldr, res_rows,
            res_before_act, act, ptr(preact), ldp, ptr(dact_src), ldd, dact, _PRECISION, splitk, ptr(ws),
            1 if atomic else 0, ptr(a_rowsum), ptr(cnt))
    check(_hip.lib().rf_gemm(*args, _stream()), "rf_gemm")
    if ev is not None:  # tag = the kernel symbol rf_gemm dispatches to (same rules as csrc/gemm.hip)
        def mode(t, ld_k, ld_row):
            al = t.data_ptr() % 16 == 0
            return 0 if (ld_k == 1 and ld_row % 4 == 0 and al) else (1 if (ld_row == 1 and ld_k % 4 == 0 and al) else 2)
        am, bm = mode(A, lda_k, lda_m), mode(B, ldb_k, ldb_n)
        if am <= 1 and bm <= 1:
            tag = f"gemm2_kernel<{_PRECISION}, {am}, {bm}, {3 if (M >= 4096 and N >= 64) else 0}>"
        else:
            tag = f"gemm_kernel<{_PRECISION}, {am}, {bm}, 0>"
        if splitk > 1 and not atomic:  # two kernels per call: kept apart from the single-kernel launches of the symbol
            tag += " + splitk_reduce_kernel"
        keep = (A, B, C, bias, residual, preact, dact_src, ws, a_rowsum)  # operands stay alive for the replay
        PROFILE.end(tag, ev, 2.0 * M * N * K, 4.0 * (M * K + K * N + M * N),
                    replay=lambda a=args, k=keep: _hip.lib().rf_gemm(*a, _stream()))


# The GPS backbone's out-projection / Conv1d(k=1) pair in front of a LayerNorm is a split-K product, a slab-sum launch and the
# norm: with RF_SLAB_LN (default on) the product leaves its raw slabs and the norm's loads sum them (rf_layernorm_fwd_slabs;
# bit-identical arithmetic, one launch less per site)
SLAB_LN = os.environ.get("RF_SLAB_LN", "1") != "0"


def _partials_plan(a_ptr, lda: int, w, M: int, N: int, K: int, ldb_k: int = 1, ldb_n: int = 0):
    """How ``gemm`` would run y = a w^T (a (M, K) with row pitch ``lda`` at ``a_ptr``, w (N, K) contiguous) if it needs a
    slab-sum launch: ("skinny", slabs) / ("tiled", splitk, slabs), or None when the product finishes in its own launch
    (or the slab path is off).  Same decisions as ``gemm``.  ``ldb_k`` / ``ldb_n``: element strides of B[k][n] inside ``w``
    (default: w (N, K) row-major; a dX product y = a w with w (K, N) row-major passes (N, 1))."""
    ldb_n = ldb_n or K
    if not SLAB_LN or IN_LAUNCH_SPLITK_REDUCE or N > 1024 or N <= 256 or not w.is_cuda:
        return None  # (N <= 256: the plain norm is the wave-per-row kernel there, another reduction tree)
    if SKINNY_GEMM and _PRECISION == 1 and M <= (SKINNY_MAX_M if K <= 1024 else SKINNY_MAX_M_DEEP):
        z = _hip.lib().rf_gemm_skinny_split(a_ptr, lda, 1, ptr(w), ldb_k, ldb_n, M, N, K)
        if z > 0:
            return ("skinny", z) if z > 1 else None
    splitk = _auto_split(M, N, K)
    if splitk <= 1:
        return None
    eff = _hip.lib().rf_gemm_split_count(K, splitk)
    return ("tiled", splitk, eff) if eff > 1 else None


def _gemm_partials(x2, w, M: int, N: int, K: int, plan, ldb_k: int = 1, ldb_n: int = 0):
    """Launch the product of ``plan`` (see ``_partials_plan``) -> (slabs [splits, M, N], splits)."""
    ldb_n = ldb_n or K
    splits = plan[-1]
    ws = torch.empty(splits * M * N, device=x2.device, dtype=torch.float32)
    ev = PROFILE.begin() if PROFILE.on else None
    if plan[0] == "skinny":
        args = (ptr(x2), x2.stride(0), 1, ptr(w), ldb_k, ldb_n, M, N, K, ptr(ws))
        fn, name = _hip.lib().rf_gemm_skinny_partials, "rf_gemm_skinny_partials"
        rtm = 1 if M <= 16 else (2 if M <= 32 else 4)
        tag = f"gemm_skinny_kernel<{rtm}, {2 if (M > 64 and N >= 1024) else 1}, {0 if ldb_k == 1 else 1}>"
    else:
        args = (ptr(x2), x2.stride(0), 1, ptr(w), ldb_k, ldb_n, M, N, K, _PRECISION, plan[1], ptr(ws))
        fn, name = _hip.lib().rf_gemm_partials, "rf_gemm_partials"
        am = 0 if (x2.stride(0) % 4 == 0 and K % 4 == 0 and x2.data_ptr() % 16 == 0) else 2
        if ldb_k == 1:
            bm = 0 if (K % 4 == 0 and ldb_n % 4 == 0 and w.data_ptr() % 16 == 0) else 2
        else:
            bm = 1 if (ldb_n == 1 and ldb_k % 4 == 0 and N % 4 == 0 and w.data_ptr() % 16 == 0) else 2
        tag = (f"gemm2_kernel<{_PRECISION}, {am}, {bm}, {3 if (M >= 4096 and N >= 64) else 0}>" if am <= 1 and bm <= 1
               else f"gemm_kernel<{_PRECISION}, {am}, {bm}, 0>")
    check(fn(*args, _stream()), name)
    if ev is not None:
        keep = (x2, w, ws)
        PROFILE.end(tag, ev, 2.0 * M * N * K, 4.0 * (M * K + K * N + splits * M * N),
                    replay=lambda a=args, k=keep, f=fn: f(*a, _stream()))
    return ws, splits


def _ln_fwd_slabs(ws, splits, bias, r2, gamma, beta, M: int, N: int, eps: float, need_grad: bool, unfold_L: int = 0):
    """LayerNorm(sum of slabs + bias + r2) -> (y, xhat, rstd).  ``unfold_L`` = L > 0: y comes out as the (B, L + 2, 3 N)
    im2col image of the distilling convolution that consumes it (rf_layernorm_fwd_slabs_unfold)."""
    xhat = torch.empty(M, N, device=ws.device, dtype=torch.float32) if need_grad else None
    rstd = torch.empty(M, device=ws.device, dtype=torch.float32) if need_grad else None
    ev = PROFILE.begin() if PROFILE.on else None
    if unfold_L:
        y = torch.empty(M // unfold_L, unfold_L + 2, 3 * N, device=ws.device, dtype=torch.float32)
        check(_hip.lib().rf_layernorm_fwd_slabs_unfold(ptr(ws), splits, ptr(bias), ptr(r2), ptr(gamma), ptr(beta), ptr(y),
                                                       ptr(xhat), ptr(rstd), M, N, unfold_L, eps, _stream()),
              "rf_layernorm_fwd_slabs_unfold")
    else:
        y = torch.empty(M, N, device=ws.device, dtype=torch.float32)
        check(_hip.lib().rf_layernorm_fwd_slabs(ptr(ws), splits, ptr(bias), ptr(r2), ptr(gamma), ptr(beta), ptr(y), ptr(xhat),
                                                ptr(rstd), M, N, eps, _stream()), "rf_layernorm_fwd_slabs")
    if ev is not None:
        PROFILE.end("layernorm_fwd_kernel", ev, (8.0 + splits) * M * N, 4.0 * M * N * (splits + 2 + (xhat is not None)))
    return y, xhat, rstd


class _Ctx:
    """Stand-in for an autograd context when one Function's backward runs another's (attributes set by the caller)."""


# Gradients that travel between two autograd nodes as split-K slabs instead of a finished tensor: the FFN's last dX product
# (d LayerNorm-1 output = dZ W1 + skip gradient) leaves its slabs, and the LayerNorm backward that consumes it sums them on
# load (rf_layernorm_bwd_slabs) -- one slab-sum launch less per layer.  The producer returns an UNWRITTEN placeholder tensor and
# registers (slabs, splits, skip gradient, placeholder) under the placeholder's address; only a consumer that looks its
# incoming gradient up here may follow.  The model code asks for it (``ffn_add_layer_norm(sole_consumer=True)``) where the
# layer's structure guarantees that: the normalised tensor goes to the FFN and nowhere else, and it was produced by
# ``_LinearAddLNSlabs``.  The engine checks after every backward pass that nothing was left unconsumed.
# One-shot callbacks the training engine plants for a step and the model fires at a point of its forward pass
# ("after_frame_embedding": the camera-token embedding of the main stream has been issued -- engine._begin_step_kernels)
STEP_HOOKS = {}

LAZY = {}
LAZY_COUNT = [0]  # registrations so far (tests: the slab-carried paths were really taken)
LAZY_DX = os.environ.get("RF_LAZY_DX", "1") != "0"
# ... the out-projection's input gradient into the attention backward (rf_attn_bwd_slabs): measured neutral (5.265 / 5.337 / 5.327
# -> 5.274 / 5.323 ms: the three launches it removes against slab loads inside an instruction-bound kernel) -- off; RF_LAZY_ATTN=1
LAZY_ATTN = os.environ.get("RF_LAZY_ATTN", "0") == "1"
LAZY_BN_FWD = os.environ.get("RF_LAZY_BN_FWD", "1") != "0"  # ... the distilling convolution's product into the BatchNorm tail


def colsum(X2d: torch.Tensor, into: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """Column sums; ``into`` = slot of the flat gradient buffer to ACCUMULATE into (returns None then)."""
    M, N = X2d.shape
    if X2d.dtype != torch.float32:
        X2d = X2d.float()
    out = into if into is not None else torch.empty(N, device=X2d.device, dtype=torch.float32)
    if into is not None and not DETERMINISTIC:  # a gradient slot: atomics like every other sink writer, one launch
        check(_hip.lib().rf_colsum(ptr(X2d), X2d.stride(0), M, N, ptr(out), 2, None, _stream()), "rf_colsum")
        return None
    parts = _hip.lib().rf_colsum_parts(M, N)
    ws = torch.empty(parts * N, device=X2d.device, dtype=torch.float32)
    check(_hip.lib().rf_colsum(ptr(X2d), X2d.stride(0), M, N, ptr(out), 1 if into is not None else 0, ptr(ws),
                               _stream()), "rf_colsum")
    return None if into is not None else out


DETERMINISTIC = False  # True: split-K through a workspace + ordered reduction instead of fp32 atomics


def _vec_ok(t: torch.Tensor) -> bool:
    return t.data_ptr() % 16 == 0 and t.stride(0) % 4 == 0 and t.stride(1) == 1


def _weight_grad(dy2: torch.Tensor, x2: torch.Tensor, into: Optional[torch.Tensor] = None, bias_into=None):
    """dW[N,K] = dY[M,N]^T X[M,K]  (split-K over the row dimension M).  ``into``: accumulate straight into
    that (N,K)-shaped slot of the flat gradient buffer.  With ``bias_into`` the bias gradient (column sums
    of dY) is produced by the same launch; returns True when it was (else the caller runs ``colsum``)."""
    M, N = dy2.shape
    K = x2.shape[1]
    tiles = -(-N // 64) * -(-K // 64)
    if dy2.dtype == torch.bfloat16 or x2.dtype == torch.bfloat16:
        # operands a fused stack kept as bf16: only the grouped transposed-read kernel takes them as they are (both or neither)
        if not (dy2.dtype == x2.dtype and into is not None and not DETERMINISTIC and WGRAD.active and into.is_contiguous() and _PRECISION == 1
                and os.environ.get("RF_WGRAD_TR", "1") != "0" and _vec_ok(dy2) and _vec_ok(x2)
                and into.data_ptr() % 16 == 0 and N % 4 == 0 and K % 4 == 0):
            dy2, x2 = dy2.float(), x2.float()
    if (into is not None and not DETERMINISTIC and _vec_ok(dy2) and _vec_ok(x2) and into.data_ptr() % 16 == 0
            and N % 4 == 0 and K % 4 == 0):
        if WGRAD.active and into.is_contiguous():
            WGRAD.push(dy2, x2, into, bias_into, M, N, K, _splits(tiles, M))
            return bias_into is not None
        gemm(dy2, 1, dy2.stride(0), x2, x2.stride(0), 1, into, K, N, K, M, splitk=_splits(tiles, M), atomic=True,
             a_rowsum=bias_into)
        return bias_into is not None
    if into is not None:
        gemm(dy2, 1, dy2.stride(0), x2, x2.stride(0), 1, into, K, N, K, M, residual=into, ldr=K, res_rows=N,
             splitk=_splits(tiles, M))
        return None
    dw = torch.empty(N, K, device=dy2.device, dtype=torch.float32)
    gemm(dy2, 1, dy2.stride(0), x2, x2.stride(0), 1, dw, K, N, K, M, splitk=_splits(tiles, M))
    return dw


class _WgradQueue:
    """Deferred weight gradients (training engine only).  dW / db feed nothing but the optimizer, so
    ``_weight_grad`` queues them while the backward pass runs and the queue is flushed -- per stream, in
    launch order -- as grouped launches (``rf_wgrad_grouped``: up to 48 problems, thousands of workgroups)
    instead of one small split-K launch per layer in the middle of the dX chain.  The queued operands are
    kept alive until their launch; "slot written" notifications (DP bucket bookkeeping) are delivered at
    flush time."""

    def __init__(self):
        self.active = False
        self.side_early = False  # set by TrainEngine._fwd_bwd: groups that fill up mid-backward go to a side stream
        self.queues = {}     # stream handle -> (torch stream, [items])
        self.pending = set() # data_ptr of slots with a queued (not yet launched) write
        self.written = set() # data_ptr of slots already written by a grouped launch this step
        self.side_slots = set()  # ... of those, the ones a group on the "wgrad" side stream wrote (not yet joined)
        self._watch = None       # (slot data_ptrs, callback(stream)): see when_launched

    def begin_step(self):
        """Gradient slots were just zeroed: the first (and, per launch, only) writer of a slot may use plain
        stores instead of atomics (``RfWgradEntry.exclusive``)."""
        self.written.clear()
        self.side_slots.clear()
        self._watch = None

    def when_launched(self, slots, callback):
        """Call ``callback(stream)`` once no queued write to any of ``slots`` (data_ptrs) is outstanding: now, on the
        current stream, or from the flush that launches the last of them, on that group's stream."""
        if not (self.pending & slots):
            callback(torch.cuda.current_stream())
        else:
            self._watch = (slots, callback)

    def push(self, dy2, x2, into, bias_into, M, N, K, splits):
        st = torch.cuda.current_stream()
        q = self.queues.setdefault(st.cuda_stream, (st, []))[1]
        q.append((dy2, x2, into, bias_into, M, N, K, splits))
        self.pending.add(into.data_ptr())
        if bias_into is not None:
            self.pending.add(bias_into.data_ptr())
        if len(q) >= _hip.WGRAD_MAX_GROUP:
            self._flush(st, q, side=self.side_early)

    # (flushing earlier, on a side stream underneath the dX chain, was measured: 560-590 vs 610 samples/s --
    #  the chip-filling launch slows the latency-bound chain more than it hides)

    def _flush(self, st, q, side=False):
        """``side``: launch on the "wgrad" side stream (forked from ``st`` here, joined by join_side_streams): a group
        that fills up in the MIDDLE of the backward pass -- the GPS backbone's 48 weights, 288 MB of gradients -- then
        runs underneath the rest of the backward instead of in front of it."""
        if not q:
            return
        slots = {p_ for it in q for p_ in (it[2].data_ptr(), None if it[3] is None else it[3].data_ptr())}
        if side and OVERLAP and not STREAMS.contains(st):
            st = fork_side_stream("wgrad", origin=st, keep=list(q))
            self.side_slots |= slots
        else:
            if self.side_slots & slots:
                # a slot the side-stream group wrote (plain exclusive stores) gets another contribution here: order the
                # two launches (they were unordered until the final join -- a race on dW)
                st.wait_stream(STREAMS.get("wgrad"))
                self.side_slots.clear()
            if STREAMS.contains(st):
                STREAMS.forked[st.cuda_stream] = st  # work queued behind autograd's own leaf-stream sync: the join waits for it
        n = len(q)
        arr = (_hip.WgradEntry * n)()
        uses = {}
        for item in q:
            uses[item[2].data_ptr()] = uses.get(item[2].data_ptr(), 0) + 1
        # Split-K was sized per problem (>= 512 workgroups each); a group that fills the chip on its own tiles needs
        # it only to bound the reduction depth a single workgroup walks.  Every split of a d=832 weight is another
        # read-modify-write pass of fp32 atomics over it: PMC showed 3 GB of HBM traffic per step here against 0.3 GB
        # of gradients, and an unsplit tile may use plain exclusive stores.
        total_tiles = sum(-(-i[5] // 64) * -(-i[6] // 64) for i in q)
        for e, (dy2, x2, into, bias_into, M, N, K, splits) in zip(arr, q):
            if total_tiles >= _WGRAD_GROUP_TILES:
                splits = max(1, min(splits, -(-M // 1024)))
            e.dy, e.x, e.dw, e.db = ptr(dy2), ptr(x2), ptr(into), ptr(bias_into)
            e.M, e.N, e.K, e.ld_dy, e.ld_x, e.splits = M, N, K, dy2.stride(0), x2.stride(0), splits
            e.dy_bf16, e.x_bf16 = int(dy2.dtype == torch.bfloat16), int(x2.dtype == torch.bfloat16)
            e.exclusive = 1 if (uses[into.data_ptr()] == 1 and into.data_ptr() not in self.written) else 0
        self.written.update(uses)
        ev = PROFILE.begin() if PROFILE.on else None
        if os.environ.get("RF_WGRAD_DEBUG"):
            import collections, sys
            hist = collections.Counter((i[4], i[5], i[6], e_.splits, e_.exclusive) for i, e_ in zip(q, arr))
            print(f"[wgrad group] {n} problems: " + ", ".join(f"{c}x(M={m},N={n_},K={k},s={s_},x={x})" for (m, n_, k, s_, x), c in
                                                               sorted(hist.items())), file=sys.stderr)
        with torch.cuda.stream(st):
            check(_hip.lib().rf_wgrad_grouped(arr, n, _PRECISION, st.cuda_stream), "rf_wgrad_grouped")
            if ev is not None:
                keep = (arr, list(q))  # operands stay alive for the replay (bench's per-launch timing; it re-accumulates
                #                        into the gradient slots, which nothing reads after the profile pass)
                tag = ("wgrad_tr_kernel" if (_PRECISION == 1 and os.environ.get("RF_WGRAD_TR", "1") != "0")
                       else f"wgrad_grouped_kernel<{_PRECISION}>")  # (the symbol rf_wgrad_grouped dispatches to)
                PROFILE.end(tag, ev, sum(2.0 * i[4] * i[5] * i[6] for i in q),
                            # algorithmic bytes: both operands once, dW written once (plain exclusive stores) or read +
                            # written (atomic accumulation into a slot another launch also writes)
                            sum(i[0].element_size() * i[4] * i[5] + i[1].element_size() * i[4] * i[6]
                                + 4.0 * (1 if e_.exclusive else 2) * i[5] * i[6] for i, e_ in zip(q, arr)),
                            replay=lambda a=arr, n_=n, pr=_PRECISION, k=keep: _hip.lib().rf_wgrad_grouped(a, n_, pr, _stream()))
            for _, _, into, bias_into, *_ in q:
                self.pending.discard(into.data_ptr())
                if bias_into is not None:
                    self.pending.discard(bias_into.data_ptr())
                if SINK.on_write is not None:
                    SINK.on_write(into)
                    if bias_into is not None:
                        SINK.on_write(bias_into)
            if self._watch is not None and not (self.pending & self._watch[0]):
                cb, self._watch = self._watch[1], None
                cb(st)
        q.clear()

    def flush(self, side=False):
        for st, q in list(self.queues.values()):
            self._flush(st, q, side=side)
        self.pending.clear()


_WGRAD_GROUP_TILES = int(os.environ.get("RF_WGRAD_GROUP_TILES", "1024"))
WGRAD = _WgradQueue()


def flush_weight_grads():
    WGRAD.flush()


class _Sink:
    """Gradient sinks: when the training engine owns a flat gradient buffer it tags every trainable
    parameter with ``_rf_grad`` (its slot).  Kernels then ACCUMULATE weight / bias / gain gradients
    straight into the slot (the buffer is zeroed once per step) instead of materialising a temporary that
    autograd adds in -- one launch and one round trip less per parameter per step."""
    active = False
    on_write = None  # callback(view) after a slot has been written (DP: bucket-ready bookkeeping)


SINK = _Sink()


def _wrote(*views):
    if SINK.on_write is not None:
        for v in views:
            if v is not None and v.data_ptr() not in WGRAD.pending:  # queued writes report at flush time
                SINK.on_write(v)


def _slot(t, shape=None):
    """The flat-buffer slot of parameter ``t`` (viewed as ``shape``) or None."""
    if not SINK.active or t is None:
        return None
    g = getattr(t, "_rf_grad", None)
    if g is None:
        return None
    return g if shape is None else g.view(shape)


def _input_grad(dy2: torch.Tensor, w: torch.Tensor, **epi) -> torch.Tensor:
    """dX[M,K] = dY[M,N] W[N,K]."""
    M, N = dy2.shape
    K = w.shape[1]
    dx = torch.empty(M, K, device=dy2.device, dtype=torch.float32)
    gemm(dy2, dy2.stride(0), 1, w, w.stride(0), 1, dx, K, M, K, N, **epi)
    return dx


ROWBLOCK = os.environ.get("RF_ROWBLOCK", "1") != "0"  # row-block kernels (csrc/rowblock.hip) in bf16 mode


def _rowblock_ok(x2, w, N, K, with_ln=False) -> bool:
    return (ROWBLOCK and _PRECISION == 1 and (K == 128 or K == 256) and N % 4 == 0 and (not with_ln or N == 128) and x2.is_cuda
            and x2.data_ptr() % 16 == 0 and x2.stride(0) % 4 == 0 and x2.stride(1) == 1 and w.data_ptr() % 16 == 0
            and w.is_contiguous())


def _rowblock_linear(x2, w, b, r2, y, M, N, K, ln=None, xhat=None, rstd=None, eps=1e-5):
    ev = PROFILE.begin() if PROFILE.on else None
    args = (ptr(x2), x2.stride(0), ptr(w), ptr(b), ptr(r2), N if r2 is not None else 0, ptr(y), N, M, N, K,
            ptr(ln[0]) if ln else None, ptr(ln[1]) if ln else None, ptr(xhat), ptr(rstd), eps)
    check(_hip.lib().rf_rowblock_linear(*args, _stream()), "rf_rowblock_linear")
    if ev is not None:
        keep = (x2, w, b, r2, y, ln, xhat, rstd)
        PROFILE.end(f"rb_linear_kernel<{K}, {'true' if ln else 'false'}, {32 if M <= 2048 else 64}>", ev, 2.0 * M * N * K,
                    4.0 * (M * K + N * K + M * N * (1 + (r2 is not None) + (xhat is not None))),
                    replay=lambda a=args, k=keep: _hip.lib().rf_rowblock_linear(*a, _stream()))


def _rowblock_nn_ok(w2d, ln: bool = False) -> bool:
    """Can dX = dY w (w = forward weight (out, in), contiguous) run as the row-block NN kernel?"""
    if not (ROWBLOCK and _PRECISION == 1 and w2d.is_cuda and w2d.is_contiguous() and w2d.data_ptr() % 16 == 0):
        return False
    return bool(_hip.lib().rf_rowblock_linear_nn_supported(w2d.shape[0], w2d.shape[1], 1 if ln else 0))


def _rowblock_nn(w2d, M, *, a=None, ln=None, dpre=None, dgam=None, dbet=None, res=None, dsrc=None, dact=0):
    """-> y (M, in_features).  ``ln`` = (dy2, xhat, rstd, gamma): LayerNorm-backward prologue (writes ``dpre``)."""
    KC, NOUT = w2d.shape
    y = torch.empty(M, NOUT, device=w2d.device, dtype=torch.float32)
    ev = PROFILE.begin() if PROFILE.on else None
    args = (ptr(a), a.stride(0) if a is not None else 0, ptr(ln[0]) if ln else None, ptr(ln[1]) if ln else None,
            ptr(ln[2]) if ln else None, ptr(ln[3]) if ln else None, ptr(dpre), ptr(dgam), ptr(dbet), ptr(w2d), ptr(res),
            res.stride(0) if res is not None else 0, ptr(dsrc), dsrc.stride(0) if dsrc is not None else 0, dact, ptr(y),
            NOUT, M, KC, NOUT)
    check(_hip.lib().rf_rowblock_linear_nn(*args, _stream()), "rf_rowblock_linear_nn")
    if ev is not None:
        keep = (a, ln, dpre, dgam, dbet, w2d, res, dsrc, y)
        lnbwd = "true" if ln else "false"
        PROFILE.end(f"rb_nn_kernel<{KC}, {NOUT}, {lnbwd}, {32 if M <= 2048 else 64}>", ev, 2.0 * M * KC * NOUT,
                    4.0 * (M * KC * (3 if ln else 1) + KC * NOUT + M * NOUT * (1 + (res is not None) + (dsrc is not None))))
    return y


def _vec_rows(t, cols) -> bool:
    return t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0 and t.shape[1] == cols


class _Linear(torch.autograd.Function):
    """y = x W^T + b (+ residual rows broadcast over the batch: y[m] += residual[m % R])."""

    @staticmethod
    def forward(ctx, x, w, b, residual, gw, gb, fork=False, lazy_out=False):
        """``fork``: also return an alias of ``x`` for the layer's skip connection.  Its gradient then comes
        back into THIS node and is added in the dX GEMM epilogue -- otherwise autograd would sum the two
        gradients of ``x`` (projection path + skip path) with a separate elementwise launch."""
        _req(x, "linear.x"); _req(w, "linear.w")
        K = x.shape[-1]
        N = w.shape[0]
        x2 = x.reshape(-1, K)
        if x2.stride(1) != 1:
            x2 = x2.contiguous()
        w = w.contiguous()
        M = x2.shape[0]
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        if residual is not None:
            r2 = residual.reshape(-1, N).contiguous()
            if r2.shape[0] == M and _rowblock_ok(x2, w, N, K):
                _rowblock_linear(x2, w, b, r2, y, M, N, K)
            else:
                gemm(x2, x2.stride(0), 1, w, 1, K, y, N, M, N, K, bias=b, residual=r2, ldr=N, res_rows=r2.shape[0])
            ctx.res_rows, ctx.res_shape = r2.shape[0], residual.shape
        elif _rowblock_ok(x2, w, N, K):
            _rowblock_linear(x2, w, b, None, y, M, N, K)
        else:
            plan = _partials_plan(ptr(x2), x2.stride(0), w, M, N, K) if (lazy_out and LAZY_DX and x2.dtype == torch.float32) else None
            if plan is not None:
                # the caller's next op sums the slabs itself (circular_conv3_unfolded(lazy=True) -> bn_elu_pool): ``y`` stays
                # UNWRITTEN until that op fills it in (see LAZY)
                ws, splits = _gemm_partials(x2, w, M, N, K, plan)
                LAZY[y.data_ptr()] = (ws, splits, b, y)
                LAZY_COUNT[0] += 1
            else:
                gemm(x2, x2.stride(0), 1, w, 1, K, y, N, M, N, K, bias=b)
        ctx.save_for_backward(x2, w)
        ctx.sinks = (gw, gb)  # plain attributes: slots of a buffer other kernels also write (no version check)
        ctx.has_bias = b is not None
        ctx.xshape = x.shape
        # ``fork``: this projection is the ONLY consumer of x (it hands the skip alias back itself).  Where x is the output of the
        # distilling tail (BatchNorm -> ELU -> MaxPool) or of an add + LayerNorm node, whose backward can sum split-K slabs on
        # load, the input gradient may travel as slabs (see LAZY)
        src = _producer_name(x) if (fork and LAZY_DX and SINK.active and not DETERMINISTIC and residual is None) else ""
        ctx.lazy_dx = bool((src == "_BnEluPoolBackward" and _hip.lib().rf_bn_elu_pool_bwd_slab_ok(1, 2 * M))  # (<= 2 M input rows)
                           or src in ("_LinearAddLNSlabsBackward", "_AddLayerNormBackward"))  # (the decoder's cross-attention q)
        if fork:
            return y.view(*x.shape[:-1], N), x.view_as(x)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x2, w = ctx.saved_tensors
        gw, gb = ctx.sinks
        if dy is None:  # only the skip branch was used downstream
            return dskip, None, None, None, None, None, None, None
        dy2 = dy.reshape(-1, dy.shape[-1])
        if dy2.stride(1) != 1 or dy2.stride(0) != dy2.shape[1]:
            dy2 = dy2.contiguous()
        dx = dw = db = dres = None
        fused_bias = False
        if gw is not None:
            fused_bias = _weight_grad(dy2, x2, into=gw, bias_into=gb if ctx.has_bias else None) is True
        elif ctx.needs_input_grad[1]:
            dw = _weight_grad(dy2, x2)
        if ctx.has_bias and not fused_bias and (gb is not None or ctx.needs_input_grad[2]):
            db = colsum(dy2, into=gb)
        if ctx.needs_input_grad[0]:
            ds2 = None
            if dskip is not None:
                K_ = w.shape[1]
                ds2 = dskip.reshape(-1, K_)
                if ds2.stride(1) != 1 or ds2.stride(0) != K_:
                    ds2 = ds2.contiguous()
            if _rowblock_nn_ok(w) and _vec_rows(dy2, w.shape[0]) and (ds2 is None or ds2.data_ptr() % 16 == 0):
                dx = _rowblock_nn(w, dy2.shape[0], a=dy2, res=ds2).view(ctx.xshape)
            elif ds2 is not None:
                plan = None
                if getattr(ctx, "lazy_dx", False) and not DETERMINISTIC and w.is_contiguous() and ds2.shape[0] == dy2.shape[0]:
                    plan = _partials_plan(ptr(dy2), dy2.stride(0), w, dy2.shape[0], w.shape[1], w.shape[0], ldb_k=w.stride(0),
                                          ldb_n=1)
                if plan is not None:  # leave the slabs to the consumer of x's gradient (see LAZY)
                    ws, splits = _gemm_partials(dy2, w, dy2.shape[0], w.shape[1], w.shape[0], plan, ldb_k=w.stride(0), ldb_n=1)
                    dx = torch.empty(dy2.shape[0], w.shape[1], device=dy2.device, dtype=torch.float32)
                    LAZY[dx.data_ptr()] = (ws, splits, ds2, dx)
                    LAZY_COUNT[0] += 1
                    dx = dx.view(ctx.xshape)
                else:
                    dx = _input_grad(dy2, w, residual=ds2, ldr=ds2.shape[1], res_rows=ds2.shape[0]).view(ctx.xshape)
            else:
                plan = None
                if getattr(ctx, "lazy_plain", False) and not DETERMINISTIC and w.is_contiguous():
                    plan = _partials_plan(ptr(dy2), dy2.stride(0), w, dy2.shape[0], w.shape[1], w.shape[0], ldb_k=w.stride(0),
                                          ldb_n=1)
                if plan is not None:
                    ws, splits = _gemm_partials(dy2, w, dy2.shape[0], w.shape[1], w.shape[0], plan, ldb_k=w.stride(0), ldb_n=1)
                    dx = torch.empty(dy2.shape[0], w.shape[1], device=dy2.device, dtype=torch.float32)
                    LAZY[dx.data_ptr()] = (ws, splits, None, dx)
                    LAZY_COUNT[0] += 1
                    dx = dx.view(ctx.xshape)
                else:
                    dx = _input_grad(dy2, w).view(ctx.xshape)
        _wrote(gw, gb)
        if ctx.needs_input_grad[3]:  # residual rows are shared by M / R row blocks
            dres = colsum(dy2.view(-1, ctx.res_rows * dy2.shape[1])).view(ctx.res_shape)
        return dx, dw, db, dres, None, None, None, None


def linear(x, w, b=None, residual=None, fork: bool = False):
    """``w`` / ``b`` may be parameters (gradient sinks are picked up from them) or plain views.
    ``fork=True`` -> (y, alias of x for the skip connection), see ``_Linear.forward``."""
    return _Linear.apply(x, w, b, residual, _slot(w), _slot(b), fork)


def linear_packed(x, w, b, gw, gb, fork: bool = False):
    """Linear over a packed (non-parameter) weight view, e.g. [Wq;Wk;Wv] in the flat buffer; gradients go
    to the matching packed gradient views."""
    return _Linear.apply(x, w, b, None, gw, gb, fork)


class _FFN(torch.autograd.Function):
    """y = act(x W1^T + b1) W2^T + b2  -- the Conv1d(k=1) pair of every encoder/decoder layer.
    The activation and its derivative ride in GEMM epilogues."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act: str, g1, gb1, g2, gb2, need_grad=True, fork=False, drop=None):
        """``drop`` = (p, (site, mask) of the hidden activation, (site, mask) of the output) or None: the two
        nn.Dropout calls of cross_modal_transformer.py:298-299, applied in place right after each product."""
        _req(x, "ffn.x")
        F, D = w1.shape[0], w1.shape[1]
        w1, w2 = w1.reshape(F, D), w2.reshape(D, F)  # Conv1d(k=1) weights (out,in,1) viewed as matrices
        x2 = x.reshape(-1, D)
        if x2.stride(1) != 1:
            x2 = x2.contiguous()
        M = x2.shape[0]
        h = torch.empty(M, F, device=x.device, dtype=torch.float32)
        z = torch.empty_like(h) if (act == "gelu" and need_grad) else None  # pre-activation: backward only
        gemm(x2, x2.stride(0), 1, w1, 1, D, h, F, M, F, D, bias=b1, act=ACT[act], preact=z, ldp=F)
        if drop is not None:
            _drop_launch(h, h, drop[0], *drop[1])  # conv2 consumes (and dW2 needs) the dropped activation
        y = torch.empty(M, D, device=x.device, dtype=torch.float32)
        gemm(h, F, 1, w2, 1, F, y, D, M, D, F, bias=b2)
        if drop is not None:
            _drop_launch(y, y, drop[0], *drop[2])
        ctx.save_for_backward(x2, w1, w2, h, z if z is not None else h)
        ctx.drop = drop
        ctx.sinks = (g1, gb1, g2, gb2)
        ctx.act = act
        ctx.xshape = x.shape
        ctx.wshapes = (F, D)
        if fork:  # alias of x for the skip connection: its gradient is folded into the last dX epilogue
            return y.view(x.shape), x.view_as(x)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x2, w1, w2, h, zsrc = ctx.saved_tensors
        g1, gb1, g2, gb2 = ctx.sinks
        F, D = ctx.wshapes
        dy2 = dy.reshape(-1, D)
        if dy2.stride(1) != 1 or dy2.stride(0) != D:
            dy2 = dy2.contiguous()
        drop = ctx.drop
        if drop is not None:  # output dropout: the same mask on the incoming gradient
            dyd = torch.empty_like(dy2)
            _drop_launch(dy2, dyd, drop[0], *drop[2])
            dy2 = dyd
        # dZ = (dY W2) * act'(Z)   (relu: mask from H > 0; gelu: from the saved pre-activation)
        dw2 = _weight_grad(dy2, h, into=None if g2 is None else g2.view(D, F), bias_into=gb2)
        db2 = None if dw2 is True else colsum(dy2, into=gb2)
        dz = _input_grad(dy2, w2, dact_src=zsrc, ldd=zsrc.stride(0), dact=ACT[ctx.act])
        if drop is not None:  # hidden dropout sits between the activation and conv2: dZ = (dH * keep/(1-p)) * act'(Z)
            _drop_launch(dz, dz, drop[0], *drop[1])
        dw1 = _weight_grad(dz, x2, into=None if g1 is None else g1.view(F, D), bias_into=gb1)
        db1 = None if dw1 is True else colsum(dz, into=gb1)
        if dw1 is True or dw1 is False:
            dw1 = None
        if dw2 is True or dw2 is False:
            dw2 = None
        dx = None
        if ctx.needs_input_grad[0]:
            if dskip is not None:
                ds2 = dskip.reshape(-1, D)
                if ds2.stride(1) != 1 or ds2.stride(0) != D:
                    ds2 = ds2.contiguous()
                plan = None
                if getattr(ctx, "lazy_dx", False) and not DETERMINISTIC and dz.dtype == torch.float32 and w1.is_contiguous():
                    plan = _partials_plan(ptr(dz), dz.stride(0), w1, dz.shape[0], D, F, ldb_k=D, ldb_n=1)
                if plan is not None:  # leave the slabs to the LayerNorm backward that follows (see LAZY)
                    ws, splits = _gemm_partials(dz, w1, dz.shape[0], D, F, plan, ldb_k=D, ldb_n=1)
                    dx = torch.empty(dz.shape[0], D, device=dz.device, dtype=torch.float32)
                    LAZY[dx.data_ptr()] = (ws, splits, ds2, dx)
                    LAZY_COUNT[0] += 1
                    dx = dx.view(ctx.xshape)
                else:
                    dx = _input_grad(dz, w1, residual=ds2, ldr=D, res_rows=ds2.shape[0]).view(ctx.xshape)
            else:
                dx = _input_grad(dz, w1).view(ctx.xshape)
        _wrote(g1, gb1, g2, gb2)
        if dw1 is not None:
            dw1, dw2 = dw1.view(F, D, 1), dw2.view(D, F, 1)
        return dx, dw1, db1, dw2, db2, None, None, None, None, None, None, None, None


def ffn(x, conv1_w, conv1_b, conv2_w, conv2_b, act: str, fork: bool = False, drop_p: float = 0.0):
    """conv*_w: the Conv1d(k=1) weight parameters, shape (out, in, 1).  ``fork`` as in ``linear``.
    ``drop_p`` > 0: dropout(conv2(dropout(act(conv1 x)))) (train mode; cross_modal_transformer.py:298-299)."""
    assert conv1_w.dim() == 3 and conv2_w.dim() == 3
    drop = None
    if drop_p > 0.0:
        F_, lead = conv1_w.shape[0], tuple(x.shape[:-1])
        sites = []
        for shape in (lead + (F_,), tuple(x.shape)):  # reference call order: hidden activation, then output
            mask = RNG.take_forced(shape, x.device)
            site = RNG.next_site()
            if RNG.record is not None:
                RNG.record.append(RNG.materialise(site, shape, drop_p, x.device, mask))
            sites.append((site, mask))
        drop = (drop_p, sites[0], sites[1])
    return _FFN.apply(x, conv1_w, conv1_b, conv2_w, conv2_b, act, _slot(conv1_w), _slot(conv1_b),
                      _slot(conv2_w), _slot(conv2_b), torch.is_grad_enabled(), fork, drop)


class _AddLayerNorm(torch.autograd.Function):
    """y = LayerNorm(x + residual) (eps 1e-5); residual optional."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, eps, gg, gb, need_grad=True):
        _req(x, "layernorm.x")
        cols = x.shape[-1]
        rows = x.numel() // cols
        # a strided view (the consumed tail of a (B, L, C) activation, a row-pitched 2-D view) is read in place
        if x.dim() == 3 and x.stride(2) == 1 and x.shape[0] * x.shape[1] == rows:
            x2, seg = x, (x.shape[1], x.stride(0), x.stride(1))
        elif x.dim() == 2 and x.stride(1) == 1:
            x2, seg = x, (rows, 0, x.stride(0))
        else:
            x2, seg = x.reshape(-1, cols).contiguous(), (rows, 0, cols)
        r2 = residual.reshape(-1, cols).contiguous() if residual is not None else None
        y = torch.empty(rows, cols, device=x.device, dtype=torch.float32)
        xhat = torch.empty_like(y) if need_grad else None  # normalised input + 1/sigma: backward only
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32) if need_grad else None
        ev = PROFILE.begin() if PROFILE.on else None
        check(_hip.lib().rf_layernorm_fwd_strided(ptr(x2), seg[0], seg[1], seg[2], ptr(r2), ptr(gamma), ptr(beta), ptr(y),
                                                  ptr(xhat), ptr(rstd), rows, cols, eps, _stream()), "rf_layernorm_fwd")
        if ev is not None:
            PROFILE.end("layernorm_fwd_kernel", ev, 8.0 * rows * cols, 4.0 * rows * cols * (4 if r2 is not None else 3))
        if need_grad:
            ctx.save_for_backward(xhat, rstd, gamma)
        ctx.sinks = (gg, gb)
        ctx.has_res = residual is not None
        ctx.xshape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        xhat, rstd, gamma = ctx.saved_tensors
        gg, gb = ctx.sinks
        rows, cols = xhat.shape
        lazy = LAZY.pop(dy.data_ptr(), None)
        if lazy is not None:  # the gradient is still the split-K slabs of the FFN's last dX product (see LAZY)
            assert lazy[3].numel() == rows * cols
            dx, dg, db = _ln_backward(None, xhat, rstd, gamma, gg, gb, slabs=lazy[:3])
        else:
            dx, dg, db = _ln_backward(dy.reshape(rows, cols).contiguous(), xhat, rstd, gamma, gg, gb)
        dx = dx.view(ctx.xshape)
        return dx, (dx if ctx.has_res else None), dg, db, None, None, None, None


def _ln_backward(dy2, xhat, rstd, gamma, gg, gb, fold_L: int = 0, slabs=None):
    """LayerNorm backward on saved (xhat, rstd): -> (d pre-norm input, dgamma, dbeta); with sinks the
    parameter gradients are accumulated there and returned as None.  ``fold_L`` = L > 0: ``dy2`` is the gradient of the
    (B, L + 2, 3 cols) im2col image the forward wrote (``_ln_fwd_slabs(unfold_L=L)``); the fold happens on load.
    ``slabs`` = (workspace, splits, skip gradient or None): the incoming gradient is still the split-K slabs of the dX product
    in front of this backward (``LAZY``), summed on load (``dy2`` is ignored)."""
    rows, cols = xhat.shape
    dx = torch.empty_like(xhat)
    sink = gg is not None and gb is not None
    dg = gg if sink else torch.empty(cols, device=xhat.device, dtype=torch.float32)
    db = gb if sink else torch.empty(cols, device=xhat.device, dtype=torch.float32)
    atomic = sink and not DETERMINISTIC
    ws = None
    if not atomic:
        parts = _hip.lib().rf_layernorm_bwd_parts(rows)
        ws = torch.empty(parts * 2 * cols, device=xhat.device, dtype=torch.float32)
    ev = PROFILE.begin() if PROFILE.on else None
    if slabs is not None:
        check(_hip.lib().rf_layernorm_bwd_slabs(ptr(slabs[0]), slabs[1], ptr(slabs[2]), ptr(xhat), ptr(rstd), ptr(gamma), ptr(dx),
                                                ptr(dg), ptr(db), 2 if atomic else (1 if sink else 0), ptr(ws), rows, cols,
                                                _stream()), "rf_layernorm_bwd_slabs")
    elif fold_L:
        check(_hip.lib().rf_layernorm_bwd_fold(ptr(dy2), ptr(xhat), ptr(rstd), ptr(gamma), ptr(dx), ptr(dg), ptr(db),
                                               2 if atomic else (1 if sink else 0), ptr(ws), rows, cols, fold_L, _stream()),
              "rf_layernorm_bwd_fold")
    else:
        check(_hip.lib().rf_layernorm_bwd(ptr(dy2), ptr(xhat), ptr(rstd), ptr(gamma), ptr(dx), ptr(dg), ptr(db),
                                          2 if atomic else (1 if sink else 0), ptr(ws), rows, cols, _stream()),
              "rf_layernorm_bwd")
    if ev is not None:
        PROFILE.end("layernorm_bwd_kernel(+ln_param_reduce)", ev, 12.0 * rows * cols, 4.0 * rows * cols * 3)
    if sink:
        _wrote(gg, gb)
        dg = db = None
    return dx, dg, db


class _LinearAddLN(torch.autograd.Function):
    """y = LayerNorm(res + a W^T + b): attention out-projection + residual + norm in one row-block launch
    (cross_modal_transformer.py:302-306 / :344-352)."""

    @staticmethod
    def forward(ctx, a, w, b, res, gamma, beta, eps, gw, gb, gg, gbeta, need_grad=True):
        _req(a, "linear_ln.a")
        K, N = a.shape[-1], w.shape[0]
        a2 = a.reshape(-1, K)
        r2 = res.reshape(-1, N).contiguous()
        M = a2.shape[0]
        y = torch.empty(M, N, device=a.device, dtype=torch.float32)
        xhat = torch.empty_like(y) if need_grad else None
        rstd = torch.empty(M, device=a.device, dtype=torch.float32) if need_grad else None
        _rowblock_linear(a2, w, b, r2, y, M, N, K, ln=(gamma, beta), xhat=xhat, rstd=rstd, eps=eps)
        if need_grad:
            ctx.save_for_backward(a2, w, xhat, rstd, gamma)
        ctx.sinks = (gw, gb, gg, gbeta)
        ctx.shapes = (a.shape, res.shape)
        return y.view(res.shape)

    @staticmethod
    def backward(ctx, dy):
        a2, w, xhat, rstd, gamma = ctx.saved_tensors
        gw, gb, gg, gbeta = ctx.sinks
        M, N = xhat.shape
        dy2 = dy.reshape(M, N).contiguous()
        da = None
        sink = gg is not None and gbeta is not None and not DETERMINISTIC
        if sink and ctx.needs_input_grad[0] and _rowblock_nn_ok(w, ln=True) and dy2.data_ptr() % 16 == 0:
            # LayerNorm backward + dX of the out-projection in one launch
            dpre = torch.empty_like(xhat)
            da = _rowblock_nn(w, M, ln=(dy2, xhat, rstd, gamma), dpre=dpre, dgam=gg, dbet=gbeta).view(ctx.shapes[0])
            _wrote(gg, gbeta)
            dgam = dbet = None
        else:
            dpre, dgam, dbet = _ln_backward(dy2, xhat, rstd, gamma, gg, gbeta)
        dw = db = None
        fused_bias = False
        if gw is not None:
            fused_bias = _weight_grad(dpre, a2, into=gw, bias_into=gb) is True
        else:
            dw = _weight_grad(dpre, a2)
        if not fused_bias:
            db = colsum(dpre, into=gb)
        if da is None and ctx.needs_input_grad[0]:
            da = _input_grad(dpre, w).view(ctx.shapes[0])
        _wrote(gw, gb)
        return da, dw, db, dpre.view(ctx.shapes[1]), dgam, dbet, None, None, None, None, None, None


class _LinearAddLNSlabs(torch.autograd.Function):
    """y = LayerNorm(res + a W^T + b) for a product that needs split-K (the GPS backbone's d_model = 832 out-projections):
    the product leaves its slabs, the norm sums them -- ``add_layer_norm(res, linear(a, w, b))`` minus the slab-sum launch,
    bit-identical forward; the backward IS that composition (LayerNorm backward, then ``_Linear.backward``)."""

    @staticmethod
    def forward(ctx, a, w, b, res, gamma, beta, eps, gw, gb, gg, gbeta, plan, need_grad=True, lazy_da=False):
        _req(a, "linear_ln.a"); _req(w, "linear_ln.w")
        ctx.lazy_da = bool(lazy_da)
        K, N = a.shape[-1], w.shape[0]
        a2 = a.reshape(-1, K)
        r2 = res.reshape(-1, N).contiguous()
        M = a2.shape[0]
        ws, splits = _gemm_partials(a2, w, M, N, K, plan)
        y, xhat, rstd = _ln_fwd_slabs(ws, splits, b, r2, gamma, beta, M, N, eps, need_grad)
        if need_grad:
            ctx.save_for_backward(a2, w, xhat, rstd, gamma)
        ctx.sinks = (gw, gb, gg, gbeta)
        ctx.has_bias = b is not None
        ctx.shapes = (a.shape, res.shape)
        return y.view(res.shape)

    @staticmethod
    def backward(ctx, dy):
        a2, w, xhat, rstd, gamma = ctx.saved_tensors
        gw, gb, gg, gbeta = ctx.sinks
        M, N = xhat.shape
        lazy = LAZY.pop(dy.data_ptr(), None)
        if lazy is not None:  # the gradient is still the slabs of the FFN's last dX product
            assert lazy[3].numel() == M * N
            dpre, dgam, dbet = _ln_backward(None, xhat, rstd, gamma, gg, gbeta, slabs=lazy[:3])
        else:
            dpre, dgam, dbet = _ln_backward(dy.reshape(M, N).contiguous(), xhat, rstd, gamma, gg, gbeta)
        lin = _Ctx()
        lin.saved_tensors, lin.sinks, lin.has_bias, lin.xshape = (a2, w), (gw, gb), ctx.has_bias, ctx.shapes[0]
        lin.needs_input_grad = (ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2], False)
        lin.lazy_plain = ctx.lazy_da  # (the attention backward that consumes d a sums split-K slabs on load: see LAZY)
        da, dw, db = _Linear.backward(lin, dpre)[:3]
        return da, dw, db, dpre.view(ctx.shapes[1]), dgam, dbet, None, None, None, None, None, None, None, None


def linear_add_layer_norm(a, w, b, res, gamma, beta, eps: float = 1e-5, sole_consumer: bool = False):
    """LayerNorm(res + linear(a, w, b)); one launch when the row-block kernel applies.  ``sole_consumer``: nothing else reads
    ``a`` -- when it is an attention context reached through views only, its gradient may travel as split-K slabs (LAZY)."""
    a2 = a.reshape(-1, a.shape[-1])
    if b is not None and a2.stride(1) == 1 and _rowblock_ok(a2, w, w.shape[0], a.shape[-1], with_ln=True):
        return _LinearAddLN.apply(a, w, b, res, gamma, beta, eps, _slot(w), _slot(b), _slot(gamma), _slot(beta),
                                  torch.is_grad_enabled())
    if a2.stride(1) == 1 and w.is_contiguous() and a.dtype == torch.float32 and res.shape[-1] == w.shape[0]:
        plan = _partials_plan(ptr(a2), a2.stride(0), w, a2.shape[0], w.shape[0], a.shape[-1])
        if plan is not None and res.numel() == a2.shape[0] * w.shape[0]:
            lazy_da = bool(sole_consumer and LAZY_DX and LAZY_ATTN and SINK.active and not DETERMINISTIC and torch.is_grad_enabled()
                           and _producer_name(a) == "_AttentionBackward")
            return _LinearAddLNSlabs.apply(a, w, b, res, gamma, beta, eps, _slot(w), _slot(b), _slot(gamma), _slot(beta),
                                           plan, torch.is_grad_enabled(), lazy_da)
    return add_layer_norm(res, linear(a, w, b), gamma, beta, eps)


class _FFNAddLN(torch.autograd.Function):
    """y = LayerNorm(x + conv2(act(conv1(x)))) for d_model 128 / d_ff 256: one launch forward; backward =
    LN backward, then the FFN chain with the residual gradient folded into the last dX epilogue."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act, gamma, beta, eps, g1, gb1, g2, gb2, gg, gbeta, need_grad=True):
        _req(x, "ffn_ln.x")
        F, D = w1.shape[0], w1.shape[1]
        w1, w2 = w1.reshape(F, D), w2.reshape(D, F)
        x2 = x.reshape(-1, D).contiguous()
        M = x2.shape[0]
        y = torch.empty_like(x2)
        h = torch.empty(M, F, device=x.device, dtype=torch.float32) if need_grad else None
        z = torch.empty_like(h) if (need_grad and act == "gelu") else None
        xhat = torch.empty_like(x2) if need_grad else None
        rstd = torch.empty(M, device=x.device, dtype=torch.float32) if need_grad else None
        ev = PROFILE.begin() if PROFILE.on else None
        args = (ptr(x2), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(h), ptr(z), ptr(y), M, D, F, ACT[act], ptr(gamma),
                ptr(beta), ptr(xhat), ptr(rstd), eps)
        check(_hip.lib().rf_rowblock_ffn_ln(*args, _stream()), "rf_rowblock_ffn_ln")
        if ev is not None:
            keep = (x2, w1, b1, w2, b2, h, z, y, gamma, beta, xhat, rstd)
            PROFILE.end(f"rb_ffn_ln_kernel<{32 if M <= 2048 else 64}>", ev, 4.0 * M * D * F,
                        4.0 * (M * D * (2 + (xhat is not None)) + 2 * D * F + M * F * ((h is not None) + (z is not None))),
                        replay=lambda a=args, k=keep: _hip.lib().rf_rowblock_ffn_ln(*a, _stream()))
        if need_grad:
            ctx.save_for_backward(x2, w1, w2, h, z if z is not None else h, xhat, rstd, gamma)
        ctx.sinks = (g1, gb1, g2, gb2, gg, gbeta)
        ctx.act = act
        ctx.xshape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, w1, w2, h, zsrc, xhat, rstd, gamma = ctx.saved_tensors
        g1, gb1, g2, gb2, gg, gbeta = ctx.sinks
        M, D = x2.shape
        F = w1.shape[0]
        dy2 = dy.reshape(M, D).contiguous()
        sink = gg is not None and gbeta is not None and not DETERMINISTIC
        fused = sink and _rowblock_nn_ok(w2, ln=True) and _rowblock_nn_ok(w1) and dy2.data_ptr() % 16 == 0
        if fused:  # LayerNorm backward + (dPre W2) * act'(z) in one launch
            dpre = torch.empty_like(xhat)
            dz = _rowblock_nn(w2, M, ln=(dy2, xhat, rstd, gamma), dpre=dpre, dgam=gg, dbet=gbeta, dsrc=zsrc,
                              dact=ACT[ctx.act])
            _wrote(gg, gbeta)
            dgam = dbet = None
        else:
            dpre, dgam, dbet = _ln_backward(dy2, xhat, rstd, gamma, gg, gbeta)
            dz = _input_grad(dpre, w2, dact_src=zsrc, ldd=zsrc.stride(0), dact=ACT[ctx.act])
        dw2 = _weight_grad(dpre, h, into=None if g2 is None else g2.view(D, F), bias_into=gb2)
        db2 = None if dw2 is True else colsum(dpre, into=gb2)
        dw1 = _weight_grad(dz, x2, into=None if g1 is None else g1.view(F, D), bias_into=gb1)
        db1 = None if dw1 is True else colsum(dz, into=gb1)
        if dw1 is True or dw1 is False:
            dw1 = None
        if dw2 is True or dw2 is False:
            dw2 = None
        # dX = dZ W1 + d(pre-norm)  -- the residual branch rides in the GEMM epilogue, no separate add
        if fused:
            dx = _rowblock_nn(w1, M, a=dz, res=dpre).view(ctx.xshape)
        else:
            dx = _input_grad(dz, w1, residual=dpre, ldr=D, res_rows=M).view(ctx.xshape)
        _wrote(g1, gb1, g2, gb2)
        if dw1 is not None:
            dw1, dw2 = dw1.view(F, D, 1), dw2.view(D, F, 1)
        return dx, dw1, db1, dw2, db2, None, dgam, dbet, None, None, None, None, None, None, None, None


class _FFNAddLNSlabs(torch.autograd.Function):
    """y = LayerNorm(x + conv2(act(conv1 x))) where conv2 needs split-K (the GPS backbone's d_ff = 3328 -> 832): conv2
    leaves its slabs and the norm sums them -- ``add_layer_norm(skip, ffn(x, fork=True))`` minus the slab-sum launch,
    bit-identical forward; the backward IS that composition (LayerNorm backward, then ``_FFN.backward``)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act, gamma, beta, eps, g1, gb1, g2, gb2, gg, gbeta, plan, need_grad=True,
                unfold_L: int = 0, lazy_dx: bool = False):
        """``unfold_L`` = L: the output is the (B, L + 2, 3 D) im2col image of the distilling convolution that follows
        (``circular_conv3(pad=2)`` without its unfold launch; the backward folds on load)."""
        _req(x, "ffn_ln.x")
        F, D = w1.shape[0], w1.shape[1]
        w1, w2 = w1.reshape(F, D), w2.reshape(D, F)
        x2 = x.reshape(-1, D)
        M = x2.shape[0]
        h = torch.empty(M, F, device=x.device, dtype=torch.float32)
        z = torch.empty_like(h) if (act == "gelu" and need_grad) else None
        gemm(x2, x2.stride(0), 1, w1, 1, D, h, F, M, F, D, bias=b1, act=ACT[act], preact=z, ldp=F)
        ws, splits = _gemm_partials(h, w2, M, D, F, plan)
        r2 = x2 if x2.is_contiguous() else x2.contiguous()
        y, xhat, rstd = _ln_fwd_slabs(ws, splits, b2, r2, gamma, beta, M, D, eps, need_grad, unfold_L)
        if need_grad:
            ctx.save_for_backward(x2, w1, w2, h, z if z is not None else h, xhat, rstd, gamma)
        ctx.sinks = (g1, gb1, g2, gb2, gg, gbeta)
        ctx.act, ctx.xshape, ctx.wshapes, ctx.unfold_L, ctx.lazy_dx = act, x.shape, (F, D), unfold_L, bool(lazy_dx)
        return y if unfold_L else y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, w1, w2, h, zsrc, xhat, rstd, gamma = ctx.saved_tensors
        g1, gb1, g2, gb2, gg, gbeta = ctx.sinks
        M, D = xhat.shape
        if ctx.unfold_L:
            dpre, dgam, dbet = _ln_backward(dy.contiguous(), xhat, rstd, gamma, gg, gbeta, fold_L=ctx.unfold_L)
        else:
            dpre, dgam, dbet = _ln_backward(dy.reshape(M, D).contiguous(), xhat, rstd, gamma, gg, gbeta)
        f = _Ctx()
        f.saved_tensors, f.sinks, f.drop = (x2, w1, w2, h, zsrc), (g1, gb1, g2, gb2), None
        f.act, f.xshape, f.wshapes, f.needs_input_grad = ctx.act, ctx.xshape, ctx.wshapes, (ctx.needs_input_grad[0],)
        f.lazy_dx = ctx.lazy_dx
        dpre_v = dpre.view(ctx.xshape)
        dx, dw1, db1, dw2, db2 = _FFN.backward(f, dpre_v, dpre_v)[:5]  # the skip branch's gradient rides in the last dX epilogue
        return dx, dw1, db1, dw2, db2, None, dgam, dbet, None, None, None, None, None, None, None, None, None, None, None


UNFOLD_IN_NORM = os.environ.get("RF_UNFOLD_IN_NORM", "1") != "0"  # measurement switch


def _producer_name(x) -> str:
    """Class name of the autograd node that produced ``x``, looking through reshape / view nodes (their backward only
    re-views the gradient: a placeholder passes through untouched)."""
    fn = x.grad_fn
    for _ in range(4):
        if fn is None:
            return ""
        name = type(fn).__name__
        if name in ("ViewBackward0", "UnsafeViewBackward0", "ReshapeAliasBackward0") and fn.next_functions:
            fn = fn.next_functions[0][0]
            continue
        return name
    return ""


def _lazy_dx_ok(x, sole_consumer: bool) -> bool:
    """May the FFN hand its input gradient to ``x``'s producer as slabs (``LAZY``)?  Only when the caller vouches that nothing
    else consumes ``x`` and ``x`` came out of one of the two norms whose backward looks the slabs up."""
    return bool(sole_consumer and LAZY_DX and SINK.active and not DETERMINISTIC and torch.is_grad_enabled()
                and x.grad_fn is not None and type(x.grad_fn).__name__ in ("_LinearAddLNSlabsBackward", "_AddLayerNormBackward"))


def ffn_add_layer_norm(x, conv1_w, conv1_b, conv2_w, conv2_b, act: str, gamma, beta, eps: float = 1e-5, unfold: bool = False,
                       sole_consumer: bool = False):
    """LayerNorm(x + ffn(x)); one launch when d_model = 128, d_ff = 256 in bf16 mode.
    ``unfold``: the caller is followed by a distilling convolution (circular k = 3, padding 2) and can take the output as
    its im2col image: -> (tensor, True) when the slab-summing norm wrote that image, else (plain output, False)."""
    F, D = conv1_w.shape[0], conv1_w.shape[1]
    if unfold:
        if (UNFOLD_IN_NORM and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.is_contiguous() and D > 256
                and (3 * D) % 4 == 0 and conv1_w.is_contiguous() and conv2_w.is_contiguous() and x.shape[1] >= 2):
            M = x.numel() // D
            plan = _partials_plan(ptr(x) // 256 * 256, F, conv2_w.view(D, F), M, D, F)
            if plan is not None:
                y = _FFNAddLNSlabs.apply(x, conv1_w, conv1_b, conv2_w, conv2_b, act, gamma, beta, eps, _slot(conv1_w),
                                         _slot(conv1_b), _slot(conv2_w), _slot(conv2_b), _slot(gamma), _slot(beta), plan,
                                         torch.is_grad_enabled(), x.shape[1], _lazy_dx_ok(x, sole_consumer))
                return y, True
        return ffn_add_layer_norm(x, conv1_w, conv1_b, conv2_w, conv2_b, act, gamma, beta, eps, sole_consumer=sole_consumer), False
    if (ROWBLOCK and _PRECISION == 1 and D == 128 and F == 256 and x.is_cuda and conv1_b is not None
            and conv2_b is not None and conv1_w.is_contiguous() and conv2_w.is_contiguous()
            and conv1_w.data_ptr() % 16 == 0 and conv2_w.data_ptr() % 16 == 0):
        return _FFNAddLN.apply(x, conv1_w, conv1_b, conv2_w, conv2_b, act, gamma, beta, eps, _slot(conv1_w),
                               _slot(conv1_b), _slot(conv2_w), _slot(conv2_b), _slot(gamma), _slot(beta),
                               torch.is_grad_enabled())
    if x.is_cuda and x.dtype == torch.float32 and conv1_w.is_contiguous() and conv2_w.is_contiguous():
        M = x.numel() // D
        # conv2's left operand is the hidden activation: a fresh contiguous (M, F) tensor, allocator-aligned like x itself
        plan = _partials_plan(ptr(x) // 256 * 256, F, conv2_w.view(D, F), M, D, F) if x.is_contiguous() else None
        if plan is not None:
            return _FFNAddLNSlabs.apply(x, conv1_w, conv1_b, conv2_w, conv2_b, act, gamma, beta, eps, _slot(conv1_w),
                                        _slot(conv1_b), _slot(conv2_w), _slot(conv2_b), _slot(gamma), _slot(beta), plan,
                                        torch.is_grad_enabled(), 0, _lazy_dx_ok(x, sole_consumer))
    y, skip = ffn(x, conv1_w, conv1_b, conv2_w, conv2_b, act, fork=True)
    return add_layer_norm(skip, y, gamma, beta, eps)


def add_layer_norm(x, residual, gamma, beta, eps: float = 1e-5):
    return _AddLayerNorm.apply(x, residual, gamma, beta, eps, _slot(gamma), _slot(beta), torch.is_grad_enabled())


class _Unfold3(torch.autograd.Function):
    """(B,L,C) -> (B,L+2p-2,ld) circular im2col for the k=3 sequence convolutions (ld >= 3C, extra columns zero)."""

    @staticmethod
    def forward(ctx, x, pad: int, ld: int = 0):
        _req(x, "unfold3.x")
        x = x.contiguous()
        B, L, C = x.shape
        ld = ld or 3 * C
        cols = torch.empty(B, L + 2 * pad - 2, ld, device=x.device, dtype=torch.float32)
        check(_hip.lib().rf_unfold3_circular_ld(ptr(x), ptr(cols), B, L, C, pad, ld, _stream()), "rf_unfold3")
        ctx.dims = (B, L, C, pad, ld)
        return cols

    @staticmethod
    def backward(ctx, dcols):
        B, L, C, pad, ld = ctx.dims
        dcols = dcols.contiguous()
        dx = torch.empty(B, L, C, device=dcols.device, dtype=torch.float32)
        check(_hip.lib().rf_fold3_circular_ld(ptr(dcols), ptr(dx), B, L, C, pad, ld, _stream()), "rf_fold3")
        return dx, None, None


class _PadCols(torch.autograd.Function):
    """(rows, cols) -> (rows, ld) with zero columns appended, one launch (was a zero-fill + a strided copy); the gradient goes
    back in one launch too -- straight into the weight's slot of the flat gradient buffer when there is one (was a
    slice copy + autograd's accumulation)."""

    @staticmethod
    def forward(ctx, w2, ld: int, gw):
        _req(w2, "pad_cols.w")
        w2 = w2.contiguous()
        rows, cols = w2.shape
        out = torch.empty(rows, ld, device=w2.device, dtype=torch.float32)
        check(_hip.lib().rf_pad_cols(ptr(w2), ptr(out), rows, cols, ld, _stream()), "rf_pad_cols")
        ctx.dims, ctx.gw = (rows, cols, ld), gw
        return out

    @staticmethod
    def backward(ctx, dwp):
        rows, cols, ld = ctx.dims
        dwp = dwp.contiguous()
        gw = ctx.gw
        if gw is not None and gw.is_contiguous():
            check(_hip.lib().rf_unpad_cols(ptr(dwp), ptr(gw), rows, cols, ld, 1, _stream()), "rf_unpad_cols")
            _wrote(gw)
            return None, None, None
        dw = torch.empty(rows, cols, device=dwp.device, dtype=torch.float32)
        check(_hip.lib().rf_unpad_cols(ptr(dwp), ptr(dw), rows, cols, ld, 0, _stream()), "rf_unpad_cols")
        return dw, None, None


def circular_conv3_unfolded(cols, weight, bias=None, lazy: bool = False):
    """The product of ``circular_conv3(pad=2)`` on an im2col image that already exists (ffn_add_layer_norm(unfold=True)).
    ``lazy``: the caller passes the result STRAIGHT to ``bn_elu_pool`` in train mode, which sums the product's split-K slabs
    itself (and writes the finished map): the tensor returned here is unwritten until then."""
    d = weight.shape[0]
    return _Linear.apply(cols, weight.view(d, -1), bias, None, _slot(weight, (d, weight.shape[1] * 3)), _slot(bias), False,
                         bool(lazy))


def circular_conv3(x, weight, bias=None, pad: int = 1, residual=None):
    """Conv1d(k=3, padding_mode='circular') on channels-last sequences: weight (d, c, 3), used in place as
    a (d, 3c) matrix (the unfold emits columns in the weight's own (c, t) memory order).
    ``residual`` (L_out, d): added to every sequence (positional / time-feature table)."""
    d, c = weight.shape[0], weight.shape[1]
    if (3 * c) % 4 != 0 and x.is_cuda:
        # 3 c not a multiple of 4 (the GPS backbone's 69 input channels -> K = 207): neither operand of the embedding GEMM
        # is 16-B addressable and it (and both backward GEMMs) would take the scalar tile kernel -- 37 us for 0.1 GFLOP on
        # the critical path.  One zero column: the unfold writes a pitch of 208, the weight is padded to match (autograd
        # slices its gradient back), all three GEMMs stay on the vector path.
        ld = (3 * c + 3) // 4 * 4
        cols = _Unfold3.apply(x, pad, ld)
        wp = _PadCols.apply(weight.reshape(d, 3 * c), ld, _slot(weight, (d, 3 * c)))
        return _Linear.apply(cols, wp, bias, residual, None, _slot(bias))
    cols = _Unfold3.apply(x, pad)
    return _Linear.apply(cols, weight.view(d, -1), bias, residual, _slot(weight, (d, weight.shape[1] * 3)),
                         _slot(bias))


class _BnEluPool(torch.autograd.Function):
    """BatchNorm1d -> ELU -> MaxPool1d(3,2,1) over (B,L,C) (Informer distilling tail)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mean, var, eps, training, gg=None, gb=None, running=None):
        """``running`` = (running_mean, running_var, num_batches_tracked, momentum) with ``mean`` / ``var`` empty buffers:
        train mode, statistics computed by the same launch (rf_bn_train_elu_pool_fwd)."""
        x = x.contiguous()
        B, L, C = x.shape
        Lout = (L - 1) // 2 + 1
        y = torch.empty(B, Lout, C, device=x.device, dtype=torch.float32)
        arg = torch.empty(B, Lout, C, device=x.device, dtype=torch.int32)
        lazy = LAZY.pop(x.data_ptr(), None)
        if lazy is not None:  # x is still the split-K slabs of the convolution's product: summed on load, x written here
            assert running is not None and lazy[3].numel() == x.numel(), "a slab-carried map reached the wrong consumer"
            rm, rv, nbt, momentum = running
            check(_hip.lib().rf_bn_train_elu_pool_fwd_slabs(ptr(lazy[0]), lazy[1], ptr(lazy[2]), ptr(x), ptr(gamma), ptr(beta),
                                                            ptr(mean), ptr(var), ptr(rm), ptr(rv), ptr(nbt), momentum, ptr(y),
                                                            ptr(arg), B, L, C, eps, _stream()), "rf_bn_train_elu_pool_fwd_slabs")
        elif running is not None:
            rm, rv, nbt, momentum = running
            check(_hip.lib().rf_bn_train_elu_pool_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(mean), ptr(var), ptr(rm), ptr(rv),
                                                      ptr(nbt), momentum, ptr(y), ptr(arg), B, L, C, eps, _stream()),
                  "rf_bn_train_elu_pool_fwd")
        else:
            check(_hip.lib().rf_bn_elu_pool_fwd(ptr(x), ptr(mean), ptr(var), ptr(gamma), ptr(beta), ptr(y),
                                                ptr(arg), B, L, C, eps, _stream()), "rf_bn_elu_pool_fwd")
        ctx.save_for_backward(x, gamma, beta, mean, var, arg)
        ctx.eps, ctx.training = eps, training
        ctx.sinks = (gg, gb)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, var, arg = ctx.saved_tensors
        gg, gb = ctx.sinks
        B, L, C = x.shape
        lazy = LAZY.pop(dy.data_ptr(), None)
        dx = torch.empty_like(x)
        sink = gg is not None and gb is not None
        dg = gg if sink else torch.empty(C, device=x.device, dtype=torch.float32)
        db = gb if sink else torch.empty(C, device=x.device, dtype=torch.float32)
        if lazy is not None:  # the gradient is still the split-K slabs of the next layer's q | k | v input gradient (see LAZY)
            assert lazy[3].numel() == arg.numel()
            check(_hip.lib().rf_bn_elu_pool_bwd_slabs(ptr(lazy[0]), lazy[1], ptr(lazy[2]), ptr(arg), ptr(x), ptr(mean), ptr(var),
                                                      ptr(gamma), ptr(beta), ptr(dx), ptr(dg), ptr(db), 1 if sink else 0, B, L, C,
                                                      ctx.eps, 1 if ctx.training else 0, _stream()), "rf_bn_elu_pool_bwd_slabs")
        else:
            dy = dy.contiguous()
            check(_hip.lib().rf_bn_elu_pool_bwd(ptr(dy), ptr(arg), ptr(x), ptr(mean), ptr(var), ptr(gamma),
                                                ptr(beta), ptr(dx), ptr(dg), ptr(db), 1 if sink else 0, B, L, C, ctx.eps,
                                                1 if ctx.training else 0, _stream()), "rf_bn_elu_pool_bwd")
        if sink:
            _wrote(gg, gb)
            dg = db = None
        return dx, dg, db, None, None, None, None, None, None, None


def bn_stats(x3: torch.Tensor, running_mean=None, running_var=None, num_batches_tracked=None, momentum: float = 0.1):
    """Batch mean / biased variance per channel; with running buffers the nn.BatchNorm1d running-statistics
    update (momentum, unbiased variance, batch counter) happens in the same launch."""
    B, L, C = x3.shape
    mean = torch.empty(C, device=x3.device, dtype=torch.float32)
    var = torch.empty(C, device=x3.device, dtype=torch.float32)
    if num_batches_tracked is not None:
        assert num_batches_tracked.dtype == torch.int64 and num_batches_tracked.is_cuda
    check(_hip.lib().rf_bn_stats(ptr(x3), ptr(mean), ptr(var), B * L, C, ptr(running_mean), ptr(running_var),
                                 ptr(num_batches_tracked), momentum, _stream()), "rf_bn_stats")
    return mean, var


def bn_elu_pool(x, gamma, beta, running_mean, running_var, num_batches_tracked, training: bool,
                momentum: float = 0.1, eps: float = 1e-5):
    x = x.contiguous()
    B, L, C = x.shape
    if training and B * L * 32 * 4 <= 96 * 1024:  # statistics + apply in one launch (a 32-channel slab of all rows in LDS)
        if num_batches_tracked is not None:
            assert num_batches_tracked.dtype == torch.int64 and num_batches_tracked.is_cuda
        mean = torch.empty(C, device=x.device, dtype=torch.float32)
        var = torch.empty(C, device=x.device, dtype=torch.float32)
        return _BnEluPool.apply(x, gamma, beta, mean, var, eps, training, _slot(gamma), _slot(beta),
                                (running_mean, running_var, num_batches_tracked, momentum))
    if training:
        mean, var = bn_stats(x.detach(), running_mean, running_var, num_batches_tracked, momentum)
    else:
        mean, var = running_mean, running_var
    return _BnEluPool.apply(x, gamma, beta, mean, var, eps, training, _slot(gamma), _slot(beta))


class TopSelection:
    """Debug / test hook around the ProbSparse top-u selection (discontinuous in its inputs).
    ``record``: list collecting the (B,H,u) ascending selections each call made.
    ``forced``: list of selections to impose, consumed in call order (teacher forcing)."""
    record: Optional[list] = None
    forced: Optional[list] = None
    # ``shadow`` (with ``forced``): list receiving, call by call, the selection the kernel WOULD have made on the same
    # (teacher-forced) inputs -- the call runs twice, free first (outputs discarded), then with the imposed selection.
    # Flip-rate diagnostics (tools/flip_rate.py, bench.py ade_vs_cpu_ref): every call is judged on inputs that are
    # still on the oracle's trajectory.  For a fused stack the free pass covers all its layers at once.
    shadow: Optional[list] = None

    def merge_forced(self, n_calls: int, n_layers: int):
        """``n_calls`` reference encoder calls (each ``n_layers`` selections) run as one batched call:
        turn their queued selections [call][layer] into per-layer batch-concatenated ones."""
        if self.forced is None or n_calls == 1:
            return
        head, rest = self.forced[: n_calls * n_layers], self.forced[n_calls * n_layers:]
        self.forced = [torch.cat([head[c * n_layers + l] for c in range(n_calls)], dim=0)
                       for l in range(n_layers)] + rest

    def swap_blocks(self, n_first: int, n_second: int, *, before: bool):
        """The host CALLS a block of ``n_second`` ProbSparse layers (the gaze-token encoder on its side stream) ahead of
        the ``n_first`` layers that precede it in the reference's order (the camera streams' frame encoders).
        ``before=True`` (ahead of the calls): queue the imposed selections in call order; ``before=False`` (after both
        blocks ran): put what was recorded back into reference order.  Test hooks only."""
        if before:
            if self.forced is not None:
                f = self.forced
                self.forced = f[n_first:n_first + n_second] + f[:n_first] + f[n_first + n_second:]
            return
        for lst in (self.record, self.shadow):
            if lst is not None and len(lst) >= n_first + n_second:
                tail = lst[-(n_first + n_second):]
                del lst[-(n_first + n_second):]
                lst.extend(tail[n_second:] + tail[:n_second])

    def split_record(self, n_calls: int, n_layers: int):
        """Inverse bookkeeping for ``record``: re-emit a batched call's selections in reference order."""
        for lst in (self.record, self.shadow):
            if lst is None or n_calls == 1 or len(lst) < n_layers:
                continue
            tail = lst[-n_layers:]
            del lst[-n_layers:]
            for c in range(n_calls):
                for l in range(n_layers):
                    lst.append(tail[l].chunk(n_calls, dim=0)[c])


TOPS = TopSelection()


def prob_sizes(L_Q: int, L_K: int, factor: int):
    """(sample_k, n_top) = (min(c*ceil(ln L_K), L_K), min(c*ceil(ln L_Q), L_Q))."""
    U = factor * int(math.ceil(math.log(L_K)))
    u = factor * int(math.ceil(math.log(L_Q)))
    return (U if U < L_K else L_K), (u if u < L_Q else L_Q)


class _Attention(torch.autograd.Function):
    """Attention core on projected row-major matrices.  ``a`` holds Q at column ``q_off``; ``b`` holds
    K and V at ``k_off`` / ``v_off`` (``a is b`` for a packed self-attention QKV projection).  Gradients
    come back in the same packed layouts, so the projection backward is one GEMM per packed matrix.
    mode 0 full, 1 ProbSparse, 2 ProbSparse masked."""

    @staticmethod
    def forward(ctx, a, b, offs, index_sample, dims, mode, n_top, out_layout, scale, forced_top, idx_group=0, drop=None):
        """``drop`` = (p, site, mask) of the dropout on the softmax probabilities (FullAttention) or None."""
        B, H, LQ, LK, E = dims
        q_off, k_off, v_off = offs
        _req(a, "attention.q")
        _req(b, "attention.kv")
        assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
        assert a.shape[0] == B * LQ and b.shape[0] == B * LK
        shape = (B, LQ, H, E) if out_layout == 0 else (B, H, LQ, E)
        out = torch.empty(shape, device=a.device, dtype=torch.float32)
        top, sample_k = None, 0
        if mode != 0:
            if forced_top is None and TOPS.forced is not None:
                forced_top = TOPS.forced.pop(0).to(device=a.device, dtype=torch.int32).contiguous()
                assert tuple(forced_top.shape) == (B, H, n_top), (tuple(forced_top.shape), (B, H, n_top))
                if TOPS.shadow is not None and index_sample is not None:
                    free = torch.empty(B, H, n_top, device=a.device, dtype=torch.int32)
                    check(_hip.lib().rf_attn_fwd(
                        a.data_ptr() + 4 * q_off, b.data_ptr() + 4 * k_off, b.data_ptr() + 4 * v_off, a.stride(0), b.stride(0),
                        b.stride(0), ptr(out), out_layout, ptr(index_sample), idx_group,
                        (index_sample.stride(0) if index_sample.dim() == 3 else 0), ptr(free), 0, B, H, LQ, LK, E,
                        index_sample.shape[-1], n_top, mode, scale, _stream()), "rf_attn_fwd(shadow)")
                    TOPS.shadow.append(free)
            top = forced_top if forced_top is not None else \
                torch.empty(B, H, n_top, device=a.device, dtype=torch.int32)
            sample_k = index_sample.shape[-1] if index_sample is not None else 0
        ev = PROFILE.begin() if PROFILE.on else None
        fargs = (a.data_ptr() + 4 * q_off, b.data_ptr() + 4 * k_off, b.data_ptr() + 4 * v_off, a.stride(0),
                 b.stride(0), b.stride(0), ptr(out), out_layout, ptr(index_sample), idx_group,
                 (index_sample.stride(0) if (index_sample is not None and index_sample.dim() == 3) else 0), ptr(top),
                 1 if forced_top is not None else 0, B, H, LQ, LK, E, sample_k, n_top, mode, scale)
        if drop is not None:
            dargs = (drop[0], ptr(RNG.state(a.device)), drop[1], ptr(drop[2]))
            check(_hip.lib().rf_attn_fwd_drop(*fargs, *dargs, _stream()), "rf_attn_fwd_drop")
            ev = None
        else:
            check(_hip.lib().rf_attn_fwd(*fargs, _stream()), "rf_attn_fwd")
        if ev is not None:
            u = LQ if mode == 0 else n_top  # sample stage + active rows (QK^T and AV)
            keep = (a, b, out, index_sample, top)
            full = _hip.lib().rf_attn_fwd_full_scores(B, H, LQ, LK, E, sample_k, n_top, mode)
            PROFILE.end("attn_fwd_kernel<true, true>" if full else "attn_fwd_kernel<true, false>", ev, B * H * (2.0 * LQ * sample_k * E + 4.0 * u * LK * E),
                        4.0 * B * H * E * (2 * LQ + 2 * LK) + 4.0 * LQ * sample_k,
                        replay=lambda fa=fargs, k=keep: _hip.lib().rf_attn_fwd(*fa, _stream()))
        if top is not None and TOPS.record is not None:
            TOPS.record.append(top.clone())
        ctx.save_for_backward(a, b, top if top is not None else a)
        ctx.cfg = (dims, offs, mode, n_top, out_layout, scale, a.data_ptr() == b.data_ptr())
        ctx.drop = drop
        return out

    @staticmethod
    def backward(ctx, dout):
        a, b, top = ctx.saved_tensors
        (B, H, LQ, LK, E), (q_off, k_off, v_off), mode, n_top, out_layout, scale, same = ctx.cfg
        lazy = LAZY.pop(dout.data_ptr(), None)
        if lazy is not None and ctx.drop is not None:  # (probability dropout: the plain entry point; sum the slabs here)
            dout = lazy[0].view(lazy[1], -1).sum(0).view(dout.shape)
            lazy = None
        dout = dout.contiguous()
        da = torch.empty(a.shape, device=dout.device, dtype=torch.float32)
        db = da if same else torch.empty(b.shape, device=dout.device, dtype=torch.float32)
        ev = PROFILE.begin() if PROFILE.on else None
        bargs = (a.data_ptr() + 4 * q_off, b.data_ptr() + 4 * k_off,
                 b.data_ptr() + 4 * v_off, a.stride(0), b.stride(0), b.stride(0),
                 ptr(dout), out_layout, ptr(top) if mode != 0 else None,
                 da.data_ptr() + 4 * q_off, db.data_ptr() + 4 * k_off,
                 db.data_ptr() + 4 * v_off, da.stride(0), db.stride(0), db.stride(0),
                 B, H, LQ, LK, E, n_top, mode, scale)
        if ctx.drop is not None:
            drop = ctx.drop
            dargs = (drop[0], ptr(RNG.state(a.device)), drop[1], ptr(drop[2]))
            check(_hip.lib().rf_attn_bwd_drop(*bargs, *dargs, _stream()), "rf_attn_bwd_drop")
            ev = None
        elif lazy is not None:  # d ctx is still the split-K slabs of the out-projection's input gradient (see LAZY)
            assert lazy[3].numel() == B * H * LQ * E
            sargs = bargs[:6] + (ptr(lazy[0]), lazy[1], B * H * LQ * E) + bargs[7:]
            check(_hip.lib().rf_attn_bwd_slabs(*sargs, _stream()), "rf_attn_bwd_slabs")
            ev = None
        else:
            check(_hip.lib().rf_attn_bwd(*bargs, _stream()), "rf_attn_bwd")
        if ev is not None:
            u = LQ if mode == 0 else n_top
            keep = (a, b, dout, top, da, db)  # operands stay alive for the replay
            PROFILE.end("attn_bwd_kernel<true>", ev, B * H * 10.0 * u * LK * E, 4.0 * B * H * E * (4 * LQ + 4 * LK),
                        replay=lambda fa=bargs, k=keep: _hip.lib().rf_attn_bwd(*fa, _stream()))
        return da, (None if same else db), None, None, None, None, None, None, None, None, None, None


def attention(a, b, offs, dims, mode: int, *, index_sample=None, n_top: int = 0, out_layout: int = 0,
              scale: Optional[float] = None, forced_top=None, idx_group: int = 0, drop_p: float = 0.0):
    """Returns ctx in (B,LQ,H,E) [out_layout 0] or (B,H,LQ,E) [out_layout 1: Informer's un-transposed
    layout, layers/SelfAttentionFamily.py:165].  Every column of ``a`` / ``b`` must be one of Q/K/V."""
    B, H, LQ, LK, E = dims
    scale = scale or 1.0 / math.sqrt(E)
    if a is b:
        assert a.shape[1] == 3 * H * E
    else:
        assert a.shape[1] == H * E and b.shape[1] == 2 * H * E
    if index_sample is not None and index_sample.dim() == 3:
        assert idx_group > 0 and index_sample.shape[0] * idx_group == B, (index_sample.shape, idx_group, B)
        # tables may be a strided view over the host-drawn buffer (equal spacing between groups), rows packed
        assert index_sample.stride(2) == 1 and index_sample.stride(1) == index_sample.shape[2]
    elif index_sample is not None:
        assert index_sample.is_contiguous()
    drop = None
    if drop_p > 0.0:  # nn.Dropout on the (B,H,L_Q,L_K) probabilities of FullAttention (cross_modal_transformer.py:63)
        assert mode == 0 or (mode == 2 and forced_top is not None and n_top == LQ), "ProbAttention applies no dropout"
        shape = (B, H, LQ, LK)
        mask = RNG.take_forced(shape, a.device)
        site = RNG.next_site()
        if RNG.record is not None:
            RNG.record.append(RNG.materialise(site, shape, drop_p, a.device, mask))
        drop = (float(drop_p), site, mask)
    return _Attention.apply(a, b, offs, index_sample, dims, mode, n_top, out_layout, scale, forced_top, idx_group, drop)


def attention_map(a, b, offs, dims, mode: int, tops, scale: Optional[float] = None):
    """The dense (B, H, L_Q, L_K) attention map the reference returns with ``output_attention=True`` -- a debugging output,
    built with torch ops OUTSIDE the attention kernel from the projections it consumed and the rows it selected:
    FullAttention (mode 0): softmax(scale Q K^T) (cross_modal_transformer.py:60-66, SelfAttentionFamily.py:59-66);
    ProbAttention (mode 1 / 2): every row 1 / L_V except the selected queries, whose rows are their (mode 2: causally masked,
    ProbMask) softmax rows (cross_modal_transformer.py:134-138, SelfAttentionFamily.py:133-137)."""
    B, H, LQ, LK, E = dims
    scale = scale or 1.0 / math.sqrt(E)
    HE = H * E
    with torch.no_grad():
        q = a[:, offs[0]:offs[0] + HE].reshape(B, LQ, H, E).permute(0, 2, 1, 3).float()
        k = b[:, offs[1]:offs[1] + HE].reshape(B, LK, H, E).permute(0, 2, 1, 3).float()
        if mode == 0:
            return torch.softmax(scale * (q @ k.transpose(-1, -2)), dim=-1)
        idx = tops.long()  # (B, H, u)
        qs = torch.gather(q, 2, idx.unsqueeze(-1).expand(-1, -1, -1, E))
        scores = scale * (qs @ k.transpose(-1, -2))  # (B, H, u, L_K)
        if mode == 2:  # ProbMask: key j > query position is masked
            scores = scores.masked_fill(torch.arange(LK, device=a.device).view(1, 1, 1, LK) > idx.unsqueeze(-1), float("-inf"))
        rows = torch.softmax(scores, dim=-1)
        full = torch.full((B, H, LK, LK), 1.0 / LK, device=a.device, dtype=rows.dtype)
        full.scatter_(2, idx.unsqueeze(-1).expand(-1, -1, -1, LK), rows)
        return full


ROWCHAIN = os.environ.get("RF_ROWCHAIN", "1") != "0"  # row-local chains of the d_model = 64 decoder as single launches


def rowchain_supported(d_model: int, d_ff: int, n_proj: int) -> bool:
    return ROWCHAIN and _PRECISION == 1 and bool(_hip.lib().rf_rowchain_supported(d_model, d_ff, n_proj))


class _RowChain(torch.autograd.Function):
    """csrc/rowchain.hip: a (attention output) -> out-projection + residual x -> LayerNorm [-> conv FFN + residual ->
    LayerNorm] [-> projection for the next attention launch], one launch; backward one launch + the weight-gradient GEMMs
    (queued).  Only ``a`` and ``x`` are differentiable INPUTS: the parameters are plain arguments whose gradients go to
    the engine's sinks (the caller checks that every one of them has a slot)."""

    @staticmethod
    def forward(ctx, a, x, lin, norm1, ffn, proj, act, eps, save, drop=None):
        """lin = (Wo, bo); norm1 = (gamma, beta); ffn = None | (W1 (F, D[, 1]), b1, W2 (D, F[, 1]), b2, gamma, beta);
        proj = None | (Wp (NP, D), bp, grad slot of Wp, grad slot of bp); drop = None | (p, first dropout site: out-projection
        output, then hidden activation and conv2 output).  -> (x1 or y, proj or None)."""
        D = a.shape[-1]
        M = a.numel() // D
        a2, x2 = a.reshape(M, D).contiguous(), x.reshape(M, D).contiguous()
        dev = a.device
        f32 = dict(device=dev, dtype=torch.float32)
        F_ = ffn[0].shape[0] if ffn is not None else 0
        NP = proj[0].shape[0] if proj is not None else 0
        x1 = torch.empty(M, D, **f32)
        y = torch.empty(M, D, **f32) if ffn is not None else None
        pr = torch.empty(M, NP, **f32) if proj is not None else None
        sv = {}
        if save:
            sv["xhat1"], sv["rstd1"] = torch.empty(M, D, **f32), torch.empty(M, **f32)
            if ffn is not None:
                sv["xhat2"], sv["rstd2"], sv["h"] = torch.empty(M, D, **f32), torch.empty(M, **f32), torch.empty(M, F_, **f32)
                if act == "gelu":
                    sv["z"] = torch.empty(M, F_, **f32)
        c = _hip.RowChain()
        c.a, c.x, c.wo, c.bo, c.g1, c.be1 = ptr(a2), ptr(x2), ptr(lin[0]), ptr(lin[1]), ptr(norm1[0]), ptr(norm1[1])
        if ffn is not None:
            c.w1, c.b1, c.w2, c.b2, c.g2, c.be2 = (ptr(t) for t in ffn)
        if proj is not None:
            c.wp, c.bp = ptr(proj[0]), ptr(proj[1])
        c.x1, c.y, c.proj = ptr(x1), ptr(y), ptr(pr)
        for name in ("xhat1", "rstd1", "z", "h", "xhat2", "rstd2"):
            setattr(c, name, ptr(sv.get(name)))
        c.d_model, c.d_ff, c.n_proj, c.act, c.eps = D, F_, NP, ACT[act], eps
        drop_p, c.drop_site = (float(drop[0]), int(drop[1])) if drop is not None else (0.0, 0)
        rng = ptr(RNG.state(dev)) if drop_p > 0.0 else None
        import ctypes
        ev = PROFILE.begin() if PROFILE.on else None
        check(_hip.lib().rf_rowchain_fwd(ctypes.byref(c), M, drop_p, rng, _stream()), "rf_rowchain_fwd")
        if ev is not None:
            keep = (a2, x2, lin, norm1, ffn, proj, x1, y, pr, sv, c)
            PROFILE.end(f"rowchain_fwd_kernel<{1 if M <= 1024 else 2}>", ev, 2.0 * M * D * (D + 2 * F_ + NP),
                        4.0 * M * (D * (3 + (ffn is not None)) + NP + (len(sv) and (2 * D + 2 * F_))),
                        replay=lambda cc=c, k=keep: _hip.lib().rf_rowchain_fwd(ctypes.byref(cc), M, drop_p, rng, _stream()))
        out = y if ffn is not None else x1
        if save:
            ctx.sv, ctx.a2, ctx.x1, ctx.out = sv, a2, x1, out
            ctx.params = (lin, norm1, ffn, proj)
            ctx.cfg = (M, D, F_, NP, act, a.shape, x.shape, drop_p, c.drop_site)
        return out.view(x.shape), (pr.view(*x.shape[:-1], NP) if pr is not None else None)

    @staticmethod
    def backward(ctx, dout, dproj):
        import ctypes
        sv, a2, x1, out = ctx.sv, ctx.a2, ctx.x1, ctx.out
        lin, norm1, ffn, proj = ctx.params
        M, D, F_, NP, act, ashape, xshape, drop_p, drop_site = ctx.cfg
        ctx.sv = None
        dev = a2.device
        f32 = dict(device=dev, dtype=torch.float32)
        dpre1, da = torch.empty(M, D, **f32), torch.empty(M, D, **f32)
        c = _hip.RowChainBwd()
        dproj2 = dproj.reshape(M, NP).contiguous() if (dproj is not None and proj is not None) else None
        dout2 = dout.reshape(M, D).contiguous() if dout is not None else None
        c.dproj, c.dyin, c.wp = ptr(dproj2), ptr(dout2), (ptr(proj[0]) if dproj2 is not None else None)
        dpre2 = dz = None
        if ffn is not None:
            dpre2, dz = torch.empty(M, D, **f32), torch.empty(M, F_, **f32)
            c.w1, c.w2, c.g2 = ptr(ffn[0]), ptr(ffn[2]), ptr(ffn[4])
            c.xhat2, c.rstd2, c.zsrc = ptr(sv["xhat2"]), ptr(sv["rstd2"]), ptr(sv["z"] if "z" in sv else sv["h"])
            c.dpre2, c.dz, c.dg2, c.db2 = ptr(dpre2), ptr(dz), ptr(_slot(ffn[4])), ptr(_slot(ffn[5]))
        c.wo, c.g1, c.xhat1, c.rstd1 = ptr(lin[0]), ptr(norm1[0]), ptr(sv["xhat1"]), ptr(sv["rstd1"])
        c.dpre1, c.da, c.dg1, c.db1 = ptr(dpre1), ptr(da), ptr(_slot(norm1[0])), ptr(_slot(norm1[1]))
        c.d_model, c.d_ff, c.n_proj, c.act = D, F_, (NP if dproj2 is not None else 0), ACT[act]
        # with dropout the residual input's gradient (unmasked) and the out-projection's weight-gradient operand (masked) differ
        dx = torch.empty(M, D, **f32) if drop_p > 0.0 else dpre1
        c.dx, c.drop_site = (ptr(dx) if drop_p > 0.0 else None), drop_site
        ev = PROFILE.begin() if PROFILE.on else None
        check(_hip.lib().rf_rowchain_bwd(ctypes.byref(c), M, drop_p, ptr(RNG.state(dev)) if drop_p > 0.0 else None, _stream()),
              "rf_rowchain_bwd")
        if ev is not None:
            keep = (dproj2, dout2, sv, dpre1, da, dpre2, dz, c)
            PROFILE.end(f"rowchain_bwd_kernel<{1 if M <= 1024 else 2}>", ev, 2.0 * M * D * (D + 2 * F_ + NP),
                        4.0 * M * (D * 6 + 2 * F_ + NP))
        _wrote(_slot(norm1[0]), _slot(norm1[1]))
        pairs = []
        if dproj2 is not None:
            pairs.append((dproj2, out, proj[2], proj[3]))
        if ffn is not None:
            _wrote(_slot(ffn[4]), _slot(ffn[5]))
            pairs.append((dpre2, sv["h"], _slot(ffn[2]).view(D, F_), _slot(ffn[3])))
            pairs.append((dz, x1, _slot(ffn[0]).view(F_, D), _slot(ffn[1])))
        pairs.append((dpre1, a2, _slot(lin[0]), _slot(lin[1])))
        for gy, xin, w_into, b_into in pairs:
            if _weight_grad(gy, xin, into=w_into, bias_into=b_into) is not True:
                colsum(gy, into=b_into)
            _wrote(w_into, b_into)
        return da.view(ashape), dx.view(xshape), None, None, None, None, None, None, None, None


def rowchain(a, x, lin, norm1, ffn, proj, act: str, eps: float, drop_p: float = 0.0):
    """See ``_RowChain``.  The caller has checked ``rowchain_supported`` and, when gradients are needed, that the sinks are
    active and every parameter has a slot.  ``drop_p`` > 0: nn.Dropout of the layer (train mode) -- one site for the
    out-projection output, two more (hidden activation, conv2 output) with an FFN block, numbered in call order."""
    need_grad = torch.is_grad_enabled() and (a.requires_grad or x.requires_grad)
    drop = None
    if drop_p > 0.0:
        n_sites = 3 if ffn is not None else 1
        site0 = RNG.site
        RNG.site += n_sites
        if RNG.record is not None:
            D = a.shape[-1]
            widths = (D, ffn[0].shape[0], D) if ffn is not None else (D,)
            for k, cols in enumerate(widths):
                RNG.record.append(RNG.materialise(site0 + k, tuple(a.shape[:-1]) + (cols,), drop_p, a.device))
        drop = (drop_p, site0)
    return _RowChain.apply(a, x, lin, norm1, ffn, proj, act, eps, need_grad, drop)


class _TrajHead(torch.autograd.Function):
    """postprocess_batch + discounted SmoothL1 losses + ADE/FDE in one launch (and one for backward)."""

    @staticmethod
    def forward(ctx, out, last_gps, target_gps, target_vis, gamma, ratio, dense_on, mstd, mmean):
        _req(out, "traj_head.out")
        B, P, C = out.shape
        E = target_vis.shape[-1] if target_vis is not None else 0

        def rows_in_place(t, width):  # (B, rows, width) with packed rows: only the batch stride is free (slices along time)
            return t.dtype == torch.float32 and t.stride(2) == 1 and t.stride(1) == width
        # the last P rows of the decoder output, the last input position and the first P rows of the target features are read
        # where they lie (batch strides): three strided-copy launches less in front of the one launch that cannot overlap anything
        if not rows_in_place(out, C):
            out = out.contiguous()
        last = last_gps.reshape(B, 1, 2) if last_gps.dim() != 3 else last_gps
        if not rows_in_place(last, 2):
            last = last.to(torch.float32).contiguous()
        tgt = target_gps.to(torch.float32).contiguous()
        tv = target_vis
        if tv is not None and not rows_in_place(tv, E):
            tv = tv.contiguous()
        pos = torch.empty(B, P, 2, device=out.device, dtype=torch.float32)
        gpos = torch.empty(B, P, 2, device=out.device, dtype=torch.float32)
        scal = torch.empty(8, device=out.device, dtype=torch.float32)
        obs, lbs, vbs = out.stride(0), last.stride(0), (tv.stride(0) if tv is not None else 0)
        check(_hip.lib().rf_traj_head_fwd(ptr(out), ptr(last), ptr(tgt), ptr(tv), ptr(pos), ptr(gpos), ptr(scal), B, P, C,
                                          E, gamma, ratio, 1 if dense_on else 0, mstd, mmean, obs, lbs, vbs, _stream()),
              "rf_traj_head_fwd")
        ctx.save_for_backward(out, tv if tv is not None else out, gpos, scal)
        ctx.cfg = (B, P, C, E, gamma, mstd, tv is not None)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(pos)
        return scal[4], scal[0], scal[1], scal[2], scal[3], pos

    @staticmethod
    def backward(ctx, g_loss, g_traj, g_dense, g_ade, g_fde, g_pos):
        out, tv, gpos, scal = ctx.saved_tensors
        B, P, C, E, gamma, mstd, has_vis = ctx.cfg
        if g_traj is not None or g_dense is not None or g_ade is not None or g_fde is not None:
            raise NotImplementedError("traj_head: only the combined loss is differentiable")
        alloc = torch.zeros if C > 2 + E else torch.empty
        dout = alloc(B, P, C, device=out.device, dtype=torch.float32)
        gl = g_loss.reshape(1).contiguous() if g_loss is not None else None
        check(_hip.lib().rf_traj_head_bwd(ptr(out), ptr(tv) if has_vis else None, ptr(gpos), ptr(scal), ptr(gl), ptr(dout),
                                          B, P, C, E, gamma, mstd, out.stride(0), tv.stride(0) if has_vis else 0, _stream()),
              "rf_traj_head_bwd")
        return dout, None, None, None, None, None, None, None, None


def traj_head(out, last_gps, target_gps, target_vis, gamma: float, ratio: float, dense_on: bool,
              motion_std: float = 1.0, motion_mean: float = 0.0):
    """-> (loss, traj_loss, dense_loss, ade, fde, positions); only ``loss`` carries gradient (to ``out``)."""
    return _TrajHead.apply(out, last_gps, target_gps, target_vis, float(gamma), float(ratio), bool(dense_on),
                           float(motion_std), float(motion_mean))


class _MotionInput(torch.autograd.Function):
    """x = [R(-origin) motion | (angle - origin)/pi | |motion| | d|motion| | visual] in one launch (rf_motion_input);
    the motion inputs carry no gradient, d visual is the matching slice of dX."""

    @staticmethod
    def forward(ctx, motion, visual, rotate_motion: bool, zero_visual: bool):
        B, T, _ = motion.shape
        E = 0 if visual is None else visual.shape[-1]
        motion = motion.contiguous().float()
        vis = None if visual is None else visual.contiguous().float()
        x = torch.empty(B, T, 5 + E, device=motion.device, dtype=torch.float32)
        origin = torch.empty(B, device=motion.device, dtype=torch.float32)
        check(_hip.lib().rf_motion_input(ptr(motion), ptr(vis), ptr(x), ptr(origin), B, T, E, 1 if rotate_motion else 0,
                                         1 if zero_visual else 0, _stream()), "rf_motion_input")
        ctx.E, ctx.zero_visual = E, zero_visual
        ctx.mark_non_differentiable(origin)
        return x, origin

    @staticmethod
    def backward(ctx, dx, _dorigin):
        dvis = None
        if ctx.E and not ctx.zero_visual and ctx.needs_input_grad[1]:
            dvis = dx[..., 5:]
        return None, dvis, None, None


class _RotateHead(torch.autograd.Function):
    """Channels 0,1 of the backbone output rotated back by +origin (rf_rotate_head); backward rotates by -origin."""

    @staticmethod
    def forward(ctx, out, origin):
        out = out.contiguous().float()
        B, P, C = out.shape
        y = torch.empty_like(out)
        check(_hip.lib().rf_rotate_head(ptr(out), ptr(origin), ptr(y), B, P, C, 1.0, _stream()), "rf_rotate_head")
        ctx.save_for_backward(origin)
        return y

    @staticmethod
    def backward(ctx, dy):
        (origin,) = ctx.saved_tensors
        dy = dy.contiguous()
        B, P, C = dy.shape
        dout = torch.empty_like(dy)
        check(_hip.lib().rf_rotate_head(ptr(dy), ptr(origin), ptr(dout), B, P, C, -1.0, _stream()), "rf_rotate_head")
        return dout, None


# ---------------------------------------------------------------------------------------------------
# small tensor plumbing as single launches (csrc/smallops.hip)
# ---------------------------------------------------------------------------------------------------
def median_windows(x: torch.Tensor, target: int) -> torch.Tensor:
    """``median_downsampler`` (utils/filter.py:5-43) of a (B,T,C) fp32 device tensor in one launch."""
    _req(x, "median_windows.x")
    x = x.contiguous()
    B, T, C = x.shape
    y = torch.empty(B, target, C, device=x.device, dtype=torch.float32)
    check(_hip.lib().rf_median_windows(ptr(x), ptr(y), B, T, C, target, _stream()), "rf_median_windows")
    return y


def motion_diff(gps: torch.Tensor, normalize: bool, mean: float, std: float) -> torch.Tensor:
    """(B,T,2) positions -> (B,T,2) motion with the zero row in front (routeformer.py:284-292); no gradient."""
    _req(gps, "motion_diff.gps")
    gps = gps.contiguous()
    B, T, C = gps.shape
    assert C == 2
    out = torch.empty_like(gps)
    check(_hip.lib().rf_motion_diff(ptr(gps), ptr(out), B, T, 1 if normalize else 0, float(mean), float(std), _stream()),
          "rf_motion_diff")
    return out


class _TimeTable(torch.autograd.Function):
    """(L, d) table  l * w + pe[l]  of DataEmbedding (time feature of the position index + positional embedding)."""

    @staticmethod
    def forward(ctx, w, pe, L, gw):
        d = w.numel()
        wv = w.reshape(d).contiguous()
        pe2 = pe.reshape(-1, d)[:L].contiguous()
        out = torch.empty(L, d, device=w.device, dtype=torch.float32)
        check(_hip.lib().rf_time_table(ptr(wv), ptr(pe2), ptr(out), L, d, _stream()), "rf_time_table")
        ctx.cfg = (L, d, w.shape, gw)
        return out

    @staticmethod
    def backward(ctx, dout):
        L, d, wshape, gw = ctx.cfg
        dout = dout.contiguous()
        if gw is not None:
            check(_hip.lib().rf_time_table_bwd(ptr(dout), ptr(gw), L, d, 1, _stream()), "rf_time_table_bwd")
            _wrote(gw)
            return None, None, None, None
        dw = torch.empty(d, device=dout.device, dtype=torch.float32)
        check(_hip.lib().rf_time_table_bwd(ptr(dout), ptr(dw), L, d, 0, _stream()), "rf_time_table_bwd")
        return dw.view(wshape), None, None, None


def time_table(w, pe, L: int):
    _req(w, "time_table.w")
    return _TimeTable.apply(w, pe, L, _slot(w))


class _Timeline(torch.autograd.Function):
    """(N,F,E) features of the sub-sampled frames -> (N,T,E) zero timeline with row idx[f] = feature f, one launch."""

    @staticmethod
    def forward(ctx, feats, idx, T):
        feats = feats.contiguous()
        N, F_, E = feats.shape
        out = torch.empty(N, T, E, device=feats.device, dtype=torch.float32)
        check(_hip.lib().rf_timeline_scatter(ptr(feats), ptr(idx), ptr(out), N, T, F_, E, _stream()), "rf_timeline_scatter")
        ctx.save_for_backward(idx)
        ctx.dims = (N, T, F_, E)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        N, T, F_, E = ctx.dims
        dout = dout.contiguous()
        dfeats = torch.empty(N, F_, E, device=dout.device, dtype=torch.float32)
        check(_hip.lib().rf_timeline_gather(ptr(dout), ptr(idx), ptr(dfeats), N, T, F_, E, _stream()), "rf_timeline_gather")
        return dfeats, None, None


def timeline(feats, idx, T: int):
    """``idx``: int64 (F) device tensor of distinct time steps."""
    _req(feats, "timeline.feats")
    assert idx.dtype == torch.int64 and idx.is_cuda and idx.is_contiguous()
    return _Timeline.apply(feats, idx, T)


class _SmartTail(torch.autograd.Function):
    """x (B,L,C) -> (decoder input (B,L+P,C) = cat(x, last row repeated | zeros), alias of x for the encoder): both
    gradients of x meet in ONE backward launch instead of slice + sum + add + autograd's add."""

    @staticmethod
    def forward(ctx, x, P, smart):
        x = x.contiguous()
        B, L, C = x.shape
        y = torch.empty(B, L + P, C, device=x.device, dtype=torch.float32)
        check(_hip.lib().rf_smart_tail_fwd(ptr(x), ptr(y), B, L, P, C, 1 if smart else 0, _stream()), "rf_smart_tail_fwd")
        ctx.dims = (B, L, P, C, smart)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dalias):
        B, L, P, C, smart = ctx.dims
        if dy is None:
            return dalias, None, None
        dy = dy.contiguous()
        extra = dalias.contiguous() if dalias is not None else None
        dx = torch.empty(B, L, C, device=dy.device, dtype=torch.float32)
        check(_hip.lib().rf_smart_tail_bwd(ptr(dy), ptr(extra), ptr(dx), B, L, P, C, 1 if smart else 0, _stream()),
              "rf_smart_tail_bwd")
        return dx, None, None


def smart_tail(x, P: int, smart: bool):
    _req(x, "smart_tail.x")
    return _SmartTail.apply(x, P, smart)


def motion_input(motion, visual, rotate_motion: bool, zero_visual: bool = False):
    return _MotionInput.apply(motion, visual, rotate_motion, zero_visual)


def rotate_head(out, origin):
    return _RotateHead.apply(out, origin)


class _AssembleStreams(torch.autograd.Function):
    """cat([stream_s + emb_s for s], dim=1) with ``None`` streams standing for zeros -- one launch each way."""

    @staticmethod
    def forward(ctx, n, *args):
        streams, embs, sinks = args[:n], args[n:2 * n], args[2 * n:3 * n]
        first = next(t for t in streams if t is not None)
        B, T, E = first.shape
        xs = [None if t is None else t.contiguous() for t in streams]
        es = [e.reshape(-1).contiguous() for e in embs]
        out = torch.empty(B, n * T, E, device=first.device, dtype=torch.float32)
        import ctypes
        sp = (ctypes.c_void_p * n)(*[ptr(t) for t in xs])
        ep = (ctypes.c_void_p * n)(*[ptr(t) for t in es])
        check(_hip.lib().rf_assemble_streams_fwd(sp, ep, ptr(out), B, T, E, n, _stream()), "rf_assemble_streams_fwd")
        ctx.dims = (n, B, T, E)
        ctx.sinks = sinks
        ctx.present = [t is not None for t in streams]
        ctx.emb_shapes = [e.shape for e in embs]
        return out

    @staticmethod
    def backward(ctx, dout):
        import ctypes
        n, B, T, E = ctx.dims
        dout = dout.contiguous()
        sink = all(s is not None for s in ctx.sinks)
        dembs = list(ctx.sinks) if sink else [torch.zeros(E, device=dout.device, dtype=torch.float32) for _ in range(n)]
        dp = (ctypes.c_void_p * n)(*[ptr(t) for t in dembs])
        check(_hip.lib().rf_assemble_streams_bwd(ptr(dout), dp, B, T, E, n, _stream()), "rf_assemble_streams_bwd")
        if sink:
            _wrote(*dembs)
        d4 = dout.view(B, n, T, E)
        dstreams = [d4[:, s] if ctx.present[s] else None for s in range(n)]
        dembs_out = [None] * n if sink else [d.view(sh) for d, sh in zip(dembs, ctx.emb_shapes)]
        return (None, *dstreams, *dembs_out, *([None] * n))


def assemble_streams(streams, embeddings):
    """Fusion-encoder input (routeformer.py:331-345): streams (B,T,E) or None (= zeros) + their learned embeddings,
    concatenated along time."""
    n = len(streams)
    assert n == len(embeddings) and 1 <= n <= 4
    first = next(t for t in streams if t is not None)
    _req(first, "assemble_streams")
    E = first.shape[-1]
    if E % 4 or E > 64 or any(t is not None and t.dtype != torch.float32 for t in streams):
        # sizes outside the kernel's range: same arithmetic from device-side torch ops
        return torch.cat([(torch.zeros_like(first) if t is None else t) + e for t, e in zip(streams, embeddings)], dim=1)
    return _AssembleStreams.apply(n, *streams, *embeddings, *[_slot(e) for e in embeddings])


# ---------------------------------------------------------------------------------------------------
# Fused per-sequence encoder stack (csrc/seqlayer.hip): every EncoderLayer of a PerceiveEncoder in ONE launch,
# one workgroup per sequence.  bf16 matrix-core mode, L <= 80, d_model 128, 8 heads; the backward pass runs the
# layer-by-layer kernels on the tensors the fused forward saved.
# ---------------------------------------------------------------------------------------------------
SEQSTACK = os.environ.get("RF_SEQSTACK", "1") != "0"
WEIGHTS_EPOCH = 0  # bumped by the training engine after every optimizer update (kernels write parameters in place)


def seqstack_supported(L: int, d_model: int, n_heads: int, d_ff: int, sample_k: int, n_top: int) -> bool:
    return bool(SEQSTACK and _PRECISION == 1 and _hip.lib().rf_seqlayer_supported(L, d_model, n_heads, d_ff, sample_k, n_top))


class PackPlan:
    """Every pack entry of every fused stack of a model as ONE table in device memory -> one rf_seqlayer_pack_table launch
    per step (the entries name static addresses: parameters inside the engine's flat buffer, the stacks' blobs)."""

    def __init__(self, ents, device):
        import ctypes
        import numpy as np
        assert ents
        arr = (_hip.SeqPackEntry * len(ents))()
        first, blocks = [], 0
        for e, (w, off, ldw, N, K_, tr) in zip(arr, ents):
            assert K_ == 0 or (N % 16 == 0 and K_ % 32 == 0 and off % 16 == 0)
            e.w, e.out, e.ldw, e.N, e.K, e.transpose, e.residual = w, off, ldw, N, K_, tr & 1, tr >> 1
            first.append(blocks)
            blocks += int(_hip.lib().rf_seqlayer_pack_blocks(N, K_))
        raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
        self.table = torch.from_numpy(raw).to(device)
        self.first = torch.tensor(first, dtype=torch.int32, device=device)
        self.count, self.blocks = len(ents), blocks

    def launch(self):
        check(_hip.lib().rf_seqlayer_pack_table(self.table.data_ptr(), self.first.data_ptr(), self.count, self.blocks,
                                                _stream()), "rf_seqlayer_pack_table")


def _pack_launch(ents, collect):
    """ents: (w_ptr, out_ptr, ldw, N, K, transpose).  collect: a list that takes them (PackPlan) instead of launching."""
    if collect is not None:
        collect.extend(ents)
        return
    for s0 in range(0, len(ents), _hip.SEQLAYER_MAX_PACK):
        chunk = ents[s0:s0 + _hip.SEQLAYER_MAX_PACK]
        arr = (_hip.SeqPackEntry * len(chunk))()
        for e, (w, off, ldw, N, K_, tr) in zip(arr, chunk):
            e.w, e.out, e.ldw, e.N, e.K, e.transpose, e.residual = w, off, ldw, N, K_, tr & 1, tr >> 1
        check(_hip.lib().rf_seqlayer_pack(arr, len(chunk), _stream()), "rf_seqlayer_pack")


def seqstack_pack(layers, out: torch.Tensor, stride: int, collect=None):
    """layers: per layer a dict of fp32 device tensors -- wq/wk/wv (128,128) or wqkv (384,128), wo (128,128), w1
    (F,128), w2 (128,F), bqkv (384) or bq/bk/bv, bo, b1, b2, g1, be1, g2, be2 -> ``out`` (uint8, n_layers * stride)."""
    import ctypes
    ents = []
    base = out.data_ptr()

    def mat(w, off, N, K, lo=0):
        assert w.dtype == torch.float32 and w.stride(-1) == 1
        ents.append((w, off, w.stride(0), N, K, 2 * lo))  # flags: bit 0 transpose, bit 1 low half of the split-bf16 weight

    def vecs(v, off, n):
        assert v.dtype == torch.float32 and v.is_contiguous() and v.numel() == n
        ents.append((v, off, 0, n, 0, 0))

    for li, d in enumerate(layers):
        F_ = d["w1"].shape[0]
        o = base + li * stride
        o_wo, o_w1 = 24 * 4096, 24 * 4096 + 8 * 4096
        o_w2 = o_w1 + (F_ // 16) * 4096
        o_vec = o_w2 + 8 * (F_ // 32) * 1024
        o_lo = (o_vec + (1152 + F_) * 4 + 255) & ~255  # low halves of Wq | Wk (split-bf16 q / k projection)
        if "wqkv" in d:
            mat(d["wqkv"], o, 384, 128)
            mat(d["wqkv"][:256], o + o_lo, 256, 128, lo=1)
            vecs(d["bqkv"], o + o_vec, 384)
        else:
            for i, n in enumerate(("q", "k", "v")):
                mat(d["w" + n], o + i * 32 * 1024, 128, 128)
                vecs(d["b" + n], o + o_vec + i * 512, 128)
                if i < 2:
                    mat(d["w" + n], o + o_lo + i * 32 * 1024, 128, 128, lo=1)
        mat(d["wo"], o + o_wo, 128, 128)
        mat(d["w1"], o + o_w1, F_, 128)
        mat(d["w2"], o + o_w2, 128, F_)
        for n, pos, cnt in (("bo", 384, 128), ("b1", 512, F_), ("b2", 512 + F_, 128), ("g1", 640 + F_, 128),
                            ("be1", 768 + F_, 128), ("g2", 896 + F_, 128), ("be2", 1024 + F_, 128)):
            vecs(d[n], o + o_vec + 4 * pos, cnt)
    _pack_launch([(w.data_ptr(), off, ldw, N, K_, fl) for (w, off, ldw, N, K_, fl) in ents], collect)


def seqstack_bwd_pack(layers, out: torch.Tensor, stride: int, collect=None):
    """Transposed fragment order of the weights for the fused backward (csrc/seqlayer_bwd.hip): per layer dict with wqkv
    (384,128), wo (128,128), w1 (F,128), w2 (128,F), g1, g2 -> ``out`` (uint8, n_layers * stride)."""
    ents, base = [], out.data_ptr()
    for li, d in enumerate(layers):
        F_ = d["w1"].shape[0]
        o = base + li * stride
        o_w1t = (F_ // 16) * 4096
        o_wot = o_w1t + 8 * (F_ // 32) * 1024
        o_wqkvt = o_wot + 32 * 1024
        o_vec = o_wqkvt + 96 * 1024
        for w, off, N, K_ in ((d["w2"], o, F_, 128), (d["w1"], o + o_w1t, 128, F_), (d["wo"], o + o_wot, 128, 128),
                              (d["wqkv"], o + o_wqkvt, 128, 384)):
            assert w.dtype == torch.float32 and w.stride(-1) == 1 and tuple(w.shape) == (K_, N)
            ents.append((w.data_ptr(), off, w.stride(0), N, K_, 1))
        for v, pos in ((d["g1"], 0), (d["g2"], 128)):
            assert v.dtype == torch.float32 and v.is_contiguous() and v.numel() == 128
            ents.append((v.data_ptr(), o + o_vec + 4 * pos, 0, 128, 0, 0))
    _pack_launch(ents, collect)


def seqstack_bwd_pack_bytes(d_ff: int) -> int:
    return int(_hip.lib().rf_seqlayer_bwd_pack_bytes(d_ff))


SEQSTACK_BWD = os.environ.get("RF_SEQSTACK_BWD", "1") != "0"  # fused backward of the encoder stacks (else per layer)


def _seqstack_bwd_launch(dy2, sv, wpack, stride, ln_slots, B, L, F_, act, n_top, drop_p: float = 0.0, drop_site0: int = 0):
    """Run the fused backward on the saves `sv` of `_seqstack_launch`.  ln_slots: per layer (dgamma1, dbeta1, dgamma2,
    dbeta2) fp32 accumulators.  -> (dx, {"dpre2", "dz", "dpre1", "dqkv"} slabs [layers, B*L, width])."""
    import ctypes
    n, M, dev = len(ln_slots), B * L, dy2.device
    f32 = dict(device=dev, dtype=torch.float32)
    # the `dy` operands of the weight-gradient GEMMs: bf16-rounded values either way (they are LDS images of the kernel);
    # kept as bf16 they cost half the bytes to write here and to read in the grouped weight-gradient launch
    gdt = dict(device=dev, dtype=torch.bfloat16 if BF16_SAVES else torch.float32)
    out = {"dpre2": torch.empty(n, M, 128, **gdt), "dz": torch.empty(n, M, F_, **gdt),
           "dpre1": torch.empty(n, M, 128, **gdt), "dqkv": torch.empty(n, M, 384, **gdt)}
    dx = torch.empty(M, 128, **f32)
    st = _hip.SeqStackBwd()
    st.wpack, st.wpack_stride, st.n_layers = wpack.data_ptr(), stride, n
    zsrc = sv["z"] if "z" in sv else sv["h"]
    assert sv["xhat1"].dtype == sv["xhat2"].dtype
    st.flags = ((1 if BF16_SAVES else 0) | (2 if sv["qkv"].dtype == torch.bfloat16 else 0)
                | (4 if sv["xhat1"].dtype == torch.bfloat16 else 0) | (8 if zsrc.dtype == torch.bfloat16 else 0))
    for name in ("qkv", "xhat1", "rstd1", "xhat2", "rstd2", "top"):
        setattr(st, name, ptr(sv[name]))
    st.zsrc = ptr(sv["z"] if "z" in sv else sv["h"])
    for name, t in out.items():
        setattr(st, name, ptr(t))
    for i, (a, b_, c, d) in enumerate(ln_slots):
        for t in (a, b_, c, d):
            assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == 128
        st.dgamma1[i], st.dbeta1[i], st.dgamma2[i], st.dbeta2[i] = a.data_ptr(), b_.data_ptr(), c.data_ptr(), d.data_ptr()
    ev = PROFILE.begin() if PROFILE.on else None
    args = (ctypes.byref(st), ptr(dy2), ptr(dx), B, L, 128, 8, F_, ACT[act], n_top, 1.0 / math.sqrt(16.0), float(drop_p),
            ptr(RNG.state(dev)) if drop_p > 0 else None, drop_site0)
    check(_hip.lib().rf_seqlayer_bwd(*args, _stream()), "rf_seqlayer_bwd")
    if ev is not None:
        flops = n * B * (2.0 * L * 128 * (384 + 128 + 2 * F_) + 8 * 10.0 * n_top * L * 16)
        gb = 2.0 if BF16_SAVES else 4.0  # bytes per element of the four gradient slabs (384 + 128 + 128 + F_ wide)
        nbytes = (4.0 * M * 128 * 2 + n * (2.0 * (4 * 128 * 128 + 2 * 128 * F_) + 4.0 * M * (384 + 128 * 2 + F_ + 2)
                                           + gb * M * (384 + 128 * 2 + F_)))
        keep = (dy2, dx, sv, out, wpack, st, ln_slots)
        PROFILE.end(f"seq_stack_bwd_kernel<{3 if L <= 48 else 5}, {'true' if drop_p > 0 else 'false'}>", ev, flops, nbytes,
                    replay=lambda a=args, k=keep: _hip.lib().rf_seqlayer_bwd(*a, _stream()))
    return dx, out


def seqstack_pack_bytes(d_ff: int) -> int:
    return int(_hip.lib().rf_seqlayer_pack_bytes(d_ff))


# bf16 storage for what only the weight-gradient GEMMs read again (fused per-sequence stacks, bf16 mode): ctx / x1 / h of
# the forward and the four `dy` slabs of the backward.  Lossless with respect to the arithmetic (they are bf16 MFMA operands
# in those GEMMs, rounded the same way) and half the bytes (PMC): 645 -> 569 MB moved by the camera-token stack's forward,
# 732 -> 553 MB by its backward, 355 -> 281 MB per weight-gradient launch.
BF16_SAVES = os.environ.get("RF_BF16_SAVES", "1") != "0"
BF16_QKV = os.environ.get("RF_BF16_QKV", "1") != "0"  # ... and the saved q | k | v (measurement switch)
# x-hat of both norms and the pre-activation z as bf16 as well: NOT lossless (the backward's fp32 element-wise math reads them:
# 2^-9 relative on x-hat / z), halves the rest of what the stacks save.  RF_BF16_NORM_SAVES=1 turns it on.
BF16_NORM_SAVES = os.environ.get("RF_BF16_NORM_SAVES", "0") == "1"


def _seqstack_launch(x2, wpack, stride, idx_list, idx_group, B, L, F_, act, sample_k, n_top, save, forced_tops, eps,
                     drop_p: float = 0.0, drop_site0: int = 0):
    """Run the fused forward.  -> dict of output / saved tensors ([layers, B*L, width] slabs)."""
    import ctypes
    n = len(idx_list)
    dev, M = x2.device, B * L
    f32 = dict(device=dev, dtype=torch.float32)
    half = save and BF16_SAVES
    # every layer's output only when a backward reads them as the next layer's input: with bf16 saves that is the `xin` image
    sv = {"y": torch.empty(n if (save and not half) else 1, M, 128, **f32)}
    force = forced_tops is not None
    if save or force:
        sv["top"] = (torch.stack([t.to(device=dev, dtype=torch.int32) for t in forced_tops]).contiguous() if force
                     else torch.empty(n, B, 8, n_top, device=dev, dtype=torch.int32))
    if save:
        # ctx, x1 and (GELU: z is there for the activation gradient) h are read again by the weight-gradient GEMMs only,
        # as bf16 MFMA operands -- BF16_SAVES keeps them as the bf16 images the kernel holds anyway
        bf = dict(device=dev, dtype=torch.bfloat16)
        # q | k | v: the backward's matrix cores consume them rounded to bf16 whatever the slab holds (lossless as well)
        for name, width in (("qkv", 384), ("ctx", 128), ("xhat1", 128), ("x1", 128), ("xhat2", 128), ("h", F_)):
            as_bf = half and (name in ("ctx", "x1") or (name == "qkv" and BF16_QKV) or (name == "h" and act == "gelu")
                              or (name in ("xhat1", "xhat2") and BF16_NORM_SAVES))
            sv[name] = torch.empty(n, M, width, **(bf if as_bf else f32))
        if act == "gelu":
            sv["z"] = torch.empty(n, M, F_, **(bf if (half and BF16_NORM_SAVES) else f32))
        if half:  # every layer's input as the projection consumed it: the x operand of its weight gradient
            sv["xin"] = torch.empty(n, M, 128, **bf)
        sv["rstd1"] = torch.empty(n, M, **f32)
        sv["rstd2"] = torch.empty(n, M, **f32)
    st = _hip.SeqStack()
    st.wpack, st.wpack_stride, st.n_layers = wpack.data_ptr(), stride, n
    # bit 0: ctx / x1 / h, bit 1: q | k | v, bit 2: xhat1 / xhat2 / z are bf16 slabs
    st.flags = (1 | (2 if BF16_QKV else 0) | (4 if BF16_NORM_SAVES else 0)) if half else 0
    idx_stride = 0
    for i, t in enumerate(idx_list):
        assert t.dim() == 3 and t.dtype == torch.int32 and t.stride(2) == 1 and t.stride(1) == t.shape[2], "key-sample table"
        st.idx[i] = t.data_ptr()
        s_ = t.stride(0) if t.shape[0] > 1 else L * sample_k
        assert i == 0 or s_ == idx_stride, "the layers' key-sample tables must share one group stride"
        idx_stride = s_
    st.idx_stride = idx_stride
    for name in ("top", "y", "qkv", "ctx", "xhat1", "rstd1", "x1", "z", "h", "xhat2", "rstd2", "xin"):
        setattr(st, name, ptr(sv.get(name)))
    ev = PROFILE.begin() if PROFILE.on else None
    args = (ctypes.byref(st), ptr(x2), B, L, 128, 8, F_, ACT[act], sample_k, n_top, idx_group, 1 if force else 0,
            1 if save else 0, 1.0 / math.sqrt(16.0), eps, float(drop_p), ptr(RNG.state(dev)) if drop_p > 0 else None,
            drop_site0)
    check(_hip.lib().rf_seqlayer_fwd(*args, _stream()), "rf_seqlayer_fwd")
    if ev is not None:
        flops = n * B * (2.0 * L * 128 * (384 + 128 + 2 * F_) + 8 * (2.0 * L * L * 16 + 4.0 * n_top * L * 16))
        nbytes = 4.0 * M * 128 * 2 + n * (2.0 * (4 * 128 * 128 + 2 * 128 * F_)
                                          + (4.0 * M * (384 + 128 * 5 + 2 * F_ + 2) if save else 0.0))
        keep = (x2, wpack, idx_list, sv, st)
        PROFILE.end(f"seq_stack_fwd_kernel<{3 if L <= 48 else 5}, {'true' if save else 'false'}, {'true' if drop_p > 0 else 'false'}>",
                    ev, flops, nbytes,
                    replay=lambda a=args, k=keep: _hip.lib().rf_seqlayer_fwd(*a, _stream()))
    return sv


class _SeqStack(torch.autograd.Function):
    """y = EncoderLayer_n(... EncoderLayer_1(x)) for a stack of ProbSparse encoder layers: ONE fused forward launch
    (dropout inside, Philox masks).  Backward: ONE fused launch for the data path of every layer
    (csrc/seqlayer_bwd.hip, masks regenerated) followed by the grouped weight-gradient GEMMs; with
    ``RF_SEQSTACK_BWD=0`` or in deterministic mode per layer the row-block / attention backward kernels on the same
    saved tensors.  Parameter gradients go through the engine's sinks (the fused path is only taken with sinks
    active or without grad)."""

    @staticmethod
    def forward(ctx, x, stack, idx_list, idx_group, save, drop_p=0.0):
        B, L, D = x.shape
        x2 = x.reshape(B * L, D).contiguous()
        lay0 = stack.layers[0]
        F_ = lay0.conv1.weight.shape[0]
        sample_k, n_top = prob_sizes(L, L, lay0.attention.factor)
        forced = None
        if TOPS.forced is not None:
            forced = [TOPS.forced.pop(0) for _ in stack.layers]
        site0 = 0
        if drop_p > 0.0:  # three dropout sites per layer, numbered in the reference's call order
            assert RNG.forced is None
            site0 = RNG.site
            RNG.site += 3 * len(stack.layers)
            if RNG.record is not None:
                for li in range(len(stack.layers)):
                    for k, cols in enumerate((D, F_, D)):
                        RNG.record.append(RNG.materialise(site0 + 3 * li + k, (B, L, cols), drop_p, x.device))
        if forced is not None and TOPS.shadow is not None:  # free pass on the same input, selections only
            free = _seqstack_launch(x2, stack.wpack, stack.stride, idx_list, idx_group, B, L, F_, lay0.act, sample_k, n_top,
                                    True, None, lay0.norm1.eps, 0.0, 0)
            TOPS.shadow.extend(t for t in free["top"])
        sv = _seqstack_launch(x2, stack.wpack, stack.stride, idx_list, idx_group, B, L, F_, lay0.act, sample_k, n_top,
                              save, forced, lay0.norm1.eps, drop_p, site0)
        if TOPS.record is not None and "top" in sv:
            for t in sv["top"]:
                TOPS.record.append(t.clone())
        if save:
            ctx.sv, ctx.stack, ctx.x2, ctx.dims, ctx.drop = sv, stack, x2, (B, L, F_, n_top), (float(drop_p), site0)
        return sv["y"][-1].view(B, L, D)

    @staticmethod
    def backward(ctx, dy):
        sv, stack, x2, (B, L, F_, n_top) = ctx.sv, ctx.stack, ctx.x2, ctx.dims
        drop_p, site0 = ctx.drop
        M, D, H, E = B * L, 128, 8, 16
        dy2 = dy.reshape(M, D).contiguous()
        if SEQSTACK_BWD and not DETERMINISTIC and stack.wpack_bwd is not None:
            # ---- one launch for the data path of every layer; the weight gradients follow as grouped GEMMs ----
            slots = [(_slot(l.norm1.weight), _slot(l.norm1.bias), _slot(l.norm2.weight), _slot(l.norm2.bias))
                     for l in stack.layers]
            dx, g = _seqstack_bwd_launch(dy2, sv, stack.wpack_bwd, stack.stride_bwd, slots, B, L, F_, stack.layers[0].act,
                                         n_top, drop_p, site0)
            for li in reversed(range(len(stack.layers))):
                lay = stack.layers[li]
                att, pk = lay.attention, lay.attention._packed
                _wrote(*slots[li])
                for gy, xin, w_into, b_into in (
                        (g["dpre2"][li], sv["h"][li], _slot(lay.conv2.weight).view(D, F_), _slot(lay.conv2.bias)),
                        (g["dz"][li], sv["x1"][li], _slot(lay.conv1.weight).view(F_, D), _slot(lay.conv1.bias)),
                        (g["dpre1"][li], sv["ctx"][li], _slot(att.out_projection.weight), _slot(att.out_projection.bias)),
                        (g["dqkv"][li], sv["xin"][li] if "xin" in sv else (x2 if li == 0 else sv["y"][li - 1]), pk["gw"],
                         pk["gb"])):
                    if _weight_grad(gy, xin, into=w_into, bias_into=b_into) is not True:
                        colsum(gy, into=b_into)
                    _wrote(w_into, b_into)
            ctx.sv = None
            return dx.view(B, L, D), None, None, None, None, None

        ctx.sv = None
        return _stack_backward_layerwise(sv, stack, x2, dy2, B, L, F_, n_top, drop_p, site0).view(B, L, D), None, None, None, None, None


def _stack_backward_layerwise(sv, stack, x2, dy2, B, L, F_, n_top, drop_p, site0):
    """Backward of a stack of ProbSparse encoder layers on the tensors its fused / row-tiled forward saved, layer by layer
    with the row-block and attention backward kernels (the unfused backward's launches).  -> d input (M, 128)."""
    M, D, H, E = B * L, 128, 8, 16
    if any(t.dtype == torch.bfloat16 for t in sv.values()):  # (BF16_SAVES of the fused forward: these kernels read fp32)
        sv = {k: (t.float() if t.dtype == torch.bfloat16 else t) for k, t in sv.items()}

    def masked(t, site):  # t * keep / (1 - p) with the mask the fused forward drew for `site` (new tensor)
        out = torch.empty_like(t)
        _drop_launch(t, out, drop_p, site, None)
        return out

    for li in reversed(range(len(stack.layers))):
        lay = stack.layers[li]
        x_in = x2 if li == 0 else (sv["xin"][li] if "xin" in sv else sv["y"][li - 1])  # (its weight gradient's operand)
        w1, w2 = lay.conv1.weight.reshape(F_, D), lay.conv2.weight.reshape(D, F_)
        zsrc = sv["z"][li] if "z" in sv else sv["h"][li]
        # ---- norm2 + conv pair (as _FFNAddLN.backward) ----
        gg, gbeta = _slot(lay.norm2.weight), _slot(lay.norm2.bias)
        xhat2, rstd2, h, x1 = sv["xhat2"][li], sv["rstd2"][li], sv["h"][li], sv["x1"][li]
        dpm = None  # gradient of the conv2 output: d(pre-norm) with the output-dropout mask (== dpre without dropout)
        if drop_p > 0.0:  # dropout sites sit between the products: unfused chain with the masks regenerated
            dpre, _, _ = _ln_backward(dy2, xhat2, rstd2, lay.norm2.weight, gg, gbeta)
            dpm = masked(dpre, site0 + 3 * li + 2)
            dz = _input_grad(dpm, w2, dact_src=zsrc, ldd=zsrc.stride(0), dact=ACT[lay.act])
            _drop_launch(dz, dz, drop_p, site0 + 3 * li + 1, None)
        elif _rowblock_nn_ok(w2, ln=True) and _rowblock_nn_ok(w1) and not DETERMINISTIC:
            dpre = torch.empty_like(xhat2)
            dz = _rowblock_nn(w2, M, ln=(dy2, xhat2, rstd2, lay.norm2.weight), dpre=dpre, dgam=gg, dbet=gbeta,
                              dsrc=zsrc, dact=ACT[lay.act])
            _wrote(gg, gbeta)
        else:
            dpre, _, _ = _ln_backward(dy2, xhat2, rstd2, lay.norm2.weight, gg, gbeta)
            dz = _input_grad(dpre, w2, dact_src=zsrc, ldd=zsrc.stride(0), dact=ACT[lay.act])
        g2, gb2, g1, gb1 = (_slot(lay.conv2.weight), _slot(lay.conv2.bias), _slot(lay.conv1.weight),
                            _slot(lay.conv1.bias))
        dpw = dpm if dpm is not None else dpre
        if _weight_grad(dpw, h, into=g2.view(D, F_), bias_into=gb2) is not True:
            colsum(dpw, into=gb2)
        if _weight_grad(dz, x1, into=g1.view(F_, D), bias_into=gb1) is not True:
            colsum(dz, into=gb1)
        if _rowblock_nn_ok(w1):
            dx1 = _rowblock_nn(w1, M, a=dz, res=dpre)
        else:
            dx1 = _input_grad(dz, w1, residual=dpre, ldr=D, res_rows=M)
        _wrote(g1, gb1, g2, gb2)
        # ---- norm1 + out-projection (as _LinearAddLN.backward) ----
        att = lay.attention
        wo = att.out_projection.weight
        gg, gbeta = _slot(lay.norm1.weight), _slot(lay.norm1.bias)
        xhat1, rstd1, ctx2 = sv["xhat1"][li], sv["rstd1"][li], sv["ctx"][li]
        dpo = None  # gradient of the out-projection output (masked by the attention-output dropout)
        if drop_p > 0.0:
            dpre1, _, _ = _ln_backward(dx1, xhat1, rstd1, lay.norm1.weight, gg, gbeta)
            dpo = masked(dpre1, site0 + 3 * li)
            dctx = _input_grad(dpo, wo)
        elif _rowblock_nn_ok(wo, ln=True) and not DETERMINISTIC:
            dpre1 = torch.empty_like(xhat1)
            dctx = _rowblock_nn(wo, M, ln=(dx1, xhat1, rstd1, lay.norm1.weight), dpre=dpre1, dgam=gg, dbet=gbeta)
            _wrote(gg, gbeta)
        else:
            dpre1, _, _ = _ln_backward(dx1, xhat1, rstd1, lay.norm1.weight, gg, gbeta)
            dctx = _input_grad(dpre1, wo)
        gwo, gbo = _slot(wo), _slot(att.out_projection.bias)
        dow = dpo if dpo is not None else dpre1
        if _weight_grad(dow, ctx2, into=gwo, bias_into=gbo) is not True:
            colsum(dow, into=gbo)
        _wrote(gwo, gbo)
        # ---- attention core ----
        qkv = sv["qkv"][li]
        dqkv = torch.empty_like(qkv)
        HE = H * E
        ev = PROFILE.begin() if PROFILE.on else None
        bargs = (qkv.data_ptr(), qkv.data_ptr() + 4 * HE, qkv.data_ptr() + 8 * HE, 3 * HE, 3 * HE, 3 * HE, ptr(dctx), 0,
                 ptr(sv["top"][li]), dqkv.data_ptr(), dqkv.data_ptr() + 4 * HE, dqkv.data_ptr() + 8 * HE, 3 * HE,
                 3 * HE, 3 * HE, B, H, L, L, E, n_top, 1, 1.0 / math.sqrt(E))
        check(_hip.lib().rf_attn_bwd(*bargs, _stream()), "rf_attn_bwd")
        if ev is not None:
            keep = (qkv, dctx, dqkv, sv)
            PROFILE.end("attn_bwd_kernel<true>", ev, B * H * 10.0 * n_top * L * E, 4.0 * B * H * E * 8 * L,
                        replay=lambda fa=bargs, k=keep: _hip.lib().rf_attn_bwd(*fa, _stream()))
        # ---- packed q | k | v projection (as _Linear.backward with the skip gradient folded in) ----
        pk = att._packed
        if _weight_grad(dqkv, x_in, into=pk["gw"], bias_into=pk["gb"]) is not True:
            colsum(dqkv, into=pk["gb"])
        if _rowblock_nn_ok(pk["w"]):
            dy2 = _rowblock_nn(pk["w"], M, a=dqkv, res=dpre1)
        else:
            dy2 = _input_grad(dqkv, pk["w"], residual=dpre1, ldr=D, res_rows=M)
        _wrote(pk["gw"], pk["gb"])
    return dy2


def tiled_stack_supported(L: int, d_model: int, n_heads: int, d_ff: int) -> bool:
    """Row-tiled encoder stack (csrc/enclayer.hip): sequences beyond the one-workgroup stack's L <= 80, bf16 mode."""
    return bool(TILED_STACK and _PRECISION == 1 and 2 <= L <= 320
                and _hip.lib().rf_enclayer_tile_supported(d_model, n_heads, d_ff))


TILED_STACK = os.environ.get("RF_TILED_STACK", "1") != "0"
TILED_STACK_BWD = os.environ.get("RF_TILED_STACK_BWD", "1") != "0"  # row-tile backward launches (else layer by layer)


class _TiledStack(torch.autograd.Function):
    """A stack of ProbSparse encoder layers whose sequences are too long for ``_SeqStack`` (the fusion `video_encoder`,
    L = 160 / 320): per layer ONE attention launch (rf_attn_fwd, needs the whole sequence) and ONE row-tile launch
    (rf_enclayer_tile_fwd: out-projection + LN1 + conv pair + LN2 + the NEXT layer's packed q | k | v) instead of four;
    the first q | k | v comes from a projection-only launch.  Same weight blobs as the fused stack.  Backward: the
    layer-by-layer kernels on the saved tensors (``_stack_backward_layerwise``)."""

    @staticmethod
    def forward(ctx, x, stack, idx_list, idx_group, save, drop_p=0.0):
        B, L, D = x.shape
        M, H, E = B * L, 8, 16
        x2 = x.reshape(M, D).contiguous()
        lay0 = stack.layers[0]
        F_ = lay0.conv1.weight.shape[0]
        n = len(stack.layers)
        sample_k, n_top = prob_sizes(L, L, lay0.attention.factor)
        dev = x.device
        f32 = dict(device=dev, dtype=torch.float32)
        gelu = lay0.act == "gelu"
        sv = {"y": torch.empty(n if save else 2, M, D, **f32), "qkv": torch.empty(n if save else 2, M, 3 * D, **f32),
              "ctx": torch.empty(n if save else 1, M, D, **f32), "top": torch.empty(n, B, H, n_top, device=dev, dtype=torch.int32)}
        if save:
            for name, width in (("xhat1", D), ("x1", D), ("xhat2", D), ("h", F_)) + ((("z", F_),) if gelu else ()):
                sv[name] = torch.empty(n, M, width, **f32)
            sv["rstd1"], sv["rstd2"] = torch.empty(n, M, **f32), torch.empty(n, M, **f32)
        lib = _hip.lib()
        wp, stride = stack.wpack.data_ptr(), stack.stride
        slab = (lambda name, li: ptr(sv[name][li]) if save else None)
        site0 = 0
        if drop_p > 0.0:  # three dropout sites per layer, numbered in the reference's call order (as _SeqStack does)
            site0 = RNG.site
            RNG.site += 3 * n
            if RNG.record is not None:
                for li in range(n):
                    for k, cols in enumerate((D, F_, D)):
                        RNG.record.append(RNG.materialise(site0 + 3 * li + k, (B, L, cols), drop_p, x.device))
        rng = ptr(RNG.state(dev)) if drop_p > 0.0 else None

        def tile(ctx_t, x_t, li, y_t, qkv_t):
            """row-tile launch of layer li (ctx_t None: projection only), q | k | v of layer li + 1 -> qkv_t (None: none)"""
            check(lib.rf_enclayer_tile_fwd(ptr(ctx_t), ptr(x_t), wp + li * stride if ctx_t is not None else None,
                                           wp + (li + 1) * stride if qkv_t is not None else None, ptr(y_t), ptr(qkv_t),
                                           slab("xhat1", li) if ctx_t is not None else None, slab("rstd1", li) if ctx_t is not None else None,
                                           slab("x1", li) if ctx_t is not None else None,
                                           (slab("z", li) if gelu else None) if ctx_t is not None else None,
                                           slab("h", li) if ctx_t is not None else None, slab("xhat2", li) if ctx_t is not None else None,
                                           slab("rstd2", li) if ctx_t is not None else None, M, D, H, F_, ACT[lay0.act],
                                           1 if (save and ctx_t is not None) else 0, lay0.norm1.eps, float(drop_p), rng,
                                           site0 + 3 * li, _stream()), "rf_enclayer_tile_fwd")

        qkv = sv["qkv"][0]
        check(lib.rf_enclayer_tile_fwd(None, ptr(x2), None, wp, None, ptr(qkv), None, None, None, None, None, None, None, M, D, H,
                                       F_, ACT[lay0.act], 0, lay0.norm1.eps, 0.0, None, 0, _stream()), "rf_enclayer_tile_fwd(projection)")
        x_in = x2
        for li in range(n):
            idx = idx_list[li]
            forced = None
            if TOPS.forced is not None:
                forced = TOPS.forced.pop(0).to(device=dev, dtype=torch.int32).contiguous()
                assert tuple(forced.shape) == (B, H, n_top)
            top = forced if forced is not None else sv["top"][li]
            c_t = sv["ctx"][li if save else 0]
            istride = idx.stride(0) if (idx.dim() == 3 and idx.shape[0] > 1) else 0
            fargs = (qkv.data_ptr(), qkv.data_ptr() + 4 * D, qkv.data_ptr() + 8 * D, 3 * D, 3 * D, 3 * D, ptr(c_t), 0, ptr(idx),
                     idx_group if idx.dim() == 3 else 0, istride if idx.dim() == 3 else 0)
            if forced is not None and TOPS.shadow is not None:  # free selection on the same (teacher-forced) q / k
                free = torch.empty(B, H, n_top, device=dev, dtype=torch.int32)
                check(lib.rf_attn_fwd(*fargs, ptr(free), 0, B, H, L, L, E, sample_k, n_top, 1, 1.0 / math.sqrt(E), _stream()),
                      "rf_attn_fwd(shadow)")
                TOPS.shadow.append(free)
            ev = PROFILE.begin() if PROFILE.on else None
            aargs = (*fargs, ptr(top), 1 if forced is not None else 0, B, H, L, L, E, sample_k, n_top, 1, 1.0 / math.sqrt(E))
            check(lib.rf_attn_fwd(*aargs, _stream()), "rf_attn_fwd")
            if ev is not None:
                full = lib.rf_attn_fwd_full_scores(B, H, L, L, E, sample_k, n_top, 1)
                PROFILE.end("attn_fwd_kernel<true, true>" if full else "attn_fwd_kernel<true, false>", ev,
                            B * H * (2.0 * L * sample_k * E + 4.0 * n_top * L * E), 4.0 * B * H * E * 4 * L + 4.0 * L * sample_k,
                            replay=lambda a=aargs, k=(qkv, c_t, idx, top): lib.rf_attn_fwd(*a, _stream()))
            if forced is not None:
                sv["top"][li].copy_(forced)
            if TOPS.record is not None:
                TOPS.record.append(sv["top"][li].clone())
            y_t = sv["y"][li if save else li % 2]
            qkv_next = None if li == n - 1 else sv["qkv"][li + 1 if save else (li + 1) % 2]
            ev = PROFILE.begin() if PROFILE.on else None
            tile(c_t, x_in, li, y_t, qkv_next)
            if ev is not None:
                PROFILE.end(f"enc_tile_fwd_kernel<{2 if M <= 4096 else 3}, {'true' if save else 'false'}>", ev,
                            2.0 * M * D * (D + 2 * F_ + (3 * D if qkv_next is not None else 0)),
                            4.0 * M * (D * 3 + (3 * D if qkv_next is not None else 0) + ((4 * D + 2 * F_) if save else 0)))
            x_in, qkv = y_t, qkv_next
        if save:
            ctx.sv, ctx.stack, ctx.x2, ctx.dims, ctx.drop = sv, stack, x2, (B, L, F_, n_top), (float(drop_p), site0)
        return x_in.view(B, L, D)

    @staticmethod
    def backward(ctx, dy):
        sv, stack, x2, (B, L, F_, n_top) = ctx.sv, ctx.stack, ctx.x2, ctx.dims
        drop_p, site0 = ctx.drop
        M, D, H, E = B * L, 128, 8, 16
        dy2 = dy.reshape(M, D).contiguous()
        ctx.sv = None
        if not (TILED_STACK_BWD and not DETERMINISTIC and stack.wpack_bwd is not None):
            return (_stack_backward_layerwise(sv, stack, x2, dy2, B, L, F_, n_top, drop_p, site0).view(B, L, D), None, None, None,
                    None, None)
        # ---- per layer: ONE row-tile launch (the next layer's q | k | v projection^T + skip, norm2 backward, conv pair^T, norm1
        # backward, out-projection^T) and ONE attention-backward launch; the weight gradients follow as grouped GEMMs ----
        lib, dev, n = _hip.lib(), dy2.device, len(stack.layers)
        f32 = dict(device=dev, dtype=torch.float32)
        g = {"dpre2": torch.empty(n, M, D, **f32), "dz": torch.empty(n, M, F_, **f32), "dpre1": torch.empty(n, M, D, **f32),
             "dqkv": torch.empty(n, M, 3 * D, **f32)}
        dctx = torch.empty(M, D, **f32)
        dx = torch.empty(M, D, **f32)
        # with dropout the skip gradient (unmasked d pre-norm-1) and the out-projection's weight-gradient operand (masked) differ
        skips = torch.empty(n, M, D, **f32) if drop_p > 0.0 else g["dpre1"]
        rng = ptr(RNG.state(dev)) if drop_p > 0.0 else None
        wt, st = stack.wpack_bwd.data_ptr(), stack.stride_bwd
        act = ACT[stack.layers[0].act]
        slots = [(_slot(l.norm1.weight), _slot(l.norm1.bias), _slot(l.norm2.weight), _slot(l.norm2.bias)) for l in stack.layers]
        zsrc = sv["z"] if "z" in sv else sv["h"]
        scale = 1.0 / math.sqrt(E)
        for li in reversed(range(n)):
            last = li == n - 1
            g1, b1, g2, b2 = slots[li]
            ev = PROFILE.begin() if PROFILE.on else None
            check(lib.rf_enclayer_tile_bwd(ptr(dy2) if last else None, None if last else ptr(g["dqkv"][li + 1]),
                                           None if last else ptr(skips[li + 1]), wt + li * st, None if last else wt + (li + 1) * st,
                                           ptr(sv["xhat1"][li]), ptr(sv["rstd1"][li]), ptr(zsrc[li]), ptr(sv["xhat2"][li]),
                                           ptr(sv["rstd2"][li]), ptr(g["dpre2"][li]), ptr(g["dz"][li]), ptr(g["dpre1"][li]), ptr(dctx),
                                           None, ptr(g1), ptr(b1), ptr(g2), ptr(b2), M, D, H, F_, act,
                                           ptr(skips[li]) if drop_p > 0.0 else None, float(drop_p), rng, site0 + 3 * li, _stream()),
                  "rf_enclayer_tile_bwd")
            if ev is not None:
                PROFILE.end(f"enc_tile_bwd_kernel<{2 if M <= 4096 else 3}>", ev, 2.0 * M * D * (D + 2 * F_ + (0 if last else 3 * D)),
                            4.0 * M * (D * 6 + 2 * F_ + (0 if last else 4 * D)))
            qkv, dq = sv["qkv"][li], g["dqkv"][li]
            ev = PROFILE.begin() if PROFILE.on else None
            bargs = (qkv.data_ptr(), qkv.data_ptr() + 4 * D, qkv.data_ptr() + 8 * D, 3 * D, 3 * D, 3 * D, ptr(dctx), 0,
                     ptr(sv["top"][li]), dq.data_ptr(), dq.data_ptr() + 4 * D, dq.data_ptr() + 8 * D, 3 * D, 3 * D, 3 * D, B, H, L, L, E,
                     n_top, 1, scale)
            check(lib.rf_attn_bwd(*bargs, _stream()), "rf_attn_bwd")
            if ev is not None:
                PROFILE.end("attn_bwd_kernel<true>", ev, B * H * 10.0 * n_top * L * E, 4.0 * B * H * E * 8 * L,
                            replay=lambda fa=bargs, k=(qkv, dq, dctx, sv): lib.rf_attn_bwd(*fa, _stream()))
        check(lib.rf_enclayer_tile_bwd(None, ptr(g["dqkv"][0]), ptr(skips[0]), None, wt, None, None, None, None, None, None, None,
                                       None, None, ptr(dx), None, None, None, None, M, D, H, F_, act, None, 0.0, None, 0, _stream()),
              "rf_enclayer_tile_bwd(projection)")
        for li in reversed(range(n)):
            lay = stack.layers[li]
            att, pk = lay.attention, lay.attention._packed
            _wrote(*slots[li])
            for gy, xin, w_into, b_into in (
                    (g["dpre2"][li], sv["h"][li], _slot(lay.conv2.weight).view(D, F_), _slot(lay.conv2.bias)),
                    (g["dz"][li], sv["x1"][li], _slot(lay.conv1.weight).view(F_, D), _slot(lay.conv1.bias)),
                    (g["dpre1"][li], sv["ctx"][li], _slot(att.out_projection.weight), _slot(att.out_projection.bias)),
                    (g["dqkv"][li], x2 if li == 0 else sv["y"][li - 1], pk["gw"], pk["gb"])):
                if _weight_grad(gy, xin, into=w_into, bias_into=b_into) is not True:
                    colsum(gy, into=b_into)
                _wrote(w_into, b_into)
        return dx.view(B, L, D), None, None, None, None, None
