"""GPU parity tests, kernel by kernel: the HIP path (through the C ABI / ctypes binding) against
CPU torch math and the CPU oracle on the same seeded inputs, plus the reference golden vectors for
attention.  fp32 mode is held to ~1e-5 relative; bf16-MFMA mode to 2e-2."""
import math
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import fro_err, golden, rel_err, t

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import routeformer_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _f32_mode():
    from routeformer_amd import kernels as K
    K.set_precision("f32")
    yield
    K.set_precision("f32")


def _g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.fixture
def request_restore():
    """Tests that widen the skinny-GEMM dispatch thresholds append the old pair here; restored afterwards."""
    from routeformer_amd import kernels as K
    saved = []
    yield saved
    for a, b in saved:
        K.SKINNY_MAX_M, K.SKINNY_MAX_M_DEEP = a, b


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (520, 128, 128), (37, 66, 207), (12480, 64, 128),
                                   (560, 3328, 832), (5, 3, 6), (130, 17, 33)])
@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_gemm_linear_layouts(M, N, K, prec):
    """All three operand layouts used by linear fwd / dX / dW, odd sizes, split-K."""
    from routeformer_amd import kernels as Kn
    Kn.set_precision(prec)
    tol = 2e-5 if prec == "f32" else 2e-2
    g = _g(M * 7 + N)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    dy = torch.randn(M, N, generator=g)
    xd, wd, bd, dyd = (a.to(DEV) for a in (x, w, b, dy))
    y = torch.empty(M, N, device=DEV)
    Kn.gemm(xd, K, 1, wd, 1, K, y, N, M, N, K, bias=bd)
    ref = x.double() @ w.double().t() + b.double()
    assert rel_err(y, ref) < tol
    dx = Kn._input_grad(dyd, wd)
    assert rel_err(dx, dy.double() @ w.double()) < tol
    dw = Kn._weight_grad(dyd, xd)
    assert rel_err(dw, dy.double().t() @ x.double()) < tol
    # explicit split-K on the forward product
    y2 = torch.empty(M, N, device=DEV)
    Kn.gemm(xd, K, 1, wd, 1, K, y2, N, M, N, K, bias=bd, splitk=3)
    assert rel_err(y2, ref) < tol
    # ... and with the in-launch reduction (last-arriving workgroup sums the slabs; off by default, slower)
    was = Kn.IN_LAUNCH_SPLITK_REDUCE
    Kn.IN_LAUNCH_SPLITK_REDUCE = True
    try:
        y3 = torch.empty(M, N, device=DEV)
        for _ in range(2):  # twice: the arrival counters must be back at zero after a launch
            y3.fill_(float("nan"))
            Kn.gemm(xd, K, 1, wd, 1, K, y3, N, M, N, K, bias=bd, splitk=3)
            assert rel_err(y3, ref) < tol
        if K >= 192:
            assert torch.equal(y3, y2)  # slabs are summed in slice order either way: same bits
    finally:
        Kn.IN_LAUNCH_SPLITK_REDUCE = was
    assert rel_err(Kn.colsum(dyd), dy.double().sum(0)) < 1e-5
    # gradient-sink form: dW and db accumulated with fp32 atomics into pre-filled slots, one launch
    if N % 4 == 0 and K % 4 == 0:
        slot_w = torch.full((N, K), 0.5, device=DEV)
        slot_b = torch.full((N,), -0.25, device=DEV)
        assert Kn._weight_grad(dyd, xd, into=slot_w, bias_into=slot_b) is True
        assert rel_err(slot_w, 0.5 + dy.double().t() @ x.double()) < tol
        assert rel_err(slot_b, -0.25 + dy.double().sum(0)) < 1e-4


@pytest.mark.parametrize("act", ["relu", "gelu"])
def test_gemm_epilogues(act):
    from routeformer_amd import kernels as Kn
    g = _g(3)
    M, N, K = 200, 96, 64
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    res = torch.randn(40, N, generator=g)
    xd, wd, bd, rd = (a.to(DEV) for a in (x, w, b, res))
    y, z = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    Kn.gemm(xd, K, 1, wd, 1, K, y, N, M, N, K, bias=bd, act=Kn.ACT[act], preact=z, ldp=N, residual=rd, ldr=N,
            res_rows=40)
    zr = x @ w.t() + b
    a = F.relu(zr) if act == "relu" else F.gelu(zr)
    assert rel_err(z, zr) < 2e-5
    assert rel_err(y, a + res.repeat(5, 1)) < 2e-5
    # derivative epilogue: C = (X W^T) * act'(src)
    src = torch.randn(M, N, generator=g)
    y2 = torch.empty(M, N, device=DEV)
    src_d = src.to(DEV)
    Kn.gemm(xd, K, 1, wd, 1, K, y2, N, M, N, K, dact_src=src_d, ldd=N, dact=Kn.ACT[act])
    s = src.clone().requires_grad_()
    (F.relu(s) if act == "relu" else F.gelu(s)).sum().backward()
    assert rel_err(y2, (x @ w.t()) * s.grad) < 2e-5


@pytest.mark.parametrize("M,N,K", [(32, 832, 832), (40, 3328, 832), (56, 832, 3328), (96, 2496, 832), (168, 832, 2496),
                                   (320, 3328, 832), (560, 832, 832), (7, 48, 72), (33, 100, 1032), (64, 16, 64)])
@pytest.mark.parametrize("bmode", [0, 1])
def test_gemm_skinny(M, N, K, bmode, request_restore):
    """The skinny GEMM of the GPS backbone's launch shapes (csrc/gemm_skinny.hip: all operands requested up front,
    8 waves interleaving the k-steps, in-launch epilogue; K > 1024: slices + slab sum) against fp32 torch math on the
    bf16-rounded operands (the kernel's arithmetic contract: bf16 inputs, fp32 accumulation) -- both weight orientations
    (y = x W^T and dX = dY W), ragged M / N, every epilogue."""
    from routeformer_amd import _hip, kernels as Kn
    Kn.set_precision("bf16")
    monkey = (Kn.SKINNY_MAX_M, Kn.SKINNY_MAX_M_DEEP)
    Kn.SKINNY_MAX_M = Kn.SKINNY_MAX_M_DEEP = 640  # (the dispatch thresholds are a speed matter: test the kernel's whole range)
    request_restore.append(monkey)
    g = _g(5)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)   # logical W[n][k]
    b = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    src = torch.randn(M, N, generator=g)
    xd, bd, rd, sd = (a.to(DEV) for a in (x, b, res, src))
    wd = (w if bmode == 0 else w.t().contiguous()).to(DEV)  # bmode 1: stored [k][n]
    ldb = (1, K) if bmode == 0 else (N, 1)
    assert _hip.lib().rf_gemm_skinny_split(xd.data_ptr(), K, 1, wd.data_ptr(), ldb[0], ldb[1], M, N, K) == -(-K // 1024)
    xr, wr = x.bfloat16().float(), w.bfloat16().float()
    zr = xr @ wr.t()
    y = torch.empty(M, N, device=DEV)
    Kn.gemm(xd, K, 1, wd, ldb[0], ldb[1], y, N, M, N, K)
    assert rel_err(y, zr) < 1e-5
    # bias + residual-before-activation + relu + pre-activation output
    y1, z1 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    Kn.gemm(xd, K, 1, wd, ldb[0], ldb[1], y1, N, M, N, K, bias=bd, residual=rd, ldr=N, res_rows=M, res_before_act=1,
            act=Kn.ACT["relu"], preact=z1, ldp=N)
    assert rel_err(z1, zr + b + res) < 1e-5 and rel_err(y1, F.relu(zr + b + res)) < 1e-5
    # activation' epilogue + residual after it (the dX form: dX = (dY W) * act'(src) + skip)
    y2 = torch.empty(M, N, device=DEV)
    Kn.gemm(xd, K, 1, wd, ldb[0], ldb[1], y2, N, M, N, K, dact_src=sd, ldd=N, dact=Kn.ACT["relu"], residual=rd, ldr=N,
            res_rows=M)
    assert rel_err(y2, zr * (src > 0).float() + res) < 1e-5
    # and the tiled kernel agrees (same rounding contract)
    Kn.SKINNY_GEMM = False
    try:
        y3 = torch.empty(M, N, device=DEV)
        Kn.gemm(xd, K, 1, wd, ldb[0], ldb[1], y3, N, M, N, K)
    finally:
        Kn.SKINNY_GEMM = True
    assert rel_err(y3, y) < 1e-5


@pytest.mark.parametrize("prec,tol", [("f32", 3e-5), ("bf16", 5e-2)])
def test_linear_ffn_autograd(prec, tol):
    from routeformer_amd import kernels as Kn
    Kn.set_precision(prec)
    rel_err = globals()["rel_err"] if prec == "f32" else fro_err  # bf16 rounding flips a few ReLU masks
    g = _g(11)
    for act in ("relu", "gelu"):
        x = torch.randn(3, 65, 128, generator=g)
        w1, b1 = torch.randn(256, 128, 1, generator=g) / 11, torch.randn(256, generator=g) * 0.1  # Conv1d(k=1)
        w2, b2 = torch.randn(128, 256, 1, generator=g) / 16, torch.randn(128, generator=g) * 0.1
        cpu = [a.clone().requires_grad_() for a in (x, w1, b1, w2, b2)]
        dev = [a.clone().to(DEV).requires_grad_() for a in (x, w1, b1, w2, b2)]
        yc = F.linear((F.relu if act == "relu" else F.gelu)(F.linear(cpu[0], cpu[1].squeeze(-1), cpu[2])),
                      cpu[3].squeeze(-1), cpu[4])
        yd = Kn.ffn(dev[0], dev[1], dev[2], dev[3], dev[4], act)
        wgt = torch.randn(yc.shape, generator=g)
        (yc * wgt).sum().backward()
        (yd * wgt.to(DEV)).sum().backward()
        assert rel_err(yd, yc) < tol
        for c, d in zip(cpu, dev):
            assert rel_err(d.grad, c.grad) < tol
    x = torch.randn(4, 40, 69, generator=g)
    w, b = torch.randn(832, 69, generator=g) / 8, torch.randn(832, generator=g)
    cpu = [a.clone().requires_grad_() for a in (x, w, b)]
    dev = [a.clone().to(DEV).requires_grad_() for a in (x, w, b)]
    yc, yd = F.linear(*cpu), Kn.linear(*dev)
    yc.square().sum().backward()
    yd.square().sum().backward()
    assert rel_err(yd, yc) < tol
    for c, d in zip(cpu, dev):
        assert rel_err(d.grad, c.grad) < tol


@pytest.mark.parametrize("rows,cols", [(7, 64), (520, 128), (12480, 128), (560, 832), (3, 1024)])
def test_layernorm(rows, cols):
    from routeformer_amd import kernels as Kn
    g = _g(rows + cols)
    x, r = torch.randn(rows, cols, generator=g) * 2 + 0.3, torch.randn(rows, cols, generator=g)
    w, b = torch.randn(cols, generator=g), torch.randn(cols, generator=g)
    cpu = [a.clone().requires_grad_() for a in (x, r, w, b)]
    dev = [a.clone().to(DEV).requires_grad_() for a in (x, r, w, b)]
    yc = F.layer_norm(cpu[0] + cpu[1], (cols,), cpu[2], cpu[3], 1e-5)
    yd = Kn.add_layer_norm(dev[0], dev[1], dev[2], dev[3])
    wt = torch.randn(rows, cols, generator=g)
    (yc * wt).sum().backward()
    (yd * wt.to(DEV)).sum().backward()
    assert rel_err(yd, yc) < 1e-5
    for c, d in zip(cpu, dev):
        assert rel_err(d.grad, c.grad) < 3e-5
    y1 = Kn.add_layer_norm(dev[0].detach(), None, dev[2].detach(), dev[3].detach())
    assert rel_err(y1, F.layer_norm(x, (cols,), w, b, 1e-5)) < 1e-5


@pytest.mark.parametrize("B,L,C,D,pad", [(3, 65, 240, 128, 1), (2, 40, 2, 128, 1), (4, 40, 69, 64, 1),
                                         (2, 21, 64, 64, 2), (3, 4, 32, 32, 2), (2, 5, 16, 16, 2)])
def test_circular_conv3(B, L, C, D, pad):
    from routeformer_amd import kernels as Kn
    g = _g(B * L + C)
    x, w, b = torch.randn(B, L, C, generator=g), torch.randn(D, C, 3, generator=g) / 4, torch.randn(D, generator=g)
    cpu = [a.clone().requires_grad_() for a in (x, w, b)]
    dev = [a.clone().to(DEV).requires_grad_() for a in (x, w, b)]
    yc = F.conv1d(F.pad(cpu[0].transpose(1, 2), (pad, pad), mode="circular"), cpu[1], cpu[2]).transpose(1, 2)
    yd = Kn.circular_conv3(dev[0], dev[1], dev[2], pad=pad)
    assert yd.shape == yc.shape
    wt = torch.randn(yc.shape, generator=g)
    (yc * wt).sum().backward()
    (yd * wt.to(DEV)).sum().backward()
    assert rel_err(yd, yc) < 2e-5
    assert rel_err(yd, O.circular_conv3(x, w, b, padding=pad)) < 2e-5
    for c, d in zip(cpu, dev):
        assert rel_err(d.grad, c.grad) < 3e-5


# (40 x 42 rows: beyond the LDS slab of the one-launch kernels -> the statistics + apply launches / the strided backward)
@pytest.mark.parametrize("B,L,C", [(4, 42, 64), (8, 23, 832), (8, 42, 832), (2, 7, 16), (3, 6, 128), (40, 42, 64)])
@pytest.mark.parametrize("training", [True, False])
def test_bn_elu_pool(B, L, C, training):
    from routeformer_amd import kernels as Kn
    g = _g(B + L + C)
    x = torch.randn(B, L, C, generator=g) * 1.5
    gam, bet = 1 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    rm, rv = 0.1 * torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    bn = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        bn.weight.copy_(gam); bn.bias.copy_(bet); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
    bn.train(training)
    xc = x.clone().requires_grad_()
    yc = F.max_pool1d(F.elu(bn(xc.transpose(1, 2))), 3, 2, 1).transpose(1, 2)
    dev = [a.clone().to(DEV).requires_grad_() for a in (x, gam, bet)]
    rmd, rvd, nbt = rm.clone().to(DEV), rv.clone().to(DEV), torch.zeros((), dtype=torch.long, device=DEV)
    yd = Kn.bn_elu_pool(dev[0], dev[1], dev[2], rmd, rvd, nbt, training)
    wt = torch.randn(yc.shape, generator=g)
    (yc * wt).sum().backward()
    (yd * wt.to(DEV)).sum().backward()
    assert rel_err(yd, yc) < 2e-5
    assert rel_err(dev[0].grad, xc.grad) < 5e-5
    assert rel_err(dev[1].grad, bn.weight.grad) < 5e-5 and rel_err(dev[2].grad, bn.bias.grad) < 5e-5
    if training:
        assert rel_err(rmd, bn.running_mean) < 1e-5 and rel_err(rvd, bn.running_var) < 1e-5 and int(nbt) == 1


# ------------------------------------------------------------------------------------------------
def _run_attn(q, k, v, mode, idx, factor, layout, forced_top=None):
    """q,k,v (B,L,H,E) CPU leaf tensors -> (ctx, dq, dk, dv, top) from the HIP kernels."""
    from routeformer_amd import kernels as Kn
    B, LQ, H, E = q.shape
    LK = k.shape[1]
    qd = q.detach().reshape(B * LQ, H * E).to(DEV).requires_grad_()
    kv = torch.cat([k.detach().reshape(B * LK, H * E), v.detach().reshape(B * LK, H * E)], dim=1).to(DEV).requires_grad_()
    n_top = 0
    if mode != 0:
        _, n_top = Kn.prob_sizes(LQ, LK, factor)
    idx_d = None if idx is None else idx.to(torch.int32).to(DEV)
    ctx = Kn.attention(qd, kv, (0, 0, H * E), (B, H, LQ, LK, E), mode, index_sample=idx_d, n_top=n_top,
                       out_layout=layout, forced_top=forced_top)
    return ctx, qd, kv


ATTN_TAGS = ["frame", "fusion", "decself", "gps_enc", "gps_enc5", "gps_decself", "gps_deccross", "gps_def_cross"]


@pytest.mark.parametrize("tag", ATTN_TAGS)
def test_prob_attention_golden(tag):
    """Against the REFERENCE's outputs and input gradients (tests/golden/attention.npz)."""
    G = golden("attention")
    LQ, LK, H, E, masked, factor, gps = (int(x) for x in G[tag + ".meta"])
    g = _g(100 + LQ * 7 + LK)
    q, k, v = (torch.randn(2, L, H, E, generator=g) for L in (LQ, LK, LK))
    ctx, qd, kv = _run_attn(q, k, v, 2 if masked else 1, t(G[tag + ".idx"]), factor, gps)
    w = torch.randn(ctx.shape, generator=g)
    (ctx * w.to(DEV)).sum().backward()
    assert rel_err(ctx, G[tag + ".ctx"]) < 2e-5
    HE = H * E
    assert rel_err(qd.grad.view(2, LQ, H, E), G[tag + ".dq"]) < 3e-5
    assert rel_err(kv.grad[:, :HE].reshape(2, LK, H, E), G[tag + ".dk"]) < 3e-5
    assert rel_err(kv.grad[:, HE:].reshape(2, LK, H, E), G[tag + ".dv"]) < 3e-5


def test_full_attention_golden():
    G = golden("attention")
    g = _g(55)
    q, k, v = (torch.randn(2, 40, 8, 8, generator=g) for _ in range(3))
    ctx, qd, kv = _run_attn(q, k, v, 0, None, 5, 0)
    w = torch.randn(ctx.shape, generator=g)
    (ctx * w.to(DEV)).sum().backward()
    assert rel_err(ctx, G["full.ctx"]) < 2e-5
    assert rel_err(qd.grad.view(2, 40, 8, 8), G["full.dq"]) < 3e-5
    assert rel_err(kv.grad[:, :64].reshape(2, 40, 8, 8), G["full.dk"]) < 3e-5
    assert rel_err(kv.grad[:, 64:].reshape(2, 40, 8, 8), G["full.dv"]) < 3e-5


@pytest.mark.parametrize("LQ,LK,H,E,masked,factor,gps", [
    (320, 320, 8, 16, False, 5, 0), (80, 80, 8, 8, True, 5, 0), (105, 105, 8, 104, True, 4, 1),
    (105, 5, 8, 104, False, 4, 1), (2, 2, 2, 8, False, 5, 0),
    (40, 30, 8, 8, None, 5, 0), (80, 25, 8, 8, None, 5, 0), (10, 10, 8, 16, False, 1, 1), (6, 6, 8, 16, True, 1, 1)])
def test_attention_vs_oracle(LQ, LK, H, E, masked, factor, gps):
    """Long-horizon (C5) and edge shapes against the CPU oracle; masked=None means full attention."""
    B = 3
    g = _g(LQ * 3 + LK + E)
    q, k, v = (torch.randn(B, L, H, E, generator=g).requires_grad_() for L in (LQ, LK, LK))
    if masked is None:
        ref = O.full_attention(q, k, v)
        idx, mode = None, 0
    else:
        sample_k, _ = O.prob_sizes(LQ, LK, factor)
        idx = torch.randint(LK, (LQ, sample_k), generator=g)
        ref, top_ref = O.prob_attention(q, k, v, idx, factor, masked, gps_variant=bool(gps), return_top=True)
        mode = 2 if masked else 1
    ctx, qd, kv = _run_attn(q, k, v, mode, idx, factor, gps)
    w = torch.randn(ref.shape, generator=g)
    (ref * w).sum().backward()
    (ctx * w.to(DEV)).sum().backward()
    assert rel_err(ctx, ref) < 3e-5
    HE = H * E
    assert rel_err(qd.grad.view(B, LQ, H, E), q.grad) < 5e-5
    assert rel_err(kv.grad[:, :HE].reshape(B, LK, H, E), k.grad) < 5e-5
    assert rel_err(kv.grad[:, HE:].reshape(B, LK, H, E), v.grad) < 5e-5


def test_attention_packed_self_and_forced_top():
    """Packed QKV buffer (one projection GEMM) + injected selection (forced_top)."""
    from routeformer_amd import kernels as Kn
    B, L, H, E, factor = 2, 65, 8, 16, 5
    g = _g(9)
    qkv = torch.randn(B * L, 3 * H * E, generator=g)
    q, k, v = (qkv[:, i * H * E:(i + 1) * H * E].reshape(B, L, H, E).clone().requires_grad_() for i in range(3))
    sample_k, n_top = O.prob_sizes(L, L, factor)
    idx = torch.randint(L, (L, sample_k), generator=g)
    ref, top = O.prob_attention(q, k, v, idx, factor, False, return_top=True)
    qd = qkv.clone().to(DEV).requires_grad_()
    idx_d = idx.to(torch.int32).to(DEV)
    ctx = Kn.attention(qd, qd, (0, H * E, 2 * H * E), (B, H, L, L, E), 1, index_sample=idx_d, n_top=n_top)
    w = torch.randn(ref.shape, generator=g)
    (ref * w).sum().backward()
    (ctx * w.to(DEV)).sum().backward()
    assert rel_err(ctx, ref) < 3e-5
    packed_ref = torch.cat([a.grad.reshape(B * L, H * E) for a in (q, k, v)], dim=1)
    assert rel_err(qd.grad, packed_ref) < 5e-5
    # forced selection: give the kernel a *different* (sorted) top set and compare with the oracle run on it
    forced = torch.stack([torch.randperm(L, generator=g)[:n_top].sort().values for _ in range(B * H)]).view(B, H, n_top)
    forced_d, qdd = forced.to(torch.int32).to(DEV), qd.detach()
    ctx2 = Kn.attention(qdd, qdd, (0, H * E, 2 * H * E), (B, H, L, L, E), 1, n_top=n_top, forced_top=forced_d)
    Q, Km, V = (a.detach().transpose(1, 2) for a in (q, k, v))
    exp = V.mean(2, keepdim=True).expand(B, H, L, E).clone()
    qs = torch.gather(Q, 2, forced.unsqueeze(-1).expand(-1, -1, -1, E))
    upd = torch.softmax(qs @ Km.transpose(-1, -2) / math.sqrt(E), -1) @ V
    exp = exp.scatter(2, forced.unsqueeze(-1).expand(-1, -1, -1, E), upd).transpose(1, 2)
    assert rel_err(ctx2, exp) < 3e-5


@pytest.mark.parametrize("B,groups,L,E,masked,factor", [
    (192, 3, 65, 16, False, 5),   # 1 536 problems: the frame encoder of the bench step (256-thread form, XCD order)
    (336, 3, 65, 16, False, 5),   # 2 688 problems: history + target frames in one pass (256-thread form)
    (96, 2, 65, 16, False, 5),    # 768 problems: 256-thread form, B % 8 == 0
    (40, 1, 40, 8, True, 5),      # 320 problems: 512-thread form, masked (decoder self-attention shape), XCD order
    (36, 2, 65, 16, False, 5),    # 288 problems: 512-thread form, B % 8 != 0 (plain problem order)
    (67, 1, 40, 16, False, 5),    # 536 problems: 256-thread form, plain problem order
])
def test_attention_launch_shapes_vs_oracle(B, groups, L, E, masked, factor):
    """Every workgroup size rf_attn_fwd / rf_attn_bwd choose (attention.hip threads_for: 1024 / 512 / 256 threads by
    problem count) and both problem orders (XCD-aware when B % 8 == 0), at the chip-filling launch shapes bench.py
    times -- ctx, dq, dk, dv against the CPU oracle, with grouped key-sample tables as the stream-batched frame
    encoder passes them (cross_modal_transformer.py:88-166)."""
    from routeformer_amd import kernels as Kn
    H = 8
    g = _g(B + L + E)
    q, k, v = (torch.randn(B, L, H, E, generator=g).requires_grad_() for _ in range(3))
    sample_k, n_top = O.prob_sizes(L, L, factor)
    per = B // groups
    assert per * groups == B
    idx = torch.randint(L, (groups, L, sample_k), generator=g)
    refs = [O.prob_attention(q[i * per:(i + 1) * per], k[i * per:(i + 1) * per], v[i * per:(i + 1) * per], idx[i], factor,
                             masked) for i in range(groups)]
    ref = torch.cat(refs)
    qkv = torch.cat([a.detach().reshape(B * L, H * E) for a in (q, k, v)], dim=1).to(DEV).requires_grad_()
    ctx = Kn.attention(qkv, qkv, (0, H * E, 2 * H * E), (B, H, L, L, E), 2 if masked else 1,
                       index_sample=idx.to(torch.int32).to(DEV), n_top=n_top, idx_group=per)
    w = torch.randn(ref.shape, generator=g)
    (ref * w).sum().backward()
    (ctx * w.to(DEV)).sum().backward()
    assert rel_err(ctx, ref) < 3e-5
    packed_ref = torch.cat([a.grad.reshape(B * L, H * E) for a in (q, k, v)], dim=1)
    assert rel_err(qkv.grad, packed_ref) < 3e-5


@pytest.mark.parametrize("L,E,masked", [(40, 104, False), (33, 104, True), (65, 16, False), (42, 104, True), (160, 16, False),
                                        (105, 104, True)])
def test_attention_full_score_form_matches_compact_form(L, E, masked):
    """rf_attn_fwd keeps the whole Q K^T in LDS when the launch's LDS budget allows (up to 64 KB per problem for <= 128
    (batch, head) problems -- RF_ATTN_FULLS_KB --, up to 32 KB for chip-filling launches) and streams the sampled scores
    otherwise: both forms must select the same queries and produce the same context.  Shapes that take one form at both
    launch sizes under the current budgets are skipped."""
    from routeformer_amd import _hip, kernels as Kn
    B, H, factor = 20, 8, 5
    g = _g(L + E)
    sample_k, n_top = O.prob_sizes(L, L, factor)
    mode = 2 if masked else 1
    small, big = (_hip.lib().rf_attn_fwd_full_scores(b, H, L, L, E, sample_k, n_top, mode) for b in (B // 2, B))
    if (small, big) == (0, 0) or (small, big) == (1, 1):
        pytest.skip(f"one form for both launch sizes at L={L}, E={E} (full-score form: {small})")
    qkv = torch.randn(B * L, 3 * H * E, generator=g).to(DEV)
    idx = torch.randint(L, (L, sample_k), generator=g).to(torch.int32).to(DEV)
    offs, tops = (0, H * E, 2 * H * E), []
    Kn.TOPS.record = tops
    try:
        whole = Kn.attention(qkv, qkv, offs, (B, H, L, L, E), mode, index_sample=idx, n_top=n_top)
        halves = [Kn.attention(part, part, offs, (B // 2, H, L, L, E), mode, index_sample=idx, n_top=n_top)
                  for part in (qkv[:B // 2 * L], qkv[B // 2 * L:])]
    finally:
        Kn.TOPS.record = None
    assert torch.equal(tops[0], torch.cat(tops[1:]))
    assert rel_err(torch.cat(halves), whole) < 1e-5


# ------------------------------------------------------------------------------------------------
def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("N,H,W,cin,cout,k,s", [(3, 28, 28, 16, 16, 3, 1), (2, 28, 28, 256, 32, 3, 2),
                                                (2, 28, 28, 64, 256, 1, 1), (2, 56, 56, 4, 64, 3, 2),
                                                (5, 7, 7, 64, 64, 3, 1), (3, 4, 4, 128, 128, 3, 1),
                                                (2, 14, 14, 32, 64, 3, 2), (2, 4, 4, 128, 16, 1, 1),
                                                (1, 1, 1, 128, 128, 3, 1), (2, 2, 2, 64, 128, 3, 2)])
@pytest.mark.parametrize("prec,tol", [("f32", 2e-5), ("bf16", 2e-2), ("bf16_maps", 2e-2)])
def test_conv2d_nhwc(N, H, W, cin, cout, k, s, prec, tol):
    """prec "bf16_maps": input, residual and output maps stored in bf16 (RF_ACT_BF16), as the trunk keeps them in the
    bf16 matrix-core mode; must agree with the fp32-map launch on the same (bf16-representable) data to rounding."""
    from routeformer_amd import _hip, kernels as Kn
    maps_bf16 = prec == "bf16_maps"
    Kn.set_precision("bf16" if maps_bf16 else prec)
    g = _g(H + cin + cout)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g)
    pad = 1 if k == 3 else 0
    res = torch.randn(F.conv2d(x, w, b, stride=s, padding=pad).shape, generator=g)
    if maps_bf16:
        x, res = x.bfloat16().float(), res.bfloat16().float()
    ref = F.relu(F.conv2d(x, w, b, stride=s, padding=pad) + res)
    Ho, Wo = ref.shape[-2:]
    xd, rd, bd = _nhwc(x).to(DEV), _nhwc(res).to(DEV), b.to(DEV)  # keep every operand alive across the launch
    wd = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = torch.empty(N, Ho, Wo, cout, device=DEV)
    _hip.check(_hip.lib().rf_conv2d_nhwc(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), rd.data_ptr(),
                                         y.data_ptr(), 0, N, H, W, cin, cout, k, s, pad, Ho, Wo, cout, cout, 1,
                                         Kn._PRECISION, Kn._stream()), "conv")
    assert rel_err(y, _nhwc(ref)) < tol
    if maps_bf16:
        xb, rb = xd.bfloat16(), rd.bfloat16()
        yb = torch.empty(N, Ho, Wo, cout, device=DEV, dtype=torch.bfloat16)
        _hip.check(_hip.lib().rf_conv2d_nhwc(xb.data_ptr(), wd.data_ptr(), bd.data_ptr(), rb.data_ptr(),
                                             yb.data_ptr(), 1, N, H, W, cin, cout, k, s, pad, Ho, Wo, cout, cout, 1,
                                             Kn._PRECISION, Kn._stream()), "conv bf16 maps")
        assert torch.equal(yb, y.bfloat16())  # same accumulation, one rounding at the store
        # bf16 maps need the bf16 matrix-core path
        assert _hip.lib().rf_conv2d_nhwc(xb.data_ptr(), wd.data_ptr(), bd.data_ptr(), rb.data_ptr(), yb.data_ptr(), 1,
                                         N, H, W, cin, cout, k, s, pad, Ho, Wo, cout, cout, 1, 0, Kn._stream()) != 0


@pytest.mark.parametrize("N,H,W,cin,cout", [(3, 28, 28, 16, 16), (5, 14, 14, 32, 32), (7, 7, 7, 64, 64), (9, 4, 4, 128, 128),
                                            (2, 28, 28, 256, 16), (1, 1, 1, 128, 128), (3, 2, 3, 64, 64), (2, 56, 56, 64, 64),
                                            (11, 5, 9, 16, 16), (1, 56, 56, 256, 16), (3, 3, 5, 256, 16)])
def test_conv3x3_raster_window(N, H, W, cin, cout):
    """The LDS raster-window 3x3 kernel (bf16 MFMA) against F.conv2d on bf16-rounded operands."""
    from routeformer_amd import _hip, kernels as Kn
    g = _g(N * 31 + H + cin)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g)
    res = torch.randn(N, cout, H, W, generator=g)
    xb, wb_ = x.bfloat16().float(), w.bfloat16().float()
    ref = F.relu(F.conv2d(xb, wb_, b, padding=1) + res)
    xd, rd, bd = _nhwc(x).to(DEV), _nhwc(res).to(DEV), b.to(DEV)
    from routeformer_amd.models.video_backbone.hrnet16 import pack_conv3x3_weights
    wd = pack_conv3x3_weights(w.permute(0, 2, 3, 1).contiguous().to(DEV))  # MFMA fragment order, bf16
    y = torch.empty(N, H, W, cout, device=DEV)
    assert _hip.lib().rf_conv3x3_bf16_supported(cin, cout) == 1
    _hip.check(_hip.lib().rf_conv3x3_bf16(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), rd.data_ptr(), y.data_ptr(),
                                          0, N, H, W, cin, cout, 1, Kn._stream()), "conv3x3")
    assert rel_err(y, _nhwc(ref)) < 2e-5  # identical operands (bf16-rounded), fp32 accumulate
    _hip.check(_hip.lib().rf_conv3x3_bf16(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), None, y.data_ptr(),
                                          0, N, H, W, cin, cout, 0, Kn._stream()), "conv3x3")
    assert rel_err(y, _nhwc(F.conv2d(xb, wb_, b, padding=1))) < 2e-5
    # bf16 maps (RF_ACT_BF16): same sums from a bf16 input / residual, one rounding at the store
    xh, rh = xd.bfloat16(), rd.bfloat16()
    yh = torch.empty(N, H, W, cout, device=DEV, dtype=torch.bfloat16)
    _hip.check(_hip.lib().rf_conv3x3_bf16(xh.data_ptr(), wd.data_ptr(), bd.data_ptr(), rh.data_ptr(), yh.data_ptr(),
                                          1, N, H, W, cin, cout, 1, Kn._stream()), "conv3x3 bf16 maps")
    _hip.check(_hip.lib().rf_conv3x3_bf16(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), rh.float().data_ptr(), y.data_ptr(),
                                          0, N, H, W, cin, cout, 1, Kn._stream()), "conv3x3")
    assert torch.equal(yh, y.bfloat16())


@pytest.mark.parametrize("shapes", [[(3, 28, 28, 16)], [(2, 56, 56, 16)], [(5, 14, 14, 32)], [(7, 7, 7, 64)], [(9, 4, 4, 128)],
                                    [(1, 1, 1, 128)], [(2, 3, 5, 32)], [(2, 9, 130, 16)],
                                    [(6, 14, 14, 32), (6, 7, 7, 64), (6, 4, 4, 128)], [(3, 28, 28, 32), (3, 14, 14, 64)]])
def test_conv3x3_basicblock_pair(shapes):
    """rf_conv3x3_pair_group_bf16 (round 4: relu(conv2(relu(conv1 x + b1)) + b2 + x) of hrnetv2.py:45-61 in one launch, the
    intermediate map in LDS) == two rf_conv3x3_bf16 launches with a bf16 intermediate map, BIT for bit (same fragments, same k
    order, same roundings) -- single maps and the grouped form, tiles that straddle rows and images, maps narrower than a tile --
    and both against torch's conv2d on the bf16-rounded operands."""
    from routeformer_amd import _hip, kernels as Kn
    from routeformer_amd.models.video_backbone.hrnet16 import pack_conv3x3_weights
    g = _g(sum(sum(sh) for sh in shapes))
    arr = (_hip.ConvPairEntry * len(shapes))()
    keep, outs, want, refs = [], [], [], []
    for i, (N, H, W, C) in enumerate(shapes):
        assert _hip.lib().rf_conv3x3_pair_supported(C, W) == 1
        x = torch.randn(N, H, W, C, generator=g).bfloat16().to(DEV)
        ws = [(torch.randn(C, 3, 3, C, generator=g) / math.sqrt(9 * C)).to(DEV) for _ in range(2)]
        bs = [(torch.randn(C, generator=g) * 0.2).to(DEV) for _ in range(2)]
        wp = [pack_conv3x3_weights(w) for w in ws]
        y = torch.full((N, H, W, C), float("nan"), device=DEV, dtype=torch.bfloat16)
        e = arr[i]
        e.x, e.w1_packed, e.bias1, e.w2_packed, e.bias2, e.y = (x.data_ptr(), wp[0].data_ptr(), bs[0].data_ptr(), wp[1].data_ptr(),
                                                                bs[1].data_ptr(), y.data_ptr())
        e.N, e.H, e.W, e.c = N, H, W, C
        # the two-launch form
        mid, two = torch.empty_like(x), torch.empty_like(x)
        _hip.check(_hip.lib().rf_conv3x3_bf16(x.data_ptr(), wp[0].data_ptr(), bs[0].data_ptr(), None, mid.data_ptr(), 1, N, H, W, C, C, 1,
                                              Kn._stream()), "conv1")
        _hip.check(_hip.lib().rf_conv3x3_bf16(mid.data_ptr(), wp[1].data_ptr(), bs[1].data_ptr(), x.data_ptr(), two.data_ptr(), 1, N, H, W,
                                              C, C, 1, Kn._stream()), "conv2")
        xf = x.float().permute(0, 3, 1, 2)
        wf = [w.bfloat16().float().permute(0, 3, 1, 2) for w in ws]
        m_ref = F.relu(F.conv2d(xf, wf[0], bs[0], padding=1)).bfloat16().float()
        refs.append(F.relu(F.conv2d(m_ref, wf[1], bs[1], padding=1) + xf).permute(0, 2, 3, 1))
        keep.append((x, ws, bs, wp, mid))
        outs.append(y)
        want.append(two)
    _hip.check(_hip.lib().rf_conv3x3_pair_group_bf16(arr, len(shapes), Kn._stream()), "rf_conv3x3_pair_group_bf16")
    torch.cuda.synchronize()
    for y, two, ref in zip(outs, want, refs):
        assert torch.isfinite(y.float()).all()
        assert torch.equal(y, two), float((y.float() - two.float()).abs().max())
        assert rel_err(y.float(), ref) < 1e-2  # (bf16 output rounding + the occasional flipped rounding of the intermediate)


@pytest.mark.parametrize("M,cin,cout,res", [(263424, 64, 256, True), (1000, 64, 256, False), (777, 256, 64, False),
                                            (16, 64, 64, True), (5, 64, 64, False), (4097, 256, 64, True)])
def test_pointwise_bf16(M, cin, cout, res):
    """Streaming 1x1 convolution over bf16 maps (weights in registers) against F.linear on the same bf16-rounded
    operands, including ragged tails (M not a multiple of the 16-pixel tile / of the 64-pixel workgroup group)."""
    from routeformer_amd import _hip, kernels as Kn
    from routeformer_amd.models.video_backbone.hrnet16 import pack_pointwise_weights
    g = _g(M + cin + cout)
    x = torch.randn(M, cin, generator=g).bfloat16()
    w = (torch.randn(cout, cin, generator=g) / math.sqrt(cin))
    b = torch.randn(cout, generator=g)
    r = torch.randn(M, cout, generator=g).bfloat16() if res else None
    ref = F.linear(x.float(), w.bfloat16().float(), b)
    if res:
        ref = ref + r.float()
    ref = F.relu(ref)
    assert _hip.lib().rf_pointwise_bf16_supported(cin, cout) == 1 and _hip.lib().rf_pointwise_bf16_supported(128, 128) == 0
    xd, bd, rd = x.to(DEV), b.to(DEV), (r.to(DEV) if res else None)
    wp = pack_pointwise_weights(w.to(DEV))
    y = torch.full((M + 3, cout), 7.0, device=DEV, dtype=torch.bfloat16)  # 3 guard rows: nothing past M is written
    _hip.check(_hip.lib().rf_pointwise_bf16(xd.data_ptr(), wp.data_ptr(), bd.data_ptr(), _hip.ptr(rd), y.data_ptr(), M,
                                            cin, cout, 1, Kn._stream()), "pointwise")
    assert torch.all(y[M:] == 7.0)
    assert torch.allclose(y[:M].float().cpu(), ref, rtol=2.0 ** -7, atol=2e-2)  # one bf16 rounding of the result
    assert rel_err(y[:M].float(), ref) < 6e-3


@pytest.mark.parametrize("B,T,P,E,rot", [(8, 40, 30, 64, True), (3, 7, 5, 0, True), (2, 40, 30, 16, False)])
def test_motion_input_and_rotate_head(B, T, P, E, rot):
    """rf_motion_input / rf_rotate_head against the elementwise formulation of routeformer.py:210-233 (angle, norm,
    acceleration, rotation by -origin, concat; output un-rotation), values and gradients."""
    from routeformer_amd import kernels as Kn
    from routeformer_amd.utils.tensor import estimate_angle_and_norm, rotate
    g = _g(B * T + E)
    motion = torch.randn(B, T, 2, generator=g).to(DEV)
    vis = torch.randn(B, T, E, generator=g).to(DEV).requires_grad_() if E else None
    angle, norm = estimate_angle_and_norm(motion)
    origin = angle[:, -1:, :] if rot else angle[:, :1, :]
    feats = [torch.cat([rotate(motion, -origin) if rot else motion, (angle - origin) / torch.pi, norm,
                        F.pad(norm[:, 1:, :] - norm[:, :-1, :], (0, 0, 1, 0))], dim=-1)]
    if E:
        feats.append(vis)
    ref = torch.cat(feats, dim=-1)
    vis2 = vis.detach().clone().requires_grad_() if E else None
    x, org = Kn.motion_input(motion, vis2, rot)
    assert torch.allclose(x, ref, rtol=1e-5, atol=1e-6) and torch.allclose(org, origin.reshape(B), rtol=1e-6, atol=1e-6)
    if E:
        w = torch.randn(ref.shape, generator=g).to(DEV)
        (ref * w).sum().backward()
        (x * w).sum().backward()
        assert torch.equal(vis2.grad, vis.grad)
        assert torch.equal(Kn.motion_input(motion, vis2, rot, True)[0][..., 5:], torch.zeros_like(vis2))  # _only_motion
    out = torch.randn(B, P, 2 + E, generator=g).to(DEV).requires_grad_()
    out2 = out.detach().clone().requires_grad_()
    ref_y = torch.cat([rotate(out[:, :, :2], origin), out[:, :, 2:]], dim=-1)
    y = Kn.rotate_head(out2, org)
    assert torch.allclose(y, ref_y, rtol=1e-5, atol=1e-6)
    w = torch.randn(ref_y.shape, generator=g).to(DEV)
    (ref_y * w).sum().backward()
    (y * w).sum().backward()
    assert torch.allclose(out2.grad, out.grad, rtol=1e-5, atol=1e-6)


def test_vision_helpers():
    from routeformer_amd import _hip, kernels as Kn
    from routeformer_amd.models.video_backbone.hrnet16 import HRNet16Backbone
    g = _g(21)
    # upsample (+addend, accumulate, relu) for the resolution pairs HRNet uses
    for (hi, ho) in ((14, 28), (4, 28), (7, 14), (1, 8), (2, 3), (28, 28)):
        x = torch.randn(2, 16, hi, hi, generator=g)
        add = torch.randn(2, 16, ho, ho, generator=g)
        ref = F.relu(add + F.interpolate(x, size=(ho, ho), mode="bilinear", align_corners=False))
        xd, ad = _nhwc(x).to(DEV), _nhwc(add).to(DEV)
        y = HRNet16Backbone._upsample(xd, (ho, ho), addend=ad, relu=True)
        assert rel_err(y, _nhwc(ref)) < 1e-5, (hi, ho)
        y2 = ad.clone()
        HRNet16Backbone._upsample(xd, (ho, ho), out=y2, ldy=16, accumulate=True)
        assert rel_err(y2, _nhwc(add + F.interpolate(x, size=(ho, ho), mode="bilinear", align_corners=False))) < 1e-5
        # bf16 maps: fp32 arithmetic on the bf16 values, one rounding at the store
        xh, ah = xd.bfloat16(), ad.bfloat16()
        yh = HRNet16Backbone._upsample(xh, (ho, ho), addend=ah, relu=True)
        assert yh.dtype == torch.bfloat16
        # (a last-bit difference in the fp32 sum can flip a round-to-even tie: compare to one bf16 ulp)
        assert torch.allclose(yh.float(), HRNet16Backbone._upsample(xh.float(), (ho, ho), addend=ah.float(), relu=True),
                              rtol=2.0 ** -7, atol=1e-6)
        assert torch.equal(HRNet16Backbone._add(xh, xh, True), F.relu(xh.float() + xh.float()).bfloat16())
    a5, b5 = torch.randn(1031, generator=g).to(DEV), torch.randn(1031, generator=g).to(DEV)  # odd length: scalar tail
    assert torch.equal(HRNet16Backbone._add(a5, b5, True), F.relu(a5 + b5))
    # adaptive avg pool -> tokens with the -1 row, for 28x28 / 8x8 / 12x12 / 56x56 / 3x5 maps
    for (h, w) in ((28, 28), (8, 8), (12, 12), (56, 56), (3, 5)):
        x = torch.randn(2, 24, h, w, generator=g)
        tok, xd = torch.empty(2, 65, 24, device=DEV), _nhwc(x).to(DEV)
        _hip.check(_hip.lib().rf_avgpool8_tokens(xd.data_ptr(), 0, tok.data_ptr(), 2, h, w, 24, Kn._stream()), "pool")
        ref = F.adaptive_avg_pool2d(x, (8, 8)).permute(0, 2, 3, 1).reshape(2, 64, 24)
        assert rel_err(tok[:, :64], ref) < 1e-5 and torch.all(tok[:, 64] == -1)
        xh, tok2 = xd.bfloat16(), torch.empty_like(tok)
        _hip.check(_hip.lib().rf_avgpool8_tokens(xh.data_ptr(), 1, tok2.data_ptr(), 2, h, w, 24, Kn._stream()), "pool bf16")
        ref2 = F.adaptive_avg_pool2d(xh.float().cpu().permute(0, 3, 1, 2), (8, 8)).permute(0, 2, 3, 1).reshape(2, 64, 24)
        assert rel_err(tok2[:, :64], ref2) < 1e-5 and torch.all(tok2[:, 64] == -1)
    # stem: frame gather + fp16 cast + conv0
    video = torch.rand(2, 5, 3, 16, 20, generator=g).half()
    w0 = torch.randn(3, 3, 2, 2, generator=g)
    idx = torch.tensor([4, 1, 3])
    ref = F.conv2d(video[:, idx].flatten(0, 1).float(), w0, stride=2)
    y, idx_d, w0_d = torch.empty(6, 8, 10, 4, device=DEV), idx.int().to(DEV), w0.to(DEV)
    for vid, is32 in ((video.to(DEV), 0), (video.float().to(DEV), 1)):
        _hip.check(_hip.lib().rf_stem_conv0(vid.data_ptr(), is32, idx_d.data_ptr(), w0_d.data_ptr(), y.data_ptr(), 0,
                                            2, 5, 3, 16, 20, Kn._stream()), "stem")
        assert rel_err(y[..., :3], _nhwc(ref)) < 1e-5 and torch.all(y[..., 3] == 0)
        yh = torch.empty(6, 8, 10, 4, device=DEV, dtype=torch.bfloat16)
        _hip.check(_hip.lib().rf_stem_conv0(vid.data_ptr(), is32, idx_d.data_ptr(), w0_d.data_ptr(), yh.data_ptr(), 1,
                                            2, 5, 3, 16, 20, Kn._stream()), "stem bf16 maps")
        assert torch.equal(yh, y.bfloat16())
    # raw uint8 camera bytes: the dataset's `astype(np.float16) / 255.0` (io/dataset.py:1506-1523) fused into the
    # stem must give EXACTLY what the fp16 path gives on the numpy-converted clip
    raw = torch.randint(0, 256, (2, 5, 3, 16, 20), generator=g, dtype=torch.uint8)
    as_f16 = torch.from_numpy(raw.numpy().astype(np.float16) / 255.0)
    assert as_f16.dtype == torch.float16
    y8, y16 = torch.empty_like(y), torch.empty_like(y)
    raw_d, f16_d = raw.to(DEV), as_f16.to(DEV)
    _hip.check(_hip.lib().rf_stem_conv0(raw_d.data_ptr(), 2, idx_d.data_ptr(), w0_d.data_ptr(), y8.data_ptr(), 0, 2, 5, 3, 16,
                                        20, Kn._stream()), "stem u8")
    _hip.check(_hip.lib().rf_stem_conv0(f16_d.data_ptr(), 0, idx_d.data_ptr(), w0_d.data_ptr(), y16.data_ptr(), 0, 2, 5, 3, 16,
                                        20, Kn._stream()), "stem f16")
    assert torch.equal(y8, y16)


def test_small_ops_match_torch():
    """csrc/smallops.hip (one launch each) against the torch expressions they replace: windowed lower median (incl. ties,
    NaN windows, a window count that does not divide T), motion differencing with / without normalisation, the time-
    feature table and its gradient, the frame timeline scatter / gather, the smart-decoder tail and its gradient."""
    from routeformer_amd import kernels as Kn
    from routeformer_amd.utils.tensor import median_downsampler
    g = _g(23)
    # median: (B,T,C) -> (B,target,C); reference = torch.median per window (lower median)
    for (B, T, C, target) in ((3, 1600, 2, 40), (2, 1203, 3, 30), (1, 64, 1, 5), (2, 50, 2, 49)):
        x = torch.randn(B, T, C, generator=g)
        x[0, : T // 2, 0] = x[0, : T // 2, 0].round()            # ties
        if T > 100:
            x[B - 1, 7, C - 1] = float("nan")                    # a NaN window
        w = T // target
        ref = torch.stack([x[:, i * w:(i + 1) * w].median(dim=1).values for i in range(target)], dim=1)
        got = median_downsampler(x.to(DEV), target).cpu()
        assert got.shape == ref.shape and torch.equal(torch.nan_to_num(got, nan=123.0), torch.nan_to_num(ref, nan=123.0))
    # motion
    gps = torch.randn(4, 40, 2, generator=g).cumsum(dim=1)
    for norm in (False, True):
        mv = gps[:, 1:] - gps[:, :-1]
        if norm:
            mv = (mv - 1.83) / 0.91
        ref = F.pad(mv, (0, 0, 1, 0))
        assert rel_err(Kn.motion_diff(gps.to(DEV), norm, 1.83, 0.91), ref) < 1e-6
    # time table + gradient
    d, L = 832, 70
    w = torch.randn(d, 1, generator=g, requires_grad=True)
    pe = torch.randn(1, 5000, d, generator=g)
    ref = torch.arange(L, dtype=torch.float32).view(L, 1) * w.view(1, d) + pe[0, :L]
    wd = w.detach().to(DEV).requires_grad_()
    got = Kn.time_table(wd, pe.to(DEV), L)
    assert rel_err(got, ref) < 1e-6
    up = torch.randn(L, d, generator=g)
    ref.backward(up)
    got.backward(up.to(DEV))
    assert rel_err(wd.grad, w.grad) < 1e-5
    # timeline scatter / gather
    N, T, E = 6, 40, 64
    idx = torch.flip(torch.arange(T - 1, 0, -5), dims=[0])
    feats = torch.randn(N, idx.numel(), E, generator=g, requires_grad=True)
    ref = torch.zeros(N, T, E)
    ref[:, idx] = feats
    fd = feats.detach().to(DEV).requires_grad_()
    got = Kn.timeline(fd, idx.to(DEV), T)
    assert torch.equal(got.cpu(), ref.detach())
    up = torch.randn(N, T, E, generator=g)
    ref.backward(up)
    got.backward(up.to(DEV))
    assert torch.equal(fd.grad.cpu(), feats.grad)
    # smart tail (both variants) + the gradient with the alias branch folded in
    B, L, P, C = 3, 40, 30, 69
    for smart in (True, False):
        x = torch.randn(B, L, C, generator=g, requires_grad=True)
        tail = x[:, -1:, :].expand(B, P, C) if smart else torch.zeros(B, P, C)
        ref = torch.cat([x, tail], dim=1)
        xd = x.detach().to(DEV).requires_grad_()
        y, alias = Kn.smart_tail(xd, P, smart)
        assert torch.equal(y.cpu(), ref.detach()) and torch.equal(alias, xd)
        u1, u2 = torch.randn(B, L + P, C, generator=g), torch.randn(B, L, C, generator=g)
        (ref * u1).sum().backward()
        x.grad += u2
        ((y * u1.to(DEV)).sum() + (alias * u2.to(DEV)).sum()).backward()
        assert rel_err(xd.grad, x.grad) < 1e-5


@pytest.mark.parametrize("adt", [torch.float32, torch.bfloat16])
def test_fuse_upsample_sum_and_concat_pool(adt):
    """csrc/fuse.hip against torch on the CPU: (i) several output branches of a fuse layer in one launch --
    relu(base + base2 + sum_j bilinear(src_j)), incl. an entry without base2, one with a single source and an in-place
    one; (ii) up-sample + concat + AdaptiveAvgPool2d((8,8)) + token layout with the -1 row, for the trunk's 28/14/7/4
    pyramid (224 x 224 input) and a 56/28/14/7 one (448 x 448), never materialising the concatenated map."""
    import ctypes
    from routeformer_amd import _hip, kernels as Kn
    g = _g(17)
    code = 1 if adt == torch.bfloat16 else 0

    def rnd(*shape):  # NCHW values representable in the storage type
        return torch.randn(*shape, generator=g).to(adt).float()

    def dev(x):  # NCHW cpu -> NHWC device in the storage type
        return x.permute(0, 2, 3, 1).contiguous().to(DEV).to(adt)

    N = 3
    cases = [  # (C, Ho, Wo, has_base, has_base2, [(Hi, Wi)], relu)
        (16, 28, 28, True, False, [(14, 14), (7, 7), (4, 4)], 1),
        (32, 14, 14, True, True, [(7, 7), (4, 4)], 1),
        (64, 7, 7, True, True, [(4, 4)], 0),
        (8, 12, 20, False, True, [(5, 7)], 1),
    ]
    arr = (_hip.FuseEntry * len(cases))()
    keep, refs = [], []
    for e, (C, Ho, Wo, hb, hb2, srcs, relu) in zip(arr, cases):
        base = rnd(N, C, Ho, Wo) if hb else None
        base2 = rnd(N, C, Ho, Wo) if hb2 else None
        ss = [rnd(N, C, h, w) for h, w in srcs]
        ref = torch.zeros(N, C, Ho, Wo)
        for t in (base, base2):
            if t is not None:
                ref = ref + t
        for t in ss:
            ref = ref + F.interpolate(t, size=(Ho, Wo), mode="bilinear", align_corners=False)
        refs.append(F.relu(ref) if relu else ref)
        bd, b2d, sd = (dev(base) if hb else None), (dev(base2) if hb2 else None), [dev(t) for t in ss]
        out = bd if (hb and C == 64) else torch.empty(N, Ho, Wo, C, device=DEV, dtype=adt)  # one in-place entry
        e.base, e.base2, e.out = _hip.ptr(bd), _hip.ptr(b2d), out.data_ptr()
        e.N, e.Ho, e.Wo, e.C, e.n_src, e.relu = N, Ho, Wo, C, len(ss), relu
        for i, (t, (h, w)) in enumerate(zip(sd, srcs)):
            e.src[i], e.Hi[i], e.Wi[i] = t.data_ptr(), h, w
        keep.append((bd, b2d, sd, out))
    _hip.check(_hip.lib().rf_fuse_upsample_sum(arr, len(cases), code, Kn._stream()), "rf_fuse_upsample_sum")
    torch.cuda.synchronize()
    for (_, _, _, out), ref in zip(keep, refs):
        got = out.float().cpu().permute(0, 3, 1, 2)
        if code == 0:
            assert rel_err(got, ref) < 1e-5
        else:  # one bf16 rounding of the fp32 sum
            assert torch.allclose(got, ref, rtol=2.0 ** -7, atol=1e-6)
    for sizes in (((28, 28), (14, 14), (7, 7), (4, 4)), ((56, 56), (28, 28), (14, 14), (7, 7))):
        chans = (16, 32, 64, 128)
        xs = [rnd(2, c, h, w) for c, (h, w) in zip(chans, sizes)]
        Hf, Wf = sizes[0]
        cat = torch.cat([t if t.shape[2:] == (Hf, Wf) else F.interpolate(t, size=(Hf, Wf), mode="bilinear", align_corners=False)
                         for t in xs], dim=1)
        ref = F.adaptive_avg_pool2d(cat, (8, 8)).permute(0, 2, 3, 1).reshape(2, 64, 240)
        xd = [dev(t) for t in xs]
        maps = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in xd])
        dims = [(ctypes.c_int32 * 4)(*[t.shape[k] for t in xd]) for k in (1, 2, 3)]
        tok = torch.empty(2, 65, 240, device=DEV)
        _hip.check(_hip.lib().rf_concat_pool_tokens(maps, dims[0], dims[1], dims[2], 4, code, tok.data_ptr(), 2, Kn._stream()),
                   "rf_concat_pool_tokens")
        assert rel_err(tok[:, :64], ref) < 1e-5 and torch.all(tok[:, 64] == -1)


@pytest.mark.parametrize("prec,tol", [("f32", 1e-4), ("bf16", 5e-2)])
def test_hrnet16_golden(prec, tol):
    """Whole frozen conv encoder against the reference's outputs (tests/golden/hrnet.npz)."""
    from routeformer_amd import kernels as Kn, synthetic
    from routeformer_amd.models.video_backbone import HRNet16Backbone
    Kn.set_precision(prec)
    G = golden("hrnet")
    net = HRNet16Backbone()
    net.load_state_dict(synthetic.synth_state_dict(net.state_dict(), 7))
    net = net.to(DEV)
    for tag, n, hw in (("s64", 2, 64), ("s96", 1, 96), ("s224", 2, 224)):
        x = synthetic.synth_video(1, n, hw, hw, 11, "hrnet." + tag)[0].to(DEV)
        y = net(x)
        assert y.shape == (n, 240, 8, 8)
        assert rel_err(y, G[tag + ".y"]) < tol, tag
        if prec == "bf16":  # the fused BasicBlock pairs (RF_CONV_PAIR, off by default: no faster) give the same bits
            from routeformer_amd.models.video_backbone import hrnet16 as HR
            was = HR.CONV_PAIR
            HR.CONV_PAIR = True
            try:
                assert torch.equal(net(x), y), tag
            finally:
                HR.CONV_PAIR = was


@pytest.mark.parametrize("cin,cout,N,H,W,res", [(4, 64, 3, 28, 28, False), (4, 64, 2, 112, 112, False), (4, 64, 1, 30, 44, False),
                                                (64, 64, 3, 28, 28, False), (64, 64, 2, 56, 56, False), (64, 64, 1, 18, 112, False),
                                                (64, 64, 5, 6, 4, True), (16, 16, 3, 28, 28, True), (16, 32, 2, 28, 28, False),
                                                (16, 64, 2, 14, 28, True), (16, 128, 1, 56, 56, True), (32, 32, 3, 14, 14, True),
                                                (32, 64, 3, 14, 14, False), (32, 128, 2, 28, 28, True), (64, 128, 2, 14, 14, True),
                                                (16, 16, 2, 112, 112, False), (256, 32, 2, 28, 28, False), (256, 32, 1, 56, 56, False),
                                                (64, 64, 1, 112, 112, False), (64, 128, 3, 8, 8, False)])
def test_conv3x3_stride2_kernel(cin, cout, N, H, W, res):
    """rf_conv3x3s2_bf16 (round 4: 3x3 / stride 2 / pad 1 on bf16 NHWC maps -- the trunk's two stem convolutions, hrnetv2.py:
    292-293,434-440, and the stride-2 chains of its fuse layers / transitions, :148-200) against torch's conv2d on the same
    bf16-rounded operands: tiles that straddle rows and images, borders, both tile sizes, with and without the residual."""
    from routeformer_amd import _hip
    from routeformer_amd.models.video_backbone.hrnet16 import pack_conv3x3s2_weights
    g = _g(41)
    assert _hip.lib().rf_conv3x3s2_bf16_supported(cin, cout, W) == 1
    x = torch.randn(N, H, W, cin, generator=g).bfloat16()
    if cin == 4:
        x[..., 3] = 0  # (the stem's zero fourth channel)
    w = (torch.randn(cout, 3, 3, cin, generator=g) / math.sqrt(9 * cin)).bfloat16().float()
    b = torch.randn(cout, generator=g) * 0.2
    r = torch.randn(N, H // 2, W // 2, cout, generator=g).bfloat16() if res else None
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), b, stride=2, padding=1).permute(0, 2, 3, 1)
    ref = F.relu(ref + r.float()) if res else F.relu(ref)
    xd, bd = x.to(DEV), b.to(DEV)
    rd = r.to(DEV) if res else None
    wp = pack_conv3x3s2_weights(w.to(DEV))
    y = torch.full((N, H // 2, W // 2, cout), float("nan"), device=DEV, dtype=torch.bfloat16)
    _hip.check(_hip.lib().rf_conv3x3s2_bf16(xd.data_ptr(), wp.data_ptr(), bd.data_ptr(), rd.data_ptr() if res else None, y.data_ptr(),
                                            N, H, W, cin, cout, 1, torch.cuda.current_stream().cuda_stream), "rf_conv3x3s2_bf16")
    torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all()
    assert rel_err(y.float(), ref) < 8e-3, rel_err(y.float(), ref)   # bf16 output rounding: 2^-9 of the largest value


def test_adamw_clip():
    from routeformer_amd import _hip, kernels as Kn
    g = _g(5)
    n = 100003
    p0, g0 = torch.randn(n, generator=g), torch.randn(n, generator=g) * 3
    pc = p0.clone().requires_grad_()
    opt = torch.optim.AdamW([pc], lr=1e-3, weight_decay=1e-2, betas=(0.9, 0.999), eps=1e-8)
    pd, m, v = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in (1, 2, 3):
        gs = g0 * step
        pc.grad = gs.clone()
        torch.nn.utils.clip_grad_norm_([pc], 2.5)
        opt.step()
        gd = gs.to(DEV)
        parts = _hip.lib().rf_sumsq_parts(n)
        ss = torch.full((parts,), float("nan"), device=DEV)  # per-workgroup partials, every slot overwritten
        ss2 = torch.empty_like(ss)
        _hip.check(_hip.lib().rf_sumsq(gd.data_ptr(), n, ss.data_ptr(), Kn._stream()), "sumsq")
        _hip.check(_hip.lib().rf_sumsq(gd.data_ptr(), n, ss2.data_ptr(), Kn._stream()), "sumsq")
        assert torch.equal(ss, ss2)  # no atomics: bit-reproducible (data-parallel replicas must not drift)
        assert abs(float(ss.sum()) - float(gs.double().square().sum())) < 1e-3 * float(gs.double().square().sum())
        _hip.check(_hip.lib().rf_adamw_clip(pd.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, ss.data_ptr(),
                                            parts, 2.5, 1e-3, 0.9, 0.999, 1e-8, 1e-2, step, 1.0, Kn._stream()), "adamw")
        assert rel_err(pd, pc) < 1e-5, step


def test_fused_adamw_skipped_ranges_keep_their_own_step_count():
    """A parameter range that took no part in a step (dropped gaze branch: ``.grad`` None in the reference) is left
    untouched by torch.optim.AdamW INCLUDING its per-parameter ``state['step']``: when it is updated again its bias
    corrections use its own update count.  ``FusedAdamW.step(skip=...)`` against torch.optim.AdamW on three parameters,
    the middle one skipped in steps 2, 3 and 5 (the fused update used the global step for every slot)."""
    from routeformer_amd.engine import FusedAdamW
    g = _g(9)
    sizes = [640, 1280, 320]
    offs = [0, 640, 1920]
    n = sum(sizes)
    p0 = torch.randn(n, generator=g)
    ps = [p0[o:o + k].clone().requires_grad_() for o, k in zip(offs, sizes)]
    ref = torch.optim.AdamW(ps, lr=1e-2, weight_decay=1e-2, betas=(0.9, 0.999), eps=1e-8)
    flat_p, flat_g = p0.clone().to(DEV), torch.zeros(n, device=DEV)
    opt = FusedAdamW(flat_p, flat_g, lr=1e-2, weight_decay=1e-2, max_grad_norm=2.5)
    for step in range(1, 8):
        skipped = step in (2, 3, 5)
        grads = [torch.randn(k, generator=g) * 0.7 for k in sizes]
        flat_g.zero_()
        for i, (o, k) in enumerate(zip(offs, sizes)):
            if i == 1 and skipped:
                ps[i].grad = None
            else:
                ps[i].grad = grads[i].clone()
                flat_g[o:o + k] = grads[i].to(DEV)
        torch.nn.utils.clip_grad_norm_([q for q in ps if q.grad is not None], 2.5)
        ref.step()
        opt.step(1.0, skip=((offs[1], offs[1] + sizes[1]),) if skipped else ())
        got = flat_p.cpu()
        for i, (o, k) in enumerate(zip(offs, sizes)):
            assert rel_err(got[o:o + k], ps[i]) < 2e-6, (step, i)


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("prec,tol", [("f32", 2e-5), ("bf16", 1e-2)])
def test_wgrad_grouped(prec, tol):
    """Several dW[N,K] += dY^T X (+ db += colsum dY) problems of different shapes in one launch; slots start
    non-zero (accumulate semantics)."""
    import ctypes
    from routeformer_amd import _hip, kernels as Kn
    from routeformer_amd._hip import ptr
    g = _g(31)
    shapes = [(12480, 384, 128, True), (325, 128, 256, True), (560, 832, 832, False), (40, 2496, 832, True),
              (130, 68, 132, True), (7, 128, 128, False)]
    arr = (_hip.WgradEntry * len(shapes))()
    keep, want = [], []
    for e, (M, N, K, with_bias) in zip(arr, shapes):
        dy, x = torch.randn(M, N, generator=g), torch.randn(M, K, generator=g)
        dw0, db0 = torch.randn(N, K, generator=g), torch.randn(N, generator=g)
        d = [t.to(DEV) for t in (dy, x, dw0, db0)]
        keep.append(d)
        e.dy, e.x, e.dw, e.db = ptr(d[0]), ptr(d[1]), ptr(d[2]), (ptr(d[3]) if with_bias else None)
        e.M, e.N, e.K, e.ld_dy, e.ld_x = M, N, K, N, K
        e.splits = Kn._splits(-(-N // 64) * -(-K // 64), M)
        # exclusive: zeroed slot, single writer -> plain stores when the problem has one K slice (560 x 832 x 832)
        excl = (M, N, K) in ((560, 832, 832), (7, 128, 128))
        e.exclusive = 1 if excl else 0
        if excl:
            d[2].zero_(); dw0 = torch.zeros_like(dw0)
        want.append((dw0 + dy.T @ x, db0 + dy.sum(0) if with_bias else db0))
    rc = _hip.lib().rf_wgrad_grouped(arr, len(shapes), 1 if prec == "bf16" else 0, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, _hip.lib().rf_last_error()
    for d, (dw, db), sh in zip(keep, want, shapes):
        assert rel_err(d[2], dw) < tol, sh
        assert rel_err(d[3], db) < 1e-4, sh


@pytest.mark.parametrize("M,F_,NP,act", [(320, 256, 192, "gelu"), (320, 0, 64, "gelu"), (77, 256, 0, "relu"), (1300, 128, 64, "gelu")])
def test_rowchain_kernels_vs_torch(M, F_, NP, act):
    """rf_rowchain_fwd / _bwd through the C ABI against torch autograd (fp32, on bf16-rounded matmul operands -- the kernel's
    arithmetic contract): a -> Wo + residual -> LayerNorm [-> conv1 -> act -> conv2 + residual -> LayerNorm] [-> projection];
    outputs, every save, and in the backward da, the residual gradient, the weight-gradient operands and dgamma / dbeta."""
    import ctypes
    import torch.nn.functional as Fn
    from routeformer_amd import _hip
    from routeformer_amd._hip import ptr
    D = 64
    g = _g(41)
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
    a, x = rnd(M, D), rnd(M, D)
    wo, bo, g1, be1 = rnd(D, D, sc=D ** -0.5), rnd(D, sc=0.1), 1 + rnd(D, sc=0.1), rnd(D, sc=0.1)
    ffn = F_ > 0
    w1, b1, w2, b2 = rnd(max(F_, 1), D, sc=D ** -0.5), rnd(max(F_, 1), sc=0.1), rnd(D, max(F_, 1), sc=max(F_, 1) ** -0.5), rnd(D, sc=0.1)
    g2, be2 = 1 + rnd(D, sc=0.1), rnd(D, sc=0.1)
    wp, bp = rnd(max(NP, 1), D, sc=D ** -0.5), rnd(max(NP, 1), sc=0.1)
    dout, dproj = rnd(M, D), rnd(M, max(NP, 1))
    # ---- torch reference (the matmul operands rounded to bf16 as the MFMAs see them) ----
    at, xt = a.clone().requires_grad_(), x.clone().requires_grad_()
    P = {k: v.clone().requires_grad_() for k, v in dict(wo=wo, bo=bo, g1=g1, be1=be1, w1=w1, b1=b1, w2=w2, b2=b2, g2=g2, be2=be2,
                                                        wp=wp, bp=bp).items()}
    bf = lambda t_: t_ + (_bf(t_.detach()) - t_.detach())  # straight-through bf16 rounding
    pre1 = bf(at) @ bf(P["wo"]).T + P["bo"] + xt
    x1 = Fn.layer_norm(pre1, (D,), P["g1"], P["be1"], 1e-5)
    out = x1
    if ffn:
        z = bf(x1) @ bf(P["w1"]).T + P["b1"]
        h = Fn.gelu(z) if act == "gelu" else torch.relu(z)
        pre2 = bf(h) @ bf(P["w2"]).T + P["b2"] + x1
        out = Fn.layer_norm(pre2, (D,), P["g2"], P["be2"], 1e-5)
    loss = (out * dout).sum()
    if NP:
        proj = bf(out) @ bf(P["wp"]).T + P["bp"]
        loss = loss + (proj * dproj).sum()
    loss.backward()
    # ---- kernels ----
    dev = lambda t_: t_.detach().to(DEV).contiguous()
    T = {k: dev(v) for k, v in dict(a=a, x=x, wo=wo, bo=bo, g1=g1, be1=be1, w1=w1, b1=b1, w2=w2, b2=b2, g2=g2, be2=be2, wp=wp, bp=bp,
                                    dout=dout, dproj=dproj).items()}
    O = {k: torch.empty(M, w, device=DEV) for k, w in dict(x1=D, y=D, proj=max(NP, 1), xhat1=D, xhat2=D, z=max(F_, 1), h=max(F_, 1)).items()}
    O["rstd1"], O["rstd2"] = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    c = _hip.RowChain()
    for k in ("a", "x", "wo", "bo", "g1", "be1"):
        setattr(c, k, ptr(T[k]))
    if ffn:
        for k in ("w1", "b1", "w2", "b2", "g2", "be2"):
            setattr(c, k, ptr(T[k]))
        c.y, c.xhat2, c.rstd2, c.h = ptr(O["y"]), ptr(O["xhat2"]), ptr(O["rstd2"]), ptr(O["h"])
        if act == "gelu":
            c.z = ptr(O["z"])
    if NP:
        c.wp, c.bp, c.proj = ptr(T["wp"]), ptr(T["bp"]), ptr(O["proj"])
    c.x1, c.xhat1, c.rstd1 = ptr(O["x1"]), ptr(O["xhat1"]), ptr(O["rstd1"])
    c.d_model, c.d_ff, c.n_proj, c.act, c.eps = D, F_, NP, {"gelu": 2, "relu": 1}[act], 1e-5
    st = torch.cuda.current_stream().cuda_stream
    assert _hip.lib().rf_rowchain_supported(D, F_, NP)
    assert _hip.lib().rf_rowchain_fwd(ctypes.byref(c), M, 0.0, None, st) == 0, _hip.lib().rf_last_error()
    torch.cuda.synchronize()
    assert rel_err(O["x1"].cpu(), x1.detach()) < 5e-3
    if ffn:
        assert rel_err(O["y"].cpu(), out.detach()) < 5e-3 and rel_err(O["h"].cpu(), h.detach()) < 5e-3
        if act == "gelu":
            assert rel_err(O["z"].cpu(), z.detach()) < 5e-3
    if NP:
        assert rel_err(O["proj"].cpu(), proj.detach()) < 5e-3
    G = {k: torch.zeros(D, device=DEV) for k in ("dg1", "db1", "dg2", "db2")}
    R = {k: torch.empty(M, w, device=DEV) for k, w in dict(dpre1=D, da=D, dpre2=D, dz=max(F_, 1)).items()}
    b = _hip.RowChainBwd()
    b.dyin = ptr(T["dout"])
    if NP:
        b.dproj, b.wp = ptr(T["dproj"]), ptr(T["wp"])
    if ffn:
        b.w1, b.w2, b.g2, b.xhat2, b.rstd2 = ptr(T["w1"]), ptr(T["w2"]), ptr(T["g2"]), ptr(O["xhat2"]), ptr(O["rstd2"])
        b.zsrc = ptr(O["z"] if act == "gelu" else O["h"])
        b.dpre2, b.dz, b.dg2, b.db2 = ptr(R["dpre2"]), ptr(R["dz"]), ptr(G["dg2"]), ptr(G["db2"])
    b.wo, b.g1, b.xhat1, b.rstd1 = ptr(T["wo"]), ptr(T["g1"]), ptr(O["xhat1"]), ptr(O["rstd1"])
    b.dpre1, b.da, b.dg1, b.db1 = ptr(R["dpre1"]), ptr(R["da"]), ptr(G["dg1"]), ptr(G["db1"])
    b.d_model, b.d_ff, b.n_proj, b.act = D, F_, NP, c.act
    assert _hip.lib().rf_rowchain_bwd(ctypes.byref(b), M, 0.0, None, st) == 0, _hip.lib().rf_last_error()
    torch.cuda.synchronize()
    tol = 2e-2  # bf16 gradient images
    assert rel_err(R["da"].cpu(), at.grad) < tol and rel_err(R["dpre1"].cpu(), xt.grad) < tol
    assert rel_err(G["dg1"].cpu(), P["g1"].grad) < tol and rel_err(G["db1"].cpu(), P["be1"].grad) < tol
    # the weight-gradient operands: dW = dy^T x for the pairs the Python glue queues
    assert rel_err(R["dpre1"].cpu().T @ a, P["wo"].grad) < tol
    if ffn:
        assert rel_err(G["dg2"].cpu(), P["g2"].grad) < tol and rel_err(G["db2"].cpu(), P["be2"].grad) < tol
        assert rel_err(R["dpre2"].cpu().T @ O["h"].cpu(), P["w2"].grad) < tol
        assert rel_err(R["dz"].cpu().T @ O["x1"].cpu(), P["w1"].grad) < tol
        assert rel_err(R["dz"].cpu().sum(0), P["b1"].grad) < tol


@pytest.mark.parametrize("ybf,xbf", [(True, True)])
def test_wgrad_grouped_bf16_operands(ybf, xbf):
    """rf_wgrad_tr with operands that already lie in memory as bf16 (the fused encoder stacks' dy slabs / activation saves,
    kernels.BF16_SAVES): BIT-identical to the fp32-operand launch on the same (bf16-representable) values -- the kernel rounds
    fp32 operands to bf16 on the way into LDS, so nothing but the bytes read changes."""
    from routeformer_amd import _hip, kernels as Kn
    from routeformer_amd._hip import ptr
    g = _g(33)
    shapes = [(12480, 384, 128, True), (4160, 128, 512, True), (325, 512, 128, True), (1030, 128, 128, False), (40, 384, 128, True)]
    results = []
    for as_bf in (False, True):
        arr = (_hip.WgradEntry * len(shapes))()
        keep = []
        gg = _g(34)
        for e, (M, N, K, with_bias) in zip(arr, shapes):
            dy, x = _bf(torch.randn(M, N, generator=gg)), _bf(torch.randn(M, K, generator=gg))
            dyd = dy.to(DEV).to(torch.bfloat16 if (as_bf and ybf) else torch.float32)
            xd = x.to(DEV).to(torch.bfloat16 if (as_bf and xbf) else torch.float32)
            dw, db = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
            keep.append((dyd, xd, dw, db, dy, x))
            e.dy, e.x, e.dw, e.db = ptr(dyd), ptr(xd), ptr(dw), (ptr(db) if with_bias else None)
            e.M, e.N, e.K, e.ld_dy, e.ld_x = M, N, K, N, K
            e.splits, e.exclusive = 1, 1   # one chunk, plain stores: a fixed summation order, so the two runs can be compared bit by bit
            e.dy_bf16, e.x_bf16 = int(as_bf and ybf), int(as_bf and xbf)
        rc = _hip.lib().rf_wgrad_grouped(arr, len(shapes), 1, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, _hip.lib().rf_last_error()
        torch.cuda.synchronize()
        results.append(keep)
    for (_, _, dw32, db32, dy, x), (_, _, dw16, db16, _, _), sh in zip(results[0], results[1], shapes):
        assert torch.equal(dw32, dw16), sh
        assert rel_err(db16.cpu(), db32.cpu()) < 1e-6, sh
        assert rel_err(dw16.cpu(), dy.T @ x) < 1e-3, sh
    # fp32- and bf16-operand problems may share a launch (a workgroup-uniform branch picks the loader) ...
    arr = (_hip.WgradEntry * 2)()
    for e, kp, flag in zip(arr, (results[1][1], results[0][2]), (1, 0)):
        kp[2].zero_()
        e.dy, e.x, e.dw, e.db = ptr(kp[0]), ptr(kp[1]), ptr(kp[2]), None
        e.M, e.N, e.K = kp[4].shape[0], kp[4].shape[1], kp[5].shape[1]
        e.ld_dy, e.ld_x, e.splits, e.exclusive, e.dy_bf16, e.x_bf16 = e.N, e.K, 1, 1, flag, flag
    assert _hip.lib().rf_wgrad_grouped(arr, 2, 1, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    for kp in (results[1][1], results[0][2]):
        assert rel_err(kp[2].cpu(), kp[4].T @ kp[5]) < 1e-3
    # ... but one problem's operands share a type, and the fp32 tiled path takes fp32 operands only
    arr[0].x_bf16 = 0
    assert _hip.lib().rf_wgrad_grouped(arr, 2, 1, torch.cuda.current_stream().cuda_stream) != 0
    arr[0].x_bf16 = 1
    assert _hip.lib().rf_wgrad_grouped(arr, 1, 0, torch.cuda.current_stream().cuda_stream) != 0


@pytest.mark.parametrize("M,N,K,res,ln", [(12480, 384, 128, False, False), (325, 128, 128, True, True),
                                          (64, 128, 256, True, True), (1300, 68, 256, True, False),
                                          (7, 128, 128, False, True), (130, 200, 128, False, False)])
def test_rowblock_linear(M, N, K, res, ln):
    """Row-block linear (+residual, +LayerNorm epilogue) vs torch on bf16-rounded operands (the kernel rounds
    x and W to bf16 while staging, accumulates in fp32; residual, bias and the norm are exact fp32)."""
    from routeformer_amd import _hip
    from routeformer_amd._hip import ptr
    g = _g(21)
    x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g); r = torch.randn(M, N, generator=g) if res else None
    gam = torch.rand(N, generator=g) + 0.5; bet = torch.randn(N, generator=g)
    pre = _bf(x) @ _bf(w).T + b + (r if res else 0)
    xd, wd, bd, gd, btd = (v.to(DEV) for v in (x, w, b, gam, bet))
    rd = r.to(DEV) if res else None
    y = torch.full((M, N), float("nan"), device=DEV)
    xhat = torch.empty(M, N, device=DEV) if ln else None
    rstd = torch.empty(M, device=DEV) if ln else None
    assert _hip.lib().rf_rowblock_linear_supported(N, K, 1 if ln else 0)
    rc = _hip.lib().rf_rowblock_linear(ptr(xd), K, ptr(wd), ptr(bd), ptr(rd), N if res else 0, ptr(y), N, M, N, K,
                                       ptr(gd) if ln else None, ptr(btd) if ln else None, ptr(xhat), ptr(rstd), 1e-5,
                                       torch.cuda.current_stream().cuda_stream)
    assert rc == 0, _hip.lib().rf_last_error()
    if ln:
        mean, var = pre.mean(-1, keepdim=True), pre.var(-1, unbiased=False, keepdim=True)
        want_hat = (pre - mean) / torch.sqrt(var + 1e-5)
        assert rel_err(xhat, want_hat) < 1e-4
        assert rel_err(rstd, (1 / torch.sqrt(var + 1e-5)).squeeze(-1)) < 1e-4
        assert rel_err(y, want_hat * gam + bet) < 1e-4
    else:
        assert rel_err(y, pre) < 1e-4


@pytest.mark.parametrize("M,act", [(12480, "gelu"), (325, "relu"), (7, "gelu")])
def test_rowblock_ffn_ln(M, act):
    """LayerNorm(x + conv2(act(conv1(x)))) in one launch vs torch with the same bf16 operand roundings."""
    from routeformer_amd import _hip, kernels as Kn
    from routeformer_amd._hip import ptr
    g = _g(22)
    D, Fd = 128, 256
    x = torch.randn(M, D, generator=g); w1 = torch.randn(Fd, D, generator=g) / math.sqrt(D)
    b1 = torch.randn(Fd, generator=g) * 0.1; w2 = torch.randn(D, Fd, generator=g) / math.sqrt(Fd)
    b2 = torch.randn(D, generator=g) * 0.1
    gam = torch.rand(D, generator=g) + 0.5; bet = torch.randn(D, generator=g)
    z = _bf(x) @ _bf(w1).T + b1
    h = F.gelu(z) if act == "gelu" else F.relu(z)
    pre = x + _bf(h) @ _bf(w2).T + b2
    want = F.layer_norm(pre, (D,), gam, bet, 1e-5)
    d = [v.to(DEV) for v in (x, w1, b1, w2, b2, gam, bet)]
    hd, zd = torch.empty(M, Fd, device=DEV), torch.empty(M, Fd, device=DEV)
    y, xhat, rstd = torch.empty(M, D, device=DEV), torch.empty(M, D, device=DEV), torch.empty(M, device=DEV)
    rc = _hip.lib().rf_rowblock_ffn_ln(ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), ptr(d[4]), ptr(hd), ptr(zd), ptr(y), M,
                                       D, Fd, Kn.ACT[act], ptr(d[5]), ptr(d[6]), ptr(xhat), ptr(rstd), 1e-5,
                                       torch.cuda.current_stream().cuda_stream)
    assert rc == 0, _hip.lib().rf_last_error()
    assert rel_err(zd, z) < 1e-4 and rel_err(hd, h) < 1e-4
    # h is re-rounded to bf16 for the second contraction: a 1-ulp bf16 flip of h moves `pre` by ~1e-3
    assert rel_err(y, want) < 2e-3
    assert rel_err(xhat, (want - bet) / gam) < 2e-3


@pytest.mark.parametrize("M,KC,NOUT,ln,res,dact", [(12480, 384, 128, False, True, 0), (325, 128, 128, False, False, 0),
                                                   (70, 256, 128, False, True, 0), (325, 128, 128, True, False, 0),
                                                   (12480, 128, 256, True, False, 2), (7, 128, 256, True, False, 1)])
def test_rowblock_linear_nn(M, KC, NOUT, ln, res, dact):
    """dX row-block kernel: y = A W (* act'(src)) (+ res), optionally with the LayerNorm-backward prologue, vs
    torch on bf16-rounded operands (A and W are rounded while staged; everything else is fp32)."""
    from routeformer_amd import _hip
    from routeformer_amd._hip import ptr
    g = _g(41)
    w = torch.randn(KC, NOUT, generator=g) / math.sqrt(KC)
    r = torch.randn(M, NOUT, generator=g) if res else None
    src = torch.randn(M, NOUT, generator=g) if dact else None
    dev = lambda v: None if v is None else v.to(DEV)
    y = torch.full((M, NOUT), float("nan"), device=DEV)
    wd, rd, sd_ = dev(w), dev(r), dev(src)
    st = torch.cuda.current_stream().cuda_stream
    if ln:
        dy = torch.randn(M, 128, generator=g); xhat = torch.randn(M, 128, generator=g)
        rstd = torch.rand(M, generator=g) + 0.5; gam = torch.rand(128, generator=g) + 0.5
        gg = dy * gam
        a = rstd[:, None] * (gg - gg.mean(-1, keepdim=True) - xhat * (gg * xhat).mean(-1, keepdim=True))
        dg0, db0 = torch.randn(128, generator=g), torch.randn(128, generator=g)
        t_ = [dev(v) for v in (dy, xhat, rstd, gam, dg0, db0)]
        dpre = torch.empty(M, 128, device=DEV)
        rc = _hip.lib().rf_rowblock_linear_nn(None, 0, ptr(t_[0]), ptr(t_[1]), ptr(t_[2]), ptr(t_[3]), ptr(dpre), ptr(t_[4]),
                                              ptr(t_[5]), ptr(wd), ptr(rd), NOUT if res else 0, ptr(sd_), NOUT if dact else 0,
                                              dact, ptr(y), NOUT, M, KC, NOUT, st)
        assert rc == 0, _hip.lib().rf_last_error()
        assert rel_err(dpre, a) < 2e-5
        assert rel_err(t_[4], dg0 + (dy * xhat).sum(0)) < 2e-5 and rel_err(t_[5], db0 + dy.sum(0)) < 2e-5
    else:
        a = torch.randn(M, KC, generator=g)
        ad = dev(a)
        rc = _hip.lib().rf_rowblock_linear_nn(ptr(ad), KC, None, None, None, None, None, None, None, ptr(wd), ptr(rd),
                                              NOUT if res else 0, ptr(sd_), NOUT if dact else 0, dact, ptr(y), NOUT, M, KC,
                                              NOUT, st)
        assert rc == 0, _hip.lib().rf_last_error()
    want = _bf(a) @ _bf(w)
    if dact == 1:
        want = want * (src > 0).float()
    elif dact == 2:
        cdf = 0.5 * (1 + torch.erf(src / math.sqrt(2.0)))
        want = want * (cdf + src * torch.exp(-0.5 * src * src) / math.sqrt(2 * math.pi))
    if res:
        want = want + r
    # the LN prologue feeds the MFMA with bf16(dpre) computed on the device: a 1-ulp bf16 flip of an operand
    assert rel_err(y, want) < (2e-3 if ln else 1e-4)


@pytest.mark.parametrize("M", [325, 1280])
def test_rowblock_layer_ops_match_unfused(M):
    """linear_add_layer_norm / ffn_add_layer_norm (row-block launches) vs the unfused bf16 path: outputs and
    every gradient (same operand roundings, so the agreement is tight)."""
    from routeformer_amd import kernels as Kn
    Kn.set_precision("bf16")
    g = _g(23)
    D, Fd = 128, 256
    base = dict(a=torch.randn(M, D, generator=g), x=torch.randn(M, D, generator=g),
                w=torch.randn(D, D, generator=g) / math.sqrt(D), b=torch.randn(D, generator=g) * 0.1,
                w1=torch.randn(Fd, D, 1, generator=g) / math.sqrt(D), b1=torch.randn(Fd, generator=g) * 0.1,
                w2=torch.randn(D, Fd, 1, generator=g) / math.sqrt(Fd), b2=torch.randn(D, generator=g) * 0.1,
                g1=torch.rand(D, generator=g) + 0.5, be1=torch.randn(D, generator=g),
                g2=torch.rand(D, generator=g) + 0.5, be2=torch.randn(D, generator=g))
    dy = torch.randn(M, D, generator=g).to(DEV)
    outs = {}
    for fused in (True, False):
        Kn.ROWBLOCK = fused
        try:
            t = {k: v.to(DEV).requires_grad_(True) for k, v in base.items()}
            u = Kn.linear_add_layer_norm(t["a"], t["w"], t["b"], t["x"], t["g1"], t["be1"])
            y = Kn.ffn_add_layer_norm(u, t["w1"], t["b1"], t["w2"], t["b2"], "gelu", t["g2"], t["be2"])
            y.backward(dy)
            outs[fused] = (y.detach(), {k: v.grad for k, v in t.items()})
        finally:
            Kn.ROWBLOCK = True
    assert rel_err(outs[True][0], outs[False][0]) < 2e-3
    for k in base:
        assert rel_err(outs[True][1][k], outs[False][1][k]) < 5e-3, k


@pytest.mark.parametrize("splits,rows,cols,with_bias,with_res", [(1, 5, 832, False, True), (3, 37, 832, True, True),
                                                                 (8, 130, 128, True, False), (13, 9, 100, False, False),
                                                                 (2, 33, 300, True, True), (5, 21, 1024, True, False)])
def test_layernorm_fwd_slabs_vs_torch(splits, rows, cols, with_bias, with_res):
    """rf_layernorm_fwd_slabs (C ABI): LayerNorm(sum of split-K slabs + bias + residual) against torch in double, and
    bit-identical to rf_layernorm_fwd on the slab sum formed in the same order."""
    from routeformer_amd import _hip
    g = _g(splits * 100 + rows)
    slabs = torch.randn(splits, rows, cols, generator=g)
    bias = torch.randn(cols, generator=g) if with_bias else None
    res = torch.randn(rows, cols, generator=g) if with_res else None
    gam, bet = torch.rand(cols, generator=g) + 0.5, torch.randn(cols, generator=g)
    sd, gd, bd = slabs.to(DEV), gam.to(DEV), bet.to(DEV)
    biasd, resd = (None if bias is None else bias.to(DEV)), (None if res is None else res.to(DEV))
    y, xhat, rstd = torch.empty(rows, cols, device=DEV), torch.empty(rows, cols, device=DEV), torch.empty(rows, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t_: None if t_ is None else t_.data_ptr()
    rc = _hip.lib().rf_layernorm_fwd_slabs(p(sd), splits, p(biasd), p(resd), p(gd), p(bd), p(y), p(xhat), p(rstd), rows, cols,
                                           1e-5, st)
    assert rc == 0, _hip.lib().rf_last_error()
    s64 = slabs.double().sum(0) + (0 if bias is None else bias.double()) + (0 if res is None else res.double())
    want = F.layer_norm(s64, (cols,), gam.double(), bet.double(), 1e-5)
    assert rel_err(y, want) < 1e-5
    assert rel_err(rstd, 1.0 / torch.sqrt(s64.var(1, unbiased=False) + 1e-5)) < 1e-5
    # the slab sum in the slab-sum launch's order (s0 + s1 + ..., then the bias), handed to the plain entry point
    acc = sd[0].clone()
    for i in range(1, splits):
        acc += sd[i]
    if biasd is not None:
        acc += biasd
    y2, xh2, rs2 = torch.empty_like(y), torch.empty_like(y), torch.empty_like(rstd)
    rc = _hip.lib().rf_layernorm_fwd(p(acc), p(resd), p(gd), p(bd), p(y2), p(xh2), p(rs2), rows, cols, 1e-5, st)
    assert rc == 0, _hip.lib().rf_last_error()
    if cols > 256:  # wide rows: both entry points run the row-per-workgroup kernel -- the same reduction tree
        assert torch.equal(y, y2) and torch.equal(xhat, xh2) and torch.equal(rstd, rs2)
    else:
        assert rel_err(y, y2) < 1e-6 and rel_err(xhat, xh2) < 1e-6 and rel_err(rstd, rs2) < 1e-6


@pytest.mark.parametrize("C", [128, 100, 832])
def test_layernorm_strided_input(C):
    """add_layer_norm on strided views (the consumed tail of a (B, L, C) activation, a row-pitched 2-D view) reads them in
    place (rf_layernorm_fwd_strided): bit-identical to the norm of a contiguous copy, gradient included."""
    from routeformer_amd import kernels as Kn
    g = _g(C)
    full = torch.randn(3, 11, C, generator=g).to(DEV)
    gam, bet = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    res = torch.randn(3, 4, C, generator=g).to(DEV)
    for view, r in ((lambda t_: t_[:, -4:, :], res), (lambda t_: t_[:, 2:3, :], None),
                    (lambda t_: t_.reshape(33, C)[:, :].as_strided((16, C), (2 * C, 1)), None)):
        outs = []
        for contiguous in (False, True):
            src = full.clone().requires_grad_(True)
            x = view(src)
            assert not x.is_contiguous() or x.numel() == C * 3
            if contiguous:
                x = x.contiguous()
            rr = None if r is None else r.reshape(x.shape)
            y = Kn.add_layer_norm(x, rr, gam, bet)
            y.backward(torch.ones_like(y) * 0.5 + y.detach())
            outs.append((y.detach(), src.grad))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        want = F.layer_norm(view(full).double() + (0 if r is None else r.reshape(view(full).shape).double()), (C,),
                            gam.double(), bet.double(), 1e-5)
        assert rel_err(outs[0][0], want.cpu()) < 1e-5


@pytest.mark.parametrize("M", [320, 560, 96, 40])
@pytest.mark.parametrize("prec", ["bf16", "f32"])
def test_slab_layernorm_layer_ops_match_unfused(M, prec):
    """The GPS backbone's d_model = 832 layer tail (out-projection + residual + norm1, Conv1d pair + residual + norm2:
    layers/TransformerEncoderDecoder.py:44-53) with the split-K slabs summed inside the norm (kernels.SLAB_LN) against the
    product + slab-sum launch + norm path: the forward must be BIT-identical (same kernels, same summation order), and so
    are the gradients (the backward is the same composition of launches)."""
    from routeformer_amd import kernels as Kn
    Kn.set_precision(prec)
    g = _g(29)
    D, Fd = 832, 3328
    base = dict(a=torch.randn(M, D, generator=g), x=torch.randn(M, D, generator=g),
                w=torch.randn(D, D, generator=g) / math.sqrt(D), b=torch.randn(D, generator=g) * 0.1,
                w1=torch.randn(Fd, D, 1, generator=g) / math.sqrt(D), b1=torch.randn(Fd, generator=g) * 0.1,
                w2=torch.randn(D, Fd, 1, generator=g) / math.sqrt(Fd), b2=torch.randn(D, generator=g) * 0.1,
                g1=torch.rand(D, generator=g) + 0.5, be1=torch.randn(D, generator=g),
                g2=torch.rand(D, generator=g) + 0.5, be2=torch.randn(D, generator=g))
    dy = torch.randn(M, D, generator=g).to(DEV)
    outs, saved = {}, Kn.SLAB_LN
    try:
        for fused in (True, False):
            Kn.SLAB_LN = fused
            t = {k: v.to(DEV).requires_grad_(True) for k, v in base.items()}
            if fused:  # the conv pair's second product always needs slabs at these shapes: the fused path is what runs
                assert Kn._partials_plan(t["x"].data_ptr(), Fd, t["w2"].view(D, Fd), M, D, Fd) is not None
            u = Kn.linear_add_layer_norm(t["a"], t["w"], t["b"], t["x"], t["g1"], t["be1"])
            y = Kn.ffn_add_layer_norm(u, t["w1"], t["b1"], t["w2"], t["b2"], "gelu", t["g2"], t["be2"])
            if fused:
                assert type(y.grad_fn).__name__.startswith("_FFNAddLNSlabs")
            y.backward(dy)
            with torch.no_grad():  # the no-grad form (eval): no saves
                y0 = Kn.ffn_add_layer_norm(Kn.linear_add_layer_norm(t["a"], t["w"], t["b"], t["x"], t["g1"], t["be1"]),
                                           t["w1"], t["b1"], t["w2"], t["b2"], "gelu", t["g2"], t["be2"])
            outs[fused] = (y.detach(), {k: v.grad for k, v in t.items()}, y0)
    finally:
        Kn.SLAB_LN = saved
    assert torch.equal(outs[True][0], outs[False][0])
    assert torch.equal(outs[True][2], outs[False][2]) and torch.equal(outs[True][2], outs[True][0])
    for k in base:
        assert rel_err(outs[True][1][k], outs[False][1][k]) < 1e-6, k


@pytest.mark.parametrize("B,L,prec", [(8, 40, "bf16"), (8, 21, "bf16"), (3, 7, "f32"), (2, 2, "bf16"), (4, 5, "bf16")])
def test_norm_writes_distil_im2col_image(B, L, prec):
    """Round 4: the last norm of an Informer encoder layer writes its output straight as the im2col image of the distilling
    convolution that follows (Conv1d k = 3, circular padding 2, layers/TransformerEncoderDecoder.py:12-18;
    rf_layernorm_fwd_slabs_unfold), and its backward folds the image's gradient on load (rf_layernorm_bwd_fold) -- the
    unfold / fold launches between norm and product are gone.  Against the unfused composition (norm, then
    circular_conv3(pad=2)): the image is BIT-identical to rf_unfold3_circular of the norm's output, the convolution's
    output too, gradients agree to summation order."""
    from routeformer_amd import kernels as Kn
    Kn.set_precision(prec)
    g = _g(31)
    D, Fd = 832, 3328
    base = dict(x=torch.randn(B, L, D, generator=g), w1=torch.randn(Fd, D, 1, generator=g) / math.sqrt(D),
                b1=torch.randn(Fd, generator=g) * 0.1, w2=torch.randn(D, Fd, 1, generator=g) / math.sqrt(Fd),
                b2=torch.randn(D, generator=g) * 0.1, g2=torch.rand(D, generator=g) + 0.5, be2=torch.randn(D, generator=g),
                wc=torch.randn(D, D, 3, generator=g) / math.sqrt(3 * D), bc=torch.randn(D, generator=g) * 0.1)
    dz = torch.randn(B, L + 2, D, generator=g).to(DEV)
    outs = {}
    for fused in (True, False):
        t = {k: v.to(DEV).requires_grad_(True) for k, v in base.items()}
        if fused:
            y, is_image = Kn.ffn_add_layer_norm(t["x"], t["w1"], t["b1"], t["w2"], t["b2"], "relu", t["g2"], t["be2"], unfold=True)
            assert is_image and tuple(y.shape) == (B, L + 2, 3 * D)
            z = Kn.circular_conv3_unfolded(y, t["wc"], t["bc"])
        else:
            y = Kn.ffn_add_layer_norm(t["x"], t["w1"], t["b1"], t["w2"], t["b2"], "relu", t["g2"], t["be2"])
            z = Kn.circular_conv3(y, t["wc"], t["bc"], pad=2)
        z.backward(dz)
        outs[fused] = (y.detach(), z.detach(), {k: v.grad for k, v in t.items()})
    image = torch.empty(B, L + 2, 3 * D, device=DEV)
    from routeformer_amd import _hip
    src = outs[False][0].contiguous()
    _hip.check(_hip.lib().rf_unfold3_circular(src.data_ptr(), image.data_ptr(), B, L, D, 2, torch.cuda.current_stream().cuda_stream), "rf_unfold3_circular")
    torch.cuda.synchronize()
    assert torch.equal(outs[True][0], image)
    assert torch.equal(outs[True][1], outs[False][1])
    # bf16 mode: the fold adds its (up to six) terms in another order than rf_fold3_circular; where that moves the fp32 sum by an
    # ulp across a bf16 rounding boundary, one operand of the weight-gradient products changes by 2^-9 -- with M = 4 rows
    # (B = 2, L = 2) that is up to 1e-3 of a gradient element (observed 7e-4); fp32 mode has no such amplifier
    for k in base:
        assert rel_err(outs[True][2][k], outs[False][2][k]) < (2e-3 if prec == "bf16" else 2e-6), k


@pytest.mark.parametrize("B,P,E,extra,normalize,dense_on", [(8, 30, 64, 0, False, True), (3, 7, 16, 5, True, True),
                                                              (4, 30, 64, 0, True, False), (1, 1, 4, 0, False, True)])
def test_traj_head(B, P, E, extra, normalize, dense_on):
    """Fused postprocess + discounted SmoothL1 (positions and dense head) + ADE/FDE vs the oracle's losses
    (routeformer.py:367-374, future_discounted_mse.py:56-95, full_comparison.py:497-521), values and d/d out."""
    from routeformer_amd import kernels as Kn
    g = _g(5)
    C = 2 + E + extra
    out = torch.randn(B, P, C, generator=g) * 1.5
    last = torch.randn(B, 1, 2, generator=g)
    tgt = torch.randn(B, P, 2, generator=g) * 3
    tvis = torch.randn(B, P, E, generator=g)
    gamma, ratio, mstd, mmean = 0.97, 0.3, (2.5 if normalize else 1.0), (0.1 if normalize else 0.0)
    o = out.clone().requires_grad_(True)
    pos = last + torch.cumsum(o[:, :, :2] * mstd + mmean, dim=1)
    traj = O.future_discounted_loss(pos, tgt, gamma, "smooth_l1")
    dense = O.future_discounted_loss(o[:, :, 2:2 + E], tvis, gamma, "smooth_l1")
    w = (ratio * traj / torch.clamp(dense, min=1e-6)).detach() if dense_on else 0
    loss = traj + w * dense
    (loss * 1.7).backward()
    od = out.to(DEV).requires_grad_(True)
    lastd, tgtd, tvisd = last.to(DEV), tgt.to(DEV), tvis.to(DEV)
    l, tr, de, a, f, p = Kn.traj_head(od, lastd, tgtd, tvisd, gamma, ratio, dense_on, mstd, mmean)
    (l * 1.7).backward()
    assert rel_err(p, pos.detach()) < 1e-5
    for got, want in ((l, loss), (tr, traj), (de, dense), (a, O.ade(pos, tgt)), (f, O.fde(pos, tgt))):
        assert abs(got.item() - want.item()) <= 2e-5 * max(1.0, abs(want.item()))
    assert rel_err(od.grad, o.grad) < 2e-5


def test_cpu_tensor_is_refused():
    """No CPU fallback: handing the product a CPU tensor must fail loudly."""
    from routeformer_amd import _hip, kernels as Kn
    with pytest.raises(_hip.HipLibraryError):
        Kn.linear(torch.randn(4, 8), torch.randn(8, 8).to(DEV), None)


# ------------------------------------------------------------------------------------------------
# nn.Dropout: device-side Philox masks (csrc/philox.h), regenerated in backward
# ------------------------------------------------------------------------------------------------
def test_dropout_kernel_statistics_and_consistency():
    """Keep-rate 1 - p and 1/(1-p) scaling; the backward launch regenerates the forward's mask; masks are a pure
    function of (seed, step, site): reproducible, and different for another site, step or seed; odd lengths."""
    from routeformer_amd import kernels as Kn
    Kn.RNG.manual_seed(123)
    Kn.RNG.begin_step(torch.device(DEV))
    for n, p in ((1 << 20, 0.05), (65 * 128 * 3 + 3, 0.3), (7, 0.5)):
        x = (torch.rand(n, device=DEV) + 0.5).requires_grad_()
        Kn.RNG.site = 5
        y = Kn.dropout(x, p)
        keep = y != 0
        if n > 1000:
            rate = float(keep.float().mean())
            assert abs(rate - (1 - p)) < 4 * math.sqrt(p * (1 - p) / n) + 1e-4, (n, p, rate)
        assert torch.allclose(y[keep], x.detach()[keep] / (1 - p), rtol=1e-6)
        w = torch.rand(n, device=DEV) + 0.5
        (y * w).sum().backward()
        assert torch.equal(x.grad != 0, keep) and torch.allclose(x.grad[keep], w[keep] / (1 - p), rtol=1e-6)
        Kn.RNG.site = 5
        assert torch.equal(Kn.dropout(x.detach(), p) != 0, keep), "same (seed, step, site) must reproduce the mask"
        assert torch.equal(Kn.RNG.materialise(5, (n,), p, torch.device(DEV)), keep)
        Kn.RNG.site = 6
        other = Kn.dropout(x.detach(), p) != 0
        if n > 1000:
            assert not torch.equal(other, keep)
            # independence of two sites: P(both kept) = (1-p)^2
            assert abs(float((other & keep).float().mean()) - (1 - p) ** 2) < 0.01
    x = torch.ones(1 << 16, device=DEV)
    Kn.RNG.site = 0
    a = Kn.dropout(x, 0.2) != 0
    Kn.RNG.begin_step(torch.device(DEV))  # step += 1 on the device
    b = Kn.dropout(x, 0.2) != 0
    Kn.RNG.manual_seed(124)
    Kn.RNG.begin_step(torch.device(DEV))
    c = Kn.dropout(x, 0.2) != 0
    assert not torch.equal(a, b) and not torch.equal(a, c) and not torch.equal(b, c)
    assert Kn.dropout(x, 0.0) is x and Kn.dropout(x, 0.5, training=False) is x


@pytest.mark.parametrize("LQ,LK,H,E,causal", [(40, 40, 8, 8, False), (30, 30, 4, 16, True), (21, 9, 2, 104, False)])
def test_attention_probability_dropout_vs_oracle(LQ, LK, H, E, causal):
    """FullAttention's dropout on the softmax probabilities, generated INSIDE the attention kernels (forward and
    backward regenerate the same Philox mask): the mask the product used is materialised and handed to the CPU oracle."""
    from routeformer_amd import kernels as Kn
    B, p = 3, 0.25
    g = _g(LQ + LK + E)
    q, k, v = (torch.randn(B, L, H, E, generator=g).requires_grad_() for L in (LQ, LK, LK))
    Kn.RNG.manual_seed(7)
    Kn.RNG.begin_step(torch.device(DEV))
    Kn.RNG.record = []
    qd = q.detach().reshape(B * LQ, H * E).to(DEV).requires_grad_()
    kv = torch.cat([k.detach().reshape(B * LK, H * E), v.detach().reshape(B * LK, H * E)], dim=1).to(DEV).requires_grad_()
    try:
        if causal:
            every = torch.arange(LQ, device=DEV, dtype=torch.int32).expand(B, H, LQ).contiguous()
            ctx = Kn.attention(qd, kv, (0, 0, H * E), (B, H, LQ, LK, E), 2, n_top=LQ, forced_top=every, drop_p=p)
        else:
            ctx = Kn.attention(qd, kv, (0, 0, H * E), (B, H, LQ, LK, E), 0, drop_p=p)
        used = Kn.RNG.record
    finally:
        Kn.RNG.record = None
    assert len(used) == 1 and tuple(used[0].shape) == (B, H, LQ, LK)
    assert abs(float(used[0].float().mean()) - (1 - p)) < 0.02
    ref = O.full_attention(q, k, v, masked=causal, dropout=p, drop=O.DropoutSource([used[0].cpu()]))
    w = torch.randn(ref.shape, generator=g)
    (ref * w).sum().backward()
    (ctx * w.to(DEV)).sum().backward()
    assert rel_err(ctx, ref) < 3e-5
    HE = H * E
    assert rel_err(qd.grad.view(B, LQ, H, E), q.grad) < 5e-5
    assert rel_err(kv.grad[:, :HE].reshape(B, LK, H, E), k.grad) < 5e-5
    assert rel_err(kv.grad[:, HE:].reshape(B, LK, H, E), v.grad) < 5e-5


# ------------------------------------------------------------------------------------------------
# fused per-sequence encoder stack (csrc/seqlayer.hip)
# ------------------------------------------------------------------------------------------------
def _bf(t_):
    return t_.to(torch.bfloat16).to(torch.float32)


def _emulate_stack(x, layers, idx_tabs, idx_group, factor, act, forced_tops=None):
    """CPU restatement of the fused kernel's arithmetic: the reference's EncoderLayer (cross_modal_transformer.py:288-301
    with ProbAttention :88-166) where every matrix-core operand is rounded to bf16 (x, weights, q / k / v, the
    softmax probabilities, ctx, x1, h) EXCEPT on the way to the discontinuous top-u selection (q / k projection, sampled
    scores: ~fp32), and everything else is fp32.  -> (y, per-layer saves, per-layer selections)."""
    B, L, D = x.shape
    H, E = 8, 16
    saves, tops = [], []
    for li, W in enumerate(layers):
        sample_k, n_top = O.prob_sizes(L, L, factor)
        # q / k projection and the scores of the sparsity measure in split-bf16 (hi + lo operands: ~2^-16, i.e. fp32 here);
        # v, the softmax rows and everything downstream on bf16-rounded operands
        qkv = torch.cat([x @ W["wqkv"][:2 * D].t(), _bf(x) @ _bf(W["wqkv"][2 * D:]).t()], dim=-1) + W["bqkv"]
        qf, kf = (qkv[..., i * D:(i + 1) * D].view(B, L, H, E).transpose(1, 2) for i in range(2))
        S_meas = qf @ kf.transpose(-1, -2)
        q, k, v = (_bf(qkv[..., i * D:(i + 1) * D]).view(B, L, H, E).transpose(1, 2) for i in range(3))  # (B,H,L,E)
        S = q @ k.transpose(-1, -2)
        top_l = []
        ctx = v.mean(dim=2, keepdim=True).expand(B, H, L, E).clone()
        for b in range(B):
            idx = idx_tabs[li][b // idx_group].long()
            samp = torch.gather(S_meas[b], 2, idx.unsqueeze(0).expand(H, L, sample_k))
            Mm = samp.max(-1).values - samp.sum(-1) / L
            if forced_tops is not None:
                top = forced_tops[li][b].long()
            else:
                top = Mm.topk(n_top, dim=-1).indices.sort(dim=-1).values
            top_l.append(top)
            for h in range(H):
                P = torch.softmax(S[b, h, top[h]] * (1.0 / math.sqrt(E)), dim=-1)
                ctx[b, h, top[h]] = _bf(P) @ v[b, h]
        tops.append(torch.stack(top_l))
        ctx = _bf(ctx.transpose(1, 2).reshape(B, L, D))
        pre1 = ctx @ _bf(W["wo"]).t() + W["bo"] + x
        x1 = F.layer_norm(pre1, (D,), W["g1"], W["be1"], 1e-5)
        z = _bf(x1) @ _bf(W["w1"]).t() + W["b1"]
        h_ = F.gelu(z) if act == "gelu" else F.relu(z)
        pre2 = _bf(h_) @ _bf(W["w2"]).t() + W["b2"] + x1
        y = F.layer_norm(pre2, (D,), W["g2"], W["be2"], 1e-5)
        saves.append(dict(qkv=qkv, ctx=ctx, x1=x1, z=z, h=h_, y=y,
                          xhat1=(pre1 - pre1.mean(-1, keepdim=True)) / torch.sqrt(pre1.var(-1, unbiased=False, keepdim=True) + 1e-5),
                          xhat2=(pre2 - pre2.mean(-1, keepdim=True)) / torch.sqrt(pre2.var(-1, unbiased=False, keepdim=True) + 1e-5)))
        x = y
    return x, saves, tops


@pytest.mark.parametrize("B,L,F_,n_layers,groups,act", [(6, 65, 256, 3, 3, "gelu"), (4, 40, 256, 2, 1, "gelu"),
                                                         (3, 80, 64, 2, 1, "relu"), (5, 17, 128, 1, 1, "gelu")])
def test_fused_encoder_stack_forward(B, L, F_, n_layers, groups, act):
    """rf_seqlayer_fwd (all layers of an encoder stack in one launch, one workgroup per sequence) against the bf16-
    operand restatement above: outputs, every tensor saved for the backward pass, and the ProbSparse selections --
    free-running (selections must agree) and with the restatement's selections imposed (tight tolerance)."""
    from routeformer_amd import kernels as Kn
    Kn.set_precision("bf16")
    g = _g(B * 7 + L)
    D = 128
    x = torch.randn(B, L, D, generator=g)
    layers = []
    for _ in range(n_layers):
        W = dict(wqkv=torch.randn(3 * D, D, generator=g) / math.sqrt(D), bqkv=0.1 * torch.randn(3 * D, generator=g),
                 wo=torch.randn(D, D, generator=g) / math.sqrt(D), bo=0.1 * torch.randn(D, generator=g),
                 w1=torch.randn(F_, D, generator=g) / math.sqrt(D), b1=0.1 * torch.randn(F_, generator=g),
                 w2=torch.randn(D, F_, generator=g) / math.sqrt(F_), b2=0.1 * torch.randn(D, generator=g),
                 g1=1 + 0.1 * torch.randn(D, generator=g), be1=0.1 * torch.randn(D, generator=g),
                 g2=1 + 0.1 * torch.randn(D, generator=g), be2=0.1 * torch.randn(D, generator=g))
        layers.append(W)
    sample_k, n_top = O.prob_sizes(L, L, 5)
    per = B // groups
    idx_tabs = [torch.randint(L, (groups, L, sample_k), generator=g) for _ in range(n_layers)]
    y_ref, saves, tops = _emulate_stack(x, layers, idx_tabs, per, 5, act)

    stride = Kn.seqstack_pack_bytes(F_)
    wpack = torch.zeros(n_layers * stride, dtype=torch.uint8, device=DEV)
    Kn.seqstack_pack([{k: v.to(DEV) for k, v in W.items()} for W in layers], wpack, stride)
    idx_d = [t_.to(torch.int32).to(DEV) for t_ in idx_tabs]
    xd = x.reshape(B * L, D).to(DEV)
    free = Kn._seqstack_launch(xd, wpack, stride, idx_d, per, B, L, F_, act, sample_k, n_top, True, None, 1e-5)
    torch.cuda.synchronize()
    same = [torch.equal(free["top"][li].cpu().long(), tops[li]) for li in range(n_layers)]
    assert same[0], "first-layer selections differ from the restatement"
    forced = Kn._seqstack_launch(xd, wpack, stride, idx_d, per, B, L, F_, act, sample_k, n_top, True,
                                 [t_.to(torch.int32) for t_ in tops], 1e-5)
    nosave = Kn._seqstack_launch(xd, wpack, stride, idx_d, per, B, L, F_, act, sample_k, n_top, False,
                                 [t_.to(torch.int32) for t_ in tops], 1e-5)
    torch.cuda.synchronize()
    M = B * L
    for li in range(n_layers):
        # (with the bf16 input images `xin` only the LAST layer's output is stored: the intermediate ones have no reader)
        per_layer_y = forced["y"].shape[0] == n_layers
        assert per_layer_y or "xin" in forced
        for name in ("qkv", "ctx", "x1", "xhat1", "h", "xhat2") + (("y",) if per_layer_y else ()) + (("z",) if act == "gelu" else ()):
            got, want = forced[name][li].float().cpu(), saves[li][name].reshape(M, -1)
            # ctx is saved as the bf16 image the out-projection consumes: one bf16 ulp (2^-8) when a rounding flips; the
            # same holds for whatever kernels.BF16_SAVES keeps as bf16 (x1, h: the images conv1 / conv2 consume)
            tol = 1e-2 if (name == "ctx" or forced[name].dtype == torch.bfloat16) else 4e-3 * (li + 1)
            assert rel_err(got, want) < tol, (li, name, rel_err(got, want))
        assert forced["rstd1"][li].shape == (M,) and bool(torch.isfinite(forced["rstd1"][li]).all())
        assert bool((forced["rstd2"][li] > 0).all())
        if "xin" in forced:  # every layer's input as the projection consumed it (bf16): x, then the previous layer's output
            want = (x if li == 0 else saves[li - 1]["y"]).reshape(M, -1)
            assert rel_err(forced["xin"][li].float().cpu(), want) < 1e-2, (li, "xin")
    assert rel_err(forced["y"][-1].cpu(), y_ref.reshape(M, D)) < 4e-3 * n_layers
    assert torch.equal(nosave["y"][0], forced["y"][-1]), "the no-save variant must compute the same output"
    if all(same):
        assert torch.equal(free["y"][-1], forced["y"][-1])


# ------------------------------------------------------------------------------------------------
# video ingest: area resize, content hash, HBM token cache
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,factor", [((2, 3, 3, 64, 96), 0.5), ((1, 2, 3, 90, 120), 1 / 3), ((3, 1, 60, 80), 0.6),
                                           ((2, 3, 37, 53), 0.75), ((1, 1, 8, 8), 1.0)])
def test_resize_area_vs_restatement(shape, factor):
    """rf_resize_area against the CPU restatement of cv2.INTER_AREA down-scaling (oracle/ingest_oracle.py; parity
    with OpenCV itself is unpinned: cv2 is absent) and, for integer factors, against exact block means."""
    from oracle import ingest_oracle as IO
    from routeformer_amd.utils.video import resize_area
    g = _g(int(shape[-1] * 10 * factor))
    x = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
    got = resize_area(x.to(DEV), factor).cpu().numpy()
    want = IO.resize_area(x.numpy(), factor)
    assert got.shape == want.shape
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert diff.max() <= 1 and (diff > 0).mean() < 2e-3, (diff.max(), (diff > 0).mean())  # fp32 vs fp64 at exact .5 ties
    inv = 1.0 / factor
    if abs(inv - round(inv)) < 1e-9 and shape[-2] % round(inv) == 0 and shape[-1] % round(inv) == 0:
        s = int(round(inv))
        blocks = x.numpy().astype(np.float64).reshape(shape[:-2] + (shape[-2] // s, s, shape[-1] // s, s)).mean(axis=(-3, -1))
        if s == 2:  # OpenCV's 2 x 2 fast path: (a + b + c + d + 2) >> 2, i.e. a half rounds UP (bit-exact here)
            assert np.array_equal(got, np.floor(blocks + 0.5).astype(np.uint8)) and np.array_equal(got, want)
        else:
            assert np.array_equal(got, np.rint(blocks).astype(np.uint8))


def test_token_cache_semantics():
    """Content-keyed HBM cache of trunk tokens: keys depend on the bytes only (not on the address, the clip layout or
    the frame sub-sampling), equal frames share a slot, lookups after an insert hit, a full cache degrades to "uncached",
    and a saved cache answers in a new process-equivalent instance (persistent=True of torchcache)."""
    from oracle import ingest_oracle as IO
    from routeformer_amd.models.video_backbone import TokenCache
    g = _g(3)
    B, T, H, W = 2, 6, 16, 24
    vid = torch.randint(0, 256, (B, T, 3, H, W), generator=g, dtype=torch.uint8)
    vid[1, 4] = vid[0, 1]                       # an identical frame inside the batch
    v = vid.to(DEV)
    idx = torch.tensor([1, 3, 4])
    cache = TokenCache(5, DEV)                  # room for 5 frames, the batch selects 6 (5 distinct)
    keys = cache.keys_of([(v, idx)])
    assert keys.shape == (B * 3,) and int(keys[0]) == int(keys[5]) and len(set(keys.tolist())) == 5
    sub = v[:, idx].contiguous()                # the same frames at other addresses, as a compact clip
    assert torch.equal(cache.keys_of([(sub, None)]), keys)
    assert not torch.equal(cache.keys_of([(sub.flip(-1).contiguous(), None)]), keys)
    slots, miss = cache.lookup(keys)
    assert miss == 6 and bool((slots < 0).all())
    tokens = torch.randn(6, 65, 240, device=DEV)
    tokens[5] = tokens[0]
    final = cache.insert(keys, slots, tokens)
    model = IO.TokenCacheModel(5)
    want = model.insert(keys.tolist())
    assert sorted(set(final.tolist())) == sorted(set(want)) and int(final[0]) == int(final[5])
    slots2, miss2 = cache.lookup(keys)
    assert miss2 == 0 and torch.equal(slots2, final) and torch.equal(cache.gather(slots2), tokens)
    other = torch.randint(0, 256, (1, 2, 3, H, W), generator=g, dtype=torch.uint8).to(DEV)
    k2 = cache.keys_of([(other, None)])
    s2, m2 = cache.lookup(k2)
    f2 = cache.insert(k2, s2, torch.randn(2, 65, 240, device=DEV))
    assert m2 == 2 and int((f2 >= 0).sum()) == 0 and int(cache.next_slot) >= 5, "5 slots were taken: the cache is full"
    again = TokenCache(5, DEV)
    again.load_state_dict(cache.state_dict())
    s3, m3 = again.lookup(keys)
    assert m3 == 0 and torch.equal(again.gather(s3), tokens)
    # key namespace: the stored tokens also depend on the trunk's weights and on the arithmetic mode
    from routeformer_amd import kernels as Kn
    ns = TokenCache(5, DEV)
    ns.bind(0x1234567)
    k_f32 = ns.keys_of([(v, idx)])
    assert not torch.equal(k_f32, keys), "weight fingerprint must change the keys"
    Kn.set_precision("bf16")
    try:
        assert not torch.equal(ns.keys_of([(v, idx)]), k_f32), "bf16-mode tokens must not answer an fp32-mode lookup"
    finally:
        Kn.set_precision("f32")
    ns.insert(k_f32, ns.lookup(k_f32)[0], tokens)
    with pytest.raises(ValueError, match="other backbone weights"):
        ns.bind(0x7654321)
    foreign = TokenCache(5, DEV)
    foreign.bind(0x7654321)
    with pytest.raises(ValueError, match="fingerprint mismatch"):
        foreign.load_state_dict(ns.state_dict())


def test_pad_cols_roundtrip_and_gradient():
    """rf_pad_cols / rf_unpad_cols (the K-padded copy of the GPS token-embedding weight, 207 -> 208 columns) against
    F.pad and its gradient; the slot-accumulating form adds into a buffer.  Bit-exact."""
    from routeformer_amd import _hip, kernels as Kn
    g = _g(43)
    w = torch.randn(832, 69, 3, generator=g)
    wd = w.to(DEV).requires_grad_(True)
    wp = Kn._PadCols.apply(wd.reshape(832, 207), 208, None)
    assert torch.equal(wp.detach().cpu(), F.pad(w.reshape(832, 207), (0, 1)))
    dwp = torch.randn(832, 208, generator=g)
    wp.backward(dwp.to(DEV))
    assert torch.equal(wd.grad.cpu(), dwp[:, :207].reshape(832, 69, 3))
    slot = torch.ones(832, 207, device=DEV)
    rc = _hip.lib().rf_unpad_cols(dwp.to(DEV).data_ptr(), slot.data_ptr(), 832, 207, 208, 1, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, _hip.lib().rf_last_error()
    assert torch.equal(slot.cpu(), 1.0 + dwp[:, :207])


def test_gather_frames():
    """rf_gather_frames (the engine's clip staging: dst[b][f] = src[b][idx[f]] for all camera streams in one launch)
    against torch.index_select -- fp16, uint8 (odd frame size: byte tail) and fp32 clips in one call.  Bit-exact."""
    from routeformer_amd import _hip
    g = _g(41)
    clips = [torch.randn(3, 10, 3, 20, 28, generator=g).half(), torch.randint(0, 255, (2, 7, 3, 11, 13), generator=g, dtype=torch.uint8),
             torch.randn(2, 5, 4, 6, generator=g)]
    idxs = [torch.tensor([9, 0, 4, 4]), torch.tensor([6, 5, 1]), torch.tensor([2])]
    arr = (_hip.GatherEntry * len(clips))()
    keep = []
    for e, v, idx in zip(arr, clips, idxs):
        vd, idd = v.to(DEV), idx.to(DEV)
        dst = torch.zeros((v.shape[0], idx.numel()) + tuple(v.shape[2:]), device=DEV, dtype=v.dtype)
        keep.append((vd, idd, dst))
        e.src, e.dst, e.idx = vd.data_ptr(), dst.data_ptr(), idd.data_ptr()
        e.B, e.T, e.F, e.pad, e.frame_bytes = v.shape[0], v.shape[1], idx.numel(), 0, v[0, 0].numel() * v.element_size()
    rc = _hip.lib().rf_gather_frames(arr, len(clips), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, _hip.lib().rf_last_error()
    for (vd, idd, dst), v, idx in zip(keep, clips, idxs):
        assert torch.equal(dst.cpu(), torch.index_select(v, 1, idx))
