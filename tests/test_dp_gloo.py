"""CPU, 2 ranks over gloo: the bucketed overlapped gradient exchange of ``GradReducer`` reproduces
"N independent replicas + gradient mean" (SURVEY 8(e)), including parameters that get no gradient."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(8, 16)
        self.b = torch.nn.Linear(16, 16)
        self.unused = torch.nn.Linear(4, 4)
        self.c = torch.nn.Linear(16, 2)

    def forward(self, x):
        return self.c(torch.tanh(self.b(torch.tanh(self.a(x)))))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from routeformer_amd.engine import GradReducer
    torch.manual_seed(0)
    net = _Net()
    if rank == 1:  # ranks start different: broadcast must make them equal
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    red = GradReducer(list(net.parameters()), bucket_mb=0.0001)  # ~26 floats per bucket -> several buckets
    red.broadcast_parameters(0)
    assert len(red.buckets) >= 3
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(5, 8, generator=g)
    for _ in range(2):  # two steps: the zero()/finish() protocol must be re-usable
        red.zero()
        net(x).square().sum().backward()
        scale = red.finish()
    out[rank] = (torch.cat([(p.grad * scale).reshape(-1) for p in net.parameters()]), red.flat_param.clone(), x)
    dist.destroy_process_group()


def test_bucketed_allreduce_matches_replica_mean():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        g0, p0, x0 = out[0]
        g1, p1, x1 = out[1]
    assert torch.equal(p0, p1), "parameters not broadcast"
    assert torch.allclose(g0, g1, atol=0, rtol=0), "ranks disagree after all-reduce"
    torch.manual_seed(0)
    ref = _Net()
    grads = []
    for x in (x0, x1):
        ref.zero_grad()
        ref(x).square().sum().backward()
        grads.append(torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                                for p in ref.parameters()]))
    assert torch.allclose(g0, (grads[0] + grads[1]) / 2, atol=1e-6)


def _worker_prefix(rank, world, port, out):
    """The graph-replay protocol: no hooks; after "stage 1" the buckets made up only of one sub-module's parameters go
    out early (launch_complete_prefix), the rest in finish(); neighbouring buckets leave as ONE collective."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from routeformer_amd.engine import GradReducer
    torch.manual_seed(0)
    net = _Net()
    red = GradReducer(list(net.parameters()), bucket_mb=0.0001)
    red.hooks_enabled = False
    red.broadcast_parameters(0)
    names = {id(p): n for n, p in net.named_parameters()}
    calls = []
    real = dist.all_reduce

    def counting(t, *a, **kw):
        calls.append(t.numel())
        return real(t, *a, **kw)

    dist.all_reduce = counting
    try:
        g = torch.Generator().manual_seed(100 + rank)
        x = torch.randn(5, 8, generator=g)
        red.zero()
        net(x).square().sum().backward()
        assert not calls, "no collective may go out from hooks in this mode"
        early = red.launch_complete_prefix(names, "c.")  # the last layer's gradients are final first
        n_early = len(calls)
        scale = red.finish()
    finally:
        dist.all_reduce = real
    out[rank] = (torch.cat([(p.grad * scale).reshape(-1) for p in net.parameters()]), early, n_early, len(calls),
                 len(red.buckets), sum(calls), red.flat_grad.numel())
    dist.destroy_process_group()


def test_prefix_launch_and_run_coalescing():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_prefix, args=(world, port, out), nprocs=world, join=True)
        r0, r1 = out[0], out[1]
    assert torch.equal(r0[0], r1[0]), "ranks disagree after all-reduce"
    _, early, n_early, n_calls, n_buckets, elems, total = r0
    assert early >= 1 and n_early == 1, "the prefix buckets are neighbours: one collective"
    assert n_calls == 2 and n_buckets >= 3, "the remaining buckets are neighbours too: one more collective"
    assert elems == total, "every element reduced exactly once"
    torch.manual_seed(0)
    ref = _Net()
    grads = []
    for rank in range(world):
        x = torch.randn(5, 8, generator=torch.Generator().manual_seed(100 + rank))
        ref.zero_grad()
        ref(x).square().sum().backward()
        grads.append(torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ref.parameters()]))
    assert torch.allclose(r0[0], (grads[0] + grads[1]) / 2, atol=1e-6)


def test_single_process_reducer_is_a_noop():
    from routeformer_amd.engine import GradReducer
    net = _Net()
    red = GradReducer(list(net.parameters()))
    red.zero()
    net(torch.randn(3, 8)).sum().backward()
    assert red.finish() == 1.0
    assert float(red.flat_grad.abs().sum()) > 0
    assert net.a.weight.data_ptr() >= red.flat_param.data_ptr()


def _agree_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from routeformer_amd.engine import agree_unused
    grp = dist.new_group(backend="gloo")
    # step 1: only rank 0 dropped the gaze branch -> nobody may skip; step 2: both dropped it -> both skip
    q.put((rank, agree_unused(rank == 0, grp), agree_unused(True, grp), agree_unused(False, grp)))
    dist.destroy_process_group()


def test_unused_parameter_agreement_two_ranks():
    """Optimizer slots of the gaze branch are skipped only when EVERY rank dropped the branch this step (the
    reference's DDP(find_unused_parameters=True) + AdamW(grad None -> skip) behaviour)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_agree_worker, args=(r, 2, 29731, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert got == [(0, False, True, False), (1, False, True, False)]


def _worker_modes(rank, world, port, mode, out):
    """Direct reduce-scatter modes: after finish() this rank's chunk of every region holds the rank-summed gradient;
    a (reference, pure-torch) sharded AdamW on the chunks + gather_params leaves identical parameters everywhere."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from routeformer_amd.engine import GradReducer
    torch.manual_seed(0)
    net = _Net()
    lead = {id(p) for n, p in net.named_parameters() if n.startswith("c.")}
    red = GradReducer(list(net.parameters()), bucket_mb=0.0001, mode=mode, lead=lead)
    red.hooks_enabled = False
    red.broadcast_parameters(0)
    names = {id(p): n for n, p in net.named_parameters()}
    assert red.sharded and len(red.regions) == 2 and all((hi - lo) % (64 * world) == 0 for lo, hi in red.regions)
    m = torch.zeros_like(red.flat_param)
    v = torch.zeros_like(red.flat_param)
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(5, 8, generator=g)
    chunks_seen = []
    for step in range(2):
        red.zero()
        net(x).square().sum().backward()
        full = red.flat_grad.clone()                       # this rank's own gradients, before the exchange
        early = red.launch_complete_prefix(names, "c.")   # the leading region goes first (stage boundary)
        scale = red.finish()
        mine = red.local_chunks()
        chunks_seen.append([red.flat_grad[a:b].clone() * scale for a, b in mine])
        for a, b in mine:                                  # reference sharded AdamW (lr 1e-2, no clip / decay)
            gr = red.flat_grad[a:b] * scale
            m[a:b].mul_(0.9).add_(gr, alpha=0.1)
            v[a:b].mul_(0.999).addcmul_(gr, gr, value=0.001)
            mh, vh = m[a:b] / (1 - 0.9 ** (step + 1)), v[a:b] / (1 - 0.999 ** (step + 1))
            red.flat_param[a:b].add_(-1e-2 * mh / (vh.sqrt() + 1e-8))
        red.gather_params()
    out[rank] = (red.flat_param.clone(), chunks_seen, mine, full, early, list(red.regions),
                 torch.cat([p.detach().reshape(-1) for p in net.parameters()]))
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["direct", "direct_bf16"])
def test_direct_reduce_scatter_modes(mode):
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_modes, args=(world, port, mode, out), nprocs=world, join=True)
        r0, r1 = out[0], out[1]
    assert torch.equal(r0[0], r1[0]), "replicas must be bit-identical after the parameter all-gather"
    assert torch.equal(r0[6], r1[6]), "module parameters (views of the flat buffer) must agree too"
    assert r0[4] == 1 and r0[5] == r1[5], "one leading region launched early; same region table on every rank"
    # last step: rank r's chunk of the mean gradient == mean over ranks of the full local gradients on that slice
    tol = 1e-6 if mode == "direct" else 1e-2
    for me, other in ((r0, r1), (r1, r0)):
        for (a, b), got in zip(me[2], me[1][-1]):
            want = (me[3][a:b] + other[3][a:b]) / 2
            assert torch.allclose(got, want, atol=tol * max(1.0, float(want.abs().max())), rtol=0), mode
    # the chunks tile every region exactly once
    cover = sorted([c for c in r0[2]] + [c for c in r1[2]])
    assert cover[0][0] == 0 and all(cover[i][1] == cover[i + 1][0] for i in range(len(cover) - 1))


def _worker_coalesce(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from routeformer_amd.engine import GradReducer
    torch.manual_seed(0)
    net = _Net()
    res = {}
    for coalesce in (True, False):
        red = GradReducer(list(net.parameters()), bucket_mb=0.0001, coalesce=coalesce)
        red.hooks_enabled = False
        calls = []
        real = dist.all_reduce
        dist.all_reduce = lambda t, *a, **k: (calls.append(t.numel()), real(t, *a, **k))[1]
        try:
            red.zero()
            net(torch.randn(5, 8, generator=torch.Generator().manual_seed(100 + rank))).square().sum().backward()
            scale = red.finish()
        finally:
            dist.all_reduce = real
        res[coalesce] = (len(calls), len(red.buckets), red.flat_grad.clone() * scale)
    out[rank] = res
    dist.destroy_process_group()


def test_per_bucket_launches_equal_coalesced_runs():
    """RF_DP_COALESCE=0: one collective per bucket instead of one per run of neighbouring buckets -- same result."""
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_coalesce, args=(world, port, out), nprocs=world, join=True)
        r0 = out[0]
    assert r0[True][0] == 1 and r0[False][0] == r0[False][1] >= 3
    assert torch.equal(r0[True][2], r0[False][2])
