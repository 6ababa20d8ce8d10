"""What does one DEPENDENT kernel boundary cost on this box?  Replays HIP graphs of n trivial kernels (a 1-KB
add_relu: ~1 us of work) chained on one stream, and of the same kernels spread over 4 streams, and reports time per
kernel; then the same for eager launches.  The step's main chain is ~800 dependent launches, so this number times
800 is a floor no kernel tuning can lower.  GPU box only."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from routeformer_amd import _hip, kernels as K
a = torch.randn(256, device="cuda"); b = torch.randn(256, device="cuda"); outs = [torch.empty(256, device="cuda") for _ in range(4)]
def tiny(i=0):
    _hip.check(_hip.lib().rf_add_relu(a.data_ptr(), b.data_ptr(), outs[i].data_ptr(), 0, 256, 1, K._stream()), "add_relu")
def timed_graph(build, reps=20):
    build(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        build()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
for n in (200, 1000):
    t1 = timed_graph(lambda: [tiny() for _ in range(n)])
    streams = [torch.cuda.Stream() for _ in range(4)]
    def fan():
        cur = torch.cuda.current_stream()
        for i, st in enumerate(streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                for _ in range(n // 4): tiny(i)
        for st in streams: cur.wait_stream(st)
    t4 = timed_graph(fan)
    print(f"graph of {n} dependent tiny kernels: {t1 / n * 1e6:.2f} us per kernel;  as 4 independent chains: {t4 / n * 1e6:.2f} us per kernel")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2000): tiny()
torch.cuda.synchronize()
print(f"eager, one stream: {(time.perf_counter() - t0) / 2000 * 1e6:.2f} us per kernel (host launch path included)")
