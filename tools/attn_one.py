"""One attention problem shape, forward + backward, a few launches (for rocprofv3 --pmc runs)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from routeformer_amd import kernels as K
B, H, L, E, mode = (int(v) for v in sys.argv[1:6]) if len(sys.argv) > 5 else (192, 8, 65, 16, 1)
HE = H * E
a = torch.randn(B * L, 3 * HE, device="cuda", requires_grad=True)
sk, nt = K.prob_sizes(L, L, 5)
idx = torch.randint(L, (L, sk), device="cuda", dtype=torch.int32)
for _ in range(5):
    o = K.attention(a, a, (0, HE, 2 * HE), (B, H, L, L, E), mode, index_sample=idx, n_top=nt, out_layout=0)
    o.backward(torch.ones_like(o))
torch.cuda.synchronize()
