import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from routeformer_amd import kernels as K
torch.manual_seed(0)
for (B, L, C) in [(4, 12, 832), (4, 8, 832), (4, 6, 832), (4, 5, 832), (4, 3, 832), (8, 42, 832), (2, 3, 40), (1, 2, 8)]:
    x = torch.randn(B, L, C, device="cuda", requires_grad=True)
    g = torch.randn(C, device="cuda").abs().requires_grad_(); b = torch.randn(C, device="cuda").requires_grad_()
    rm, rv, nbt = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), torch.zeros((), device="cuda", dtype=torch.int64)
    y = K.bn_elu_pool(x, g, b, rm, rv, nbt, training=True)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    gx, gg, gb = x.grad.clone(), g.grad.clone(), b.grad.clone()
    x.grad = g.grad = b.grad = None
    xr = x.permute(0, 2, 1)
    yr = F.max_pool1d(F.elu(F.batch_norm(xr, None, None, g, b, True, 0.1, 1e-5)), 3, 2, 1).permute(0, 2, 1)
    (yr * w).sum().backward()
    e = lambda a, r: float((a - r).abs().max() / r.abs().max())
    print((B, L, C), "y", e(y, yr), "dx", e(gx, x.grad), "dg", e(gg, g.grad), "db", e(gb, b.grad))
