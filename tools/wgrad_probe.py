"""Shader-clock phases of one 32-row step of the weight-gradient kernel (private -DRF_WT_TIMING build).  GPU box only:
    python tools/wgrad_probe.py [M] [N] [K]"""
import ctypes, os, subprocess, sys
import numpy as np
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
src = os.path.join(root, "routeformer_amd", "csrc")
out = os.path.join(root, "gpurun_out", "librf_wttiming.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-DRF_WT_TIMING",
                       f"-I{root}/include", f"-I{src}", os.path.join(src, "wgrad_tr.hip"), os.path.join(src, "vision.hip"), "-o", out])
from routeformer_amd import _hip
h = ctypes.CDLL(out)
M, N, K = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 560), (2, 3328), (3, 832)))
dev = "cuda"
dy, x, dw = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev), torch.zeros(N, K, device=dev)
arr = (_hip.WgradEntry * 1)()
e = arr[0]
e.dy, e.x, e.dw, e.db = dy.data_ptr(), x.data_ptr(), dw.data_ptr(), None
e.M, e.N, e.K, e.ld_dy, e.ld_x, e.splits, e.kchunk, e.exclusive = M, N, K, N, K, 1, 0, 1
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    assert h.rf_wgrad_tr(arr, 1, st) == 0
torch.cuda.synchronize()
s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); [h.rf_wgrad_tr(arr, 1, st) for _ in range(10)]; t.record(); torch.cuda.synchronize()
print(f"launch: {s.elapsed_time(t) / 10 * 1e3:.1f} us for M={M} N={N} K={K}; max err {float((dw - dy.bfloat16().float().T @ x.bfloat16().float()).abs().max()):.2e}")
h.rf_wt_timing_address.restype = ctypes.c_void_p
buf = torch.zeros(64 * 8 * 8, device=dev, dtype=torch.int64)
ctypes.CDLL("libamdhip64.so").hipMemcpy(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(h.rf_wt_timing_address()),
                                        ctypes.c_size_t(8 * 64 * 8 * 8), 3)
tt = buf.cpu().numpy().reshape(64, 8, 8).astype(np.float64)
d = np.diff(tt[:, :, :5], axis=2)
for i, nm in enumerate(["issue the loads of step s + 3", "transposed reads + 16 MFMAs", "wait for step s + 1's loads, convert, LDS stores", "barrier"]):
    print(f"{nm:52s} mean {d[:, :, i].mean():8.0f}   max-wave {d[:, :, i].max(axis=1).mean():8.0f} cycles")
print(f"{'one 32-row step':52s} {(tt[:, :, 4] - tt[:, :, 0]).mean():8.0f} cycles")
