"""Shader-clock breakdown of gemm2_kernel for one shape (private -DRF_GEMM_TIMING build).  GPU box only:
    python tools/gemm_phase_probe.py [M N K]"""
import ctypes, os, subprocess, sys
import numpy as np
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "routeformer_amd", "csrc")
out = os.path.join(root, "gpurun_out", "librf_gemmtiming.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-DRF_GEMM_TIMING",
                       f"-I{root}/include", f"-I{src}", os.path.join(src, "gemm.hip"), os.path.join(src, "vision.hip"), "-o", out])
lib = ctypes.CDLL(out); hip = ctypes.CDLL("libamdhip64.so")
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (320, 3328, 832)
WGRAD = len(sys.argv) > 4 and sys.argv[4] == "wgrad"  # weight-gradient form: dW[M,N] += dY[K,M]^T X[K,N] (fp32 atomics)
P, L = ctypes.c_void_p, ctypes.c_int64
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda"); y = torch.empty(M, N, device="cuda")
dyw = torch.randn(K, M, device="cuda"); xw = torch.randn(K, N, device="cuda"); dw = torch.zeros(M, N, device="cuda")
def call():
    if WGRAD:
        return lib.rf_gemm(P(dyw.data_ptr()), L(1), L(M), P(xw.data_ptr()), L(N), L(1), P(dw.data_ptr()), L(N), M, N, K, None, None, L(0), 0,
                           0, 0, None, L(0), None, L(0), 0, 1, 1, None, 1, None, None, P(torch.cuda.current_stream().cuda_stream))
    return lib.rf_gemm(P(x.data_ptr()), L(K), L(1), P(w.data_ptr()), L(1), L(K), P(y.data_ptr()), L(N), M, N, K, P(b.data_ptr()), None, L(0), 0,
                       0, 0, None, L(0), None, L(0), 0, 1, 1, None, 0, None, None, P(torch.cuda.current_stream().cuda_stream))
for _ in range(3): assert call() == 0
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); [call() for _ in range(20)]; e.record(); torch.cuda.synchronize()
print(f"launch: {s.elapsed_time(e) / 20 * 1e3:.1f} us for {M} x {N} x {K} (bf16 inputs, no split-K{', weight-gradient layouts' if WGRAD else ''})")
lib.rf_gemm_timing_address.restype = ctypes.c_void_p
buf = torch.zeros(16 * 1024, device="cuda", dtype=torch.int64)
hip.hipMemcpy(P(buf.data_ptr()), P(lib.rf_gemm_timing_address()), ctypes.c_size_t(8 * 16 * 1024), 3)
t = buf.cpu().numpy().reshape(1024, 16).astype(np.float64)
t = t[t[:, 8] > 0][:, :9]
d = np.diff(t, axis=1).mean(0)
print(f"workgroups sampled {len(t)}")
print(f"prologue (first tile: load, LDS store, barrier) {d[0]:8.0f} cycles")
print(f"first K trip                                    {t[:, 2].mean() - t[:, 1].mean():8.0f}")
print(f"second K trip: issue next loads {d[2]:6.0f} | MFMA {d[3]:6.0f} | wait + LDS store {d[4]:6.0f} | barrier {d[5]:6.0f}")
print(f"remaining K trips                               {t[:, 7].mean() - t[:, 6].mean():8.0f}")
print(f"epilogue                                        {d[7]:8.0f}")
print(f"total                                           {(t[:, 8] - t[:, 0]).mean():8.0f}")
