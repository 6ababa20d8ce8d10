"""Per-phase shader-clock breakdown of conv3x3_kernel (private -DRF_CONV_TIMING build).  GPU box only:
    python tools/conv_phase_probe.py N H W CIN COUT"""
import ctypes, os, subprocess, sys
import numpy as np
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "routeformer_amd", "csrc")
out = os.path.join(root, "gpurun_out", "librf_convtiming.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-DRF_CONV_TIMING",
                       f"-I{root}/include", f"-I{src}", os.path.join(src, "conv3x3.hip"), os.path.join(src, "vision.hip"), "-o", out])
lib = ctypes.CDLL(out); hip = ctypes.CDLL("libamdhip64.so")
N, H, W, CIN, COUT = (int(a) for a in sys.argv[1:6]) if len(sys.argv) > 5 else (336, 4, 4, 128, 128)
dev = "cuda"; P = ctypes.c_void_p
x = torch.randn(N, H, W, CIN, device=dev).bfloat16(); res = torch.randn(N, H, W, COUT, device=dev).bfloat16()
lib.rf_conv3x3_packed_elems.restype = ctypes.c_int64
w32 = torch.randn(COUT, 3, 3, CIN, device=dev) / (3 * CIN ** 0.5); b = torch.randn(COUT, device=dev)
w = torch.empty(lib.rf_conv3x3_packed_elems(CIN, COUT), device=dev, dtype=torch.bfloat16)
assert lib.rf_conv3x3_pack_bf16(P(w32.data_ptr()), P(w.data_ptr()), CIN, COUT, P(torch.cuda.current_stream().cuda_stream)) == 0
y = torch.empty(N, H, W, COUT, device=dev, dtype=torch.bfloat16)
def call():
    return lib.rf_conv3x3_bf16(P(x.data_ptr()), P(w.data_ptr()), P(b.data_ptr()), P(res.data_ptr()), P(y.data_ptr()), 1, N, H, W, CIN, COUT, 1,
                               P(torch.cuda.current_stream().cuda_stream))
for _ in range(3): assert call() == 0
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); [call() for _ in range(20)]; e.record(); torch.cuda.synchronize()
wgs = (N * H * W + 127) // 128
print(f"launch: {s.elapsed_time(e) / 20 * 1e3:.1f} us, {wgs} workgroups  (N={N}, {H}x{W}, {CIN}->{COUT})")
lib.rf_conv_timing_address.restype = ctypes.c_void_p
buf = torch.zeros(8 * 4096, device=dev, dtype=torch.int64)
hip.hipMemcpy(P(buf.data_ptr()), P(lib.rf_conv_timing_address()), ctypes.c_size_t(8 * 8 * 4096), 3)
t = buf.cpu().numpy().reshape(4096, 8)[: min(wgs, 4096), :5].astype(np.float64)
for nm, v in zip(["stage window + first weight loads issued", "barrier (window + weights arrive)", "k loop", "epilogue"], np.diff(t, axis=1).mean(0)):
    print(f"{nm:45s} {v:9.0f} cycles")
print(f"{'total per workgroup':45s} {(t[:, 4] - t[:, 0]).mean():9.0f} cycles; first start -> last end {t[:, 4].max() - t[:, 0].min():.0f}")
