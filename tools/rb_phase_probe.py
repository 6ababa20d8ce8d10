"""Per-phase shader-clock breakdown of rb_ffn_ln_kernel (private -DRF_RB_TIMING build).  GPU box only:
    python tools/rb_phase_probe.py [M]"""
import ctypes, os, subprocess, sys
import numpy as np
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "routeformer_amd", "csrc")
out = os.path.join(root, "gpurun_out", "librf_rbtiming.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-DRF_RB_TIMING",
                       f"-I{root}/include", f"-I{src}", os.path.join(src, "rowblock.hip"), os.path.join(src, "vision.hip"), "-o", out])
lib = ctypes.CDLL(out); hip = ctypes.CDLL("libamdhip64.so")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 320
dev = "cuda"; P = ctypes.c_void_p
x = torch.randn(M, 128, device=dev); w1 = torch.randn(256, 128, device=dev); b1 = torch.randn(256, device=dev)
w2 = torch.randn(128, 256, device=dev); b2 = torch.randn(128, device=dev); g = torch.ones(128, device=dev); be = torch.zeros(128, device=dev)
h = torch.empty(M, 256, device=dev); z = torch.empty(M, 256, device=dev); y = torch.empty(M, 128, device=dev)
xh = torch.empty(M, 128, device=dev); rs = torch.empty(M, device=dev)
def call():
    return lib.rf_rowblock_ffn_ln(P(x.data_ptr()), P(w1.data_ptr()), P(b1.data_ptr()), P(w2.data_ptr()), P(b2.data_ptr()), P(h.data_ptr()),
                                  P(z.data_ptr()), P(y.data_ptr()), M, 128, 256, 2, P(g.data_ptr()), P(be.data_ptr()), P(xh.data_ptr()),
                                  P(rs.data_ptr()), ctypes.c_float(1e-5), P(torch.cuda.current_stream().cuda_stream))
for _ in range(3): assert call() == 0
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); [call() for _ in range(20)]; e.record(); torch.cuda.synchronize()
print(f"launch: {s.elapsed_time(e) / 20 * 1e3:.1f} us for M = {M} ({(M + 63) // 64} workgroups)")
lib.rf_rb_timing_address.restype = ctypes.c_void_p
buf = torch.zeros(16 * 1024, device=dev, dtype=torch.int64)
hip.hipMemcpy(P(buf.data_ptr()), P(lib.rf_rb_timing_address()), ctypes.c_size_t(8 * 16 * 1024), 3)
n = min((M + 63) // 64, 1024)
t = buf.cpu().numpy().reshape(1024, 16)[:n, :6].astype(np.float64)
names = ["load x, W1 -> LDS", "phase 1: x W1^T, GELU, z/h stores", "W2 -> LDS, epilogue operand loads", "phase 2: h W2^T", "LayerNorm + stores"]
for nm, v in zip(names, np.diff(t, axis=1).mean(0)): print(f"{nm:40s} {v:9.0f} cycles")
print(f"{'total per workgroup':40s} {(t[:, 5] - t[:, 0]).mean():9.0f} cycles")
t2 = buf.cpu().numpy().reshape(1024, 16)[:n, 6:9].astype(np.float64)
print(f"chunk 0 of phase 1 (wave 0): MFMA + tile write {np.mean(t2[:, 1] - t2[:, 0]):.0f} cycles, bias/GELU/stores loop {np.mean(t2[:, 2] - t2[:, 1]):.0f} cycles")

# ---- rb_nn_kernel<128, 256, true>: LayerNorm-backward prologue + dZ = (dPre W2) * gelu'(z) ----
dy = torch.randn(M, 128, device=dev); rstd = torch.rand(M, device=dev) + 0.5; dpre = torch.empty(M, 128, device=dev)
dg = torch.zeros(128, device=dev); db = torch.zeros(128, device=dev); yy = torch.empty(M, 256, device=dev)
def call_nn():
    return lib.rf_rowblock_linear_nn(None, ctypes.c_int64(0), P(dy.data_ptr()), P(xh.data_ptr()), P(rstd.data_ptr()), P(g.data_ptr()),
                                     P(dpre.data_ptr()), P(dg.data_ptr()), P(db.data_ptr()), P(w2.data_ptr()), None, ctypes.c_int64(0),
                                     P(z.data_ptr()), ctypes.c_int64(256), 2, P(yy.data_ptr()), ctypes.c_int64(256), M, 128, 256,
                                     P(torch.cuda.current_stream().cuda_stream))
for _ in range(3): assert call_nn() == 0
torch.cuda.synchronize()
s.record(); [call_nn() for _ in range(20)]; e.record(); torch.cuda.synchronize()
print(f"rb_nn<128,256,ln>: {s.elapsed_time(e) / 20 * 1e3:.1f} us")
hip.hipMemcpy(P(buf.data_ptr()), P(lib.rf_rb_timing_address()), ctypes.c_size_t(8 * 16 * 1024), 3)
t = buf.cpu().numpy().reshape(1024, 16)[:n, :6].astype(np.float64)
for nm, v in zip(["LN backward prologue (+dpre store)", "W -> LDS (transposed)", "column sums (dgamma, dbeta)", "MFMA", "epilogue (gelu', stores)"],
                 np.diff(t, axis=1).mean(0)):
    print(f"{nm:40s} {v:9.0f} cycles")
