"""Where does attn_fwd_kernel spend its time?  Builds a private copy of the library with -DRF_ATTN_TIMING
(shader-clock stamps after every phase), runs one frame-encoder-shaped launch and prints the mean phase
durations over the workgroups.  GPU box only:  python tools/attn_phase_probe.py [B H L E mode]"""
import ctypes, os, subprocess, sys
import numpy as np
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "routeformer_amd", "csrc")
out = os.path.join(root, "gpurun_out", "librf_timing.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-DRF_ATTN_TIMING",
                       f"-I{root}/include", f"-I{src}", os.path.join(src, "attention.hip"), os.path.join(src, "vision.hip"), "-o", out])
os.environ["RF_HIP_LIB"] = out
B, H, L, E, mode = (int(v) for v in sys.argv[1:6]) if len(sys.argv) > 5 else (192, 8, 65, 16, 1)
lib = ctypes.CDLL(out)
hip = ctypes.CDLL("libamdhip64.so")
dev = "cuda"
HE = H * E
qkv = torch.randn(B * L, 3 * HE, device=dev)
ctx = torch.empty(B * L, HE, device=dev)
import math
sk = min(L, 5 * math.ceil(math.log(L))); nt = min(L, 5 * math.ceil(math.log(L)))
idx = torch.randint(L, (L, sk), device=dev, dtype=torch.int32)
top = torch.empty(B * H * nt, device=dev, dtype=torch.int32)
P = ctypes.c_void_p
def call():
    return lib.rf_attn_fwd(P(qkv.data_ptr()), P(qkv.data_ptr() + 4 * HE), P(qkv.data_ptr() + 8 * HE), ctypes.c_int64(3 * HE),
                           ctypes.c_int64(3 * HE), ctypes.c_int64(3 * HE), P(ctx.data_ptr()), 0, P(idx.data_ptr()), 0, ctypes.c_int64(0), P(top.data_ptr()), 0,
                           B, H, L, L, E, sk, nt, mode, ctypes.c_float(1 / math.sqrt(E)), P(torch.cuda.current_stream().cuda_stream))
for _ in range(3):
    assert call() == 0
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); [call() for _ in range(20)]; e.record(); torch.cuda.synchronize()
print(f"launch: {s.elapsed_time(e) / 20 * 1e3:.1f} us  (B*H = {B * H} workgroups, L = {L}, E = {E}, mode {mode}, sample_k {sk}, n_top {nt})")
sym = ctypes.c_void_p(); size = ctypes.c_size_t()
# the timing array is a device global of the private library
rc = hip.hipGetSymbolAddress  # not usable for a dlopen'ed module symbol by name -> use hipModule API through the runtime's symbol lookup
buf = torch.zeros(16 * 4096, device=dev, dtype=torch.int64)
getaddr = lib.rf_attn_timing_address
getaddr.restype = ctypes.c_void_p
addr = getaddr()
hip.hipMemcpy(P(buf.data_ptr()), P(addr), ctypes.c_size_t(8 * 16 * 4096), 3)
t = buf.cpu().numpy().reshape(4096, 16)[: min(B * H, 4096), :9].astype(np.float64)
names = ["load q,k,v", "sampled scores", "sparsity measure", "select top", "lazy rows", "scores (mfma)", "softmax", "P V (mfma) + store"]
d = np.diff(t, axis=1)
print("phase                 mean cycles   (shader clock)")
for n, v in zip(names, d.mean(0)):
    print(f"{n:22s} {v:10.0f}")
print(f"{'total per workgroup':22s} {(t[:, 8] - t[:, 0]).mean():10.0f};  first start -> last end: {t[:, 8].max() - t[:, 0].min():.0f} cycles")

# ---- backward ----
dctx = torch.randn(B * L, HE, device=dev)
dqkv = torch.empty(B * L, 3 * HE, device=dev)
def call_bwd():
    return lib.rf_attn_bwd(P(qkv.data_ptr()), P(qkv.data_ptr() + 4 * HE), P(qkv.data_ptr() + 8 * HE), ctypes.c_int64(3 * HE),
                           ctypes.c_int64(3 * HE), ctypes.c_int64(3 * HE), P(dctx.data_ptr()), 0, P(top.data_ptr()),
                           P(dqkv.data_ptr()), P(dqkv.data_ptr() + 4 * HE), P(dqkv.data_ptr() + 8 * HE), ctypes.c_int64(3 * HE),
                           ctypes.c_int64(3 * HE), ctypes.c_int64(3 * HE), B, H, L, L, E, nt, mode, ctypes.c_float(1 / math.sqrt(E)),
                           P(torch.cuda.current_stream().cuda_stream))
for _ in range(3):
    assert call_bwd() == 0
torch.cuda.synchronize()
s.record(); [call_bwd() for _ in range(20)]; e.record(); torch.cuda.synchronize()
print(f"backward launch: {s.elapsed_time(e) / 20 * 1e3:.1f} us")
hip.hipMemcpy(P(buf.data_ptr()), P(addr), ctypes.c_size_t(8 * 16 * 4096), 3)
t = buf.cpu().numpy().reshape(4096, 16)[: min(B * H, 4096), 9:15].astype(np.float64)
for n_, v in zip(["load K,V + gather Qsel,dCsel", "P and dP (mfma)", "softmax + dS", "dQ (mfma) + zero fill + colsum", "dK, dV (mfma) + stores"],
                 np.diff(t, axis=1).mean(0)):
    print(f"{n_:34s} {v:10.0f}")
print(f"{'total per workgroup':34s} {(t[:, 5] - t[:, 0]).mean():10.0f}")
