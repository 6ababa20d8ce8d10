"""Per-phase shader-clock breakdown of the fused encoder-stack kernel (private -DRF_SL_TIMING build).  GPU box only:
    python tools/seqlayer_probe.py [B] [L] [F] [layers] [save]"""
import ctypes, math, os, subprocess, sys
import numpy as np
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
src = os.path.join(root, "routeformer_amd", "csrc")
out = os.path.join(root, "gpurun_out", "librf_sltiming.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-DRF_SL_TIMING", *os.environ.get("RF_PROBE_DEFS", "").split(),
                       f"-I{root}/include", f"-I{src}", os.path.join(src, "seqlayer.hip"), os.path.join(src, "seqlayer_bwd.hip"), os.path.join(src, "vision.hip"), "-o", out])
from routeformer_amd import _hip
_hip.LIB_PATH = out  # the whole binding layer on the private build (vision.hip carries the library's globals)
_hip._lib = None
handle = ctypes.CDLL(out)
B, L, F_, n, save = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 192), (2, 65), (3, 256), (4, 8), (5, 1)))
dev = "cuda"
P = ctypes.c_void_p
g = torch.Generator().manual_seed(0)
D = 128
stride = int(handle.rf_seqlayer_pack_bytes(F_)) if hasattr(handle, "rf_seqlayer_pack_bytes") else 0
handle.rf_seqlayer_pack_bytes.restype = ctypes.c_int64
stride = handle.rf_seqlayer_pack_bytes(F_)
wpack = torch.zeros(n * stride, dtype=torch.uint8, device=dev)
ents = []
keep = []
for li in range(n):
    o = wpack.data_ptr() + li * stride
    o_wo, o_w1 = 24 * 4096, 24 * 4096 + 8 * 4096
    o_w2 = o_w1 + (F_ // 16) * 4096
    o_vec = o_w2 + 8 * (F_ // 32) * 1024
    for (N, K_, off) in ((384, 128, 0), (128, 128, o_wo), (F_, 128, o_w1), (128, F_, o_w2)):
        w = (torch.randn(N, K_, generator=g) / math.sqrt(K_)).to(dev); keep.append(w)
        ents.append((w, o + off, K_, N, K_))
    v = (0.1 * torch.randn(1152 + F_, generator=g)).to(dev); v[640 + F_:768 + F_] += 1; v[896 + F_:1024 + F_] += 1; keep.append(v)
    ents.append((v, o + o_vec, 0, 1152 + F_, 0))
arr = (_hip.SeqPackEntry * len(ents))()
for e, (w, off, ldw, N, K_) in zip(arr, ents):
    e.w, e.out, e.ldw, e.N, e.K, e.transpose, e.pad = w.data_ptr(), off, ldw, N, K_, 0, 0
st0 = torch.cuda.current_stream().cuda_stream
assert handle.rf_seqlayer_pack(arr, len(ents), P(st0)) == 0
sample_k = min(5 * math.ceil(math.log(L)), L); n_top = sample_k
M = B * L
x = torch.randn(M, D, device=dev)
idx = [torch.randint(L, (1, L, sample_k), generator=g).to(torch.int32).to(dev) for _ in range(n)]
f32 = dict(device=dev, dtype=torch.float32)
sv = {"y": torch.empty(n if save else 1, M, 128, **f32), "top": torch.empty(n, B, 8, n_top, device=dev, dtype=torch.int32)}
if save:
    for name, width in (("qkv", 384), ("ctx", 128), ("xhat1", 128), ("x1", 128), ("xhat2", 128), ("h", F_), ("z", F_)):
        sv[name] = torch.empty(n, M, width, **f32)
    sv["rstd1"] = torch.empty(n, M, **f32); sv["rstd2"] = torch.empty(n, M, **f32)
st = _hip.SeqStack()
st.wpack, st.wpack_stride, st.n_layers, st.idx_stride = wpack.data_ptr(), stride, n, L * sample_k
for i, t in enumerate(idx): st.idx[i] = t.data_ptr()
for name in ("top", "y", "qkv", "ctx", "xhat1", "rstd1", "x1", "z", "h", "xhat2", "rstd2"):
    setattr(st, name, sv[name].data_ptr() if name in sv else None)
def call():
    return handle.rf_seqlayer_fwd(ctypes.byref(st), P(x.data_ptr()), B, L, 128, 8, F_, 2, sample_k, n_top, B, 0, save,
                                  ctypes.c_float(0.25), ctypes.c_float(1e-5), ctypes.c_float(0.0), None, 0, P(st0))
for _ in range(3): assert call() == 0, handle.rf_last_error()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); [call() for _ in range(10)]; e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) / 10 * 1e3
flops = n * B * (2.0 * L * 128 * (384 + 128 + 2 * F_) + 8 * (2.0 * L * L * 16 + 4.0 * n_top * L * 16))
print(f"launch: {us:.1f} us for B={B} L={L} F={F_} layers={n} save={save}  ({us / n:.1f} us per layer, {flops / us / 1e6:.2f} TFLOP/s)")
handle.rf_sl_timing_address.restype = ctypes.c_void_p
hip = ctypes.CDLL("libamdhip64.so")
nb = min(B, 512)
buf = torch.zeros(512 * 8 * 16, device=dev, dtype=torch.int64)
hip.hipMemcpy(P(buf.data_ptr()), P(handle.rf_sl_timing_address()), ctypes.c_size_t(8 * 512 * 8 * 16), 3)
t = buf.cpu().numpy().reshape(512, 8, 16)[:nb].astype(np.float64)
names = ["wait barrier (x image)", "QKV projection (+ saves)", "wait barrier", "attention: score tiles + measure", "attention: rank + select",
         "attention: lazy rows", "attention: softmax + PV", "wait barrier (ctx)", "ctx save + out-proj + LN1 (+ saves)", "wait barrier",
         "conv1 + act (+ saves)", "wait barrier", "conv2 + LN2 (+ saves)"]
d = np.diff(t[:, :, :14], axis=2)
for i, nm in enumerate(names):
    print(f"{nm:42s} mean {d[:, :, i].mean():9.0f}   max-wave {d[:, :, i].max(axis=1).mean():9.0f} cycles")
print(f"{'layer 0 total':42s} {(t[:, :, 13] - t[:, :, 0]).mean():9.0f} cycles")
print(f"exact tie-breaking passes taken (all layers, all launches so far): {int(t[:, :, 15].sum())} of {nb * 8 * n * 13} head-layers")

if save:
    # ---- the fused backward on these saves (csrc/seqlayer_bwd.hip) ----
    handle.rf_seqlayer_bwd_pack_bytes.restype = ctypes.c_int64
    sb = handle.rf_seqlayer_bwd_pack_bytes(F_)
    wb = torch.zeros(n * sb, dtype=torch.uint8, device=dev)
    ents = []
    for li in range(n):
        o = wb.data_ptr() + li * sb
        o_w1t = (F_ // 16) * 4096
        o_wot = o_w1t + 8 * (F_ // 32) * 1024
        o_wqkvt = o_wot + 32 * 1024
        o_vec = o_wqkvt + 96 * 1024
        wqkv, wo, w1, w2, v = keep[5 * li:5 * li + 5]
        for w, off, N, K_ in ((w2, o, F_, 128), (w1, o + o_w1t, 128, F_), (wo, o + o_wot, 128, 128), (wqkv, o + o_wqkvt, 128, 384)):
            ents.append((w.data_ptr(), off, w.stride(0), N, K_, 1))
        ents.append((v[640 + F_:].data_ptr(), o + o_vec, 0, 128, 0, 0))
        ents.append((v[896 + F_:].data_ptr(), o + o_vec + 512, 0, 128, 0, 0))
    for s0 in range(0, len(ents), 64):
        chunk = ents[s0:s0 + 64]
        arr = (_hip.SeqPackEntry * len(chunk))()
        for e_, (w, off, ldw, N, K_, tr) in zip(arr, chunk):
            e_.w, e_.out, e_.ldw, e_.N, e_.K, e_.transpose, e_.pad = w, off, ldw, N, K_, tr, 0
        assert handle.rf_seqlayer_pack(arr, len(chunk), P(st0)) == 0
    bs = _hip.SeqStackBwd()
    bs.wpack, bs.wpack_stride, bs.n_layers = wb.data_ptr(), sb, n
    for name in ("qkv", "xhat1", "rstd1", "xhat2", "rstd2", "top"):
        setattr(bs, name, sv[name].data_ptr())
    bs.zsrc = sv["z"].data_ptr()
    outs = {"dpre2": torch.empty(n, M, 128, **f32), "dz": torch.empty(n, M, F_, **f32), "dpre1": torch.empty(n, M, 128, **f32),
            "dqkv": torch.empty(n, M, 384, **f32)}
    for name, t_ in outs.items():
        setattr(bs, name, t_.data_ptr())
    lnacc = torch.zeros(n, 4, 128, **f32)
    for i in range(n):
        bs.dgamma1[i], bs.dbeta1[i], bs.dgamma2[i], bs.dbeta2[i] = (lnacc[i, j].data_ptr() for j in range(4))
    dy = torch.randn(M, D, device=dev)
    dx = torch.empty(M, D, device=dev)

    def bcall():
        return handle.rf_seqlayer_bwd(ctypes.byref(bs), P(dy.data_ptr()), P(dx.data_ptr()), B, L, 128, 8, F_, 2, n_top,
                                      ctypes.c_float(0.25), P(st0))
    for _ in range(3): assert bcall() == 0, handle.rf_last_error()
    torch.cuda.synchronize()
    s.record(); [bcall() for _ in range(10)]; e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 10 * 1e3
    flops = n * B * (4.0 * L * 128 * (384 + 128 + 2 * F_) / 2 + 8 * 10.0 * n_top * L * 16)
    print(f"\nbackward launch: {us:.1f} us ({us / n:.1f} us per layer, {flops / us / 1e6:.2f} TFLOP/s); dx finite: {bool(torch.isfinite(dx).all())}")
    handle.rf_slb_timing_address.restype = ctypes.c_void_p
    hip.hipMemcpy(P(buf.data_ptr()), P(handle.rf_slb_timing_address()), ctypes.c_size_t(8 * 512 * 8 * 16), 3)
    t = buf.cpu().numpy().reshape(512, 8, 16)[:nb].astype(np.float64)
    names = ["norm2 backward (+ image)", "wait barrier", "dpre2 save + conv2^T + act' (dz image)", "wait barrier",
             "dz save + conv1^T + norm1 backward", "wait barrier", "dpre1 save + out-proj^T + dC^T + selection", "P (scores + softmax)",
             "dP, dS", "wait barrier", "dV", "dK", "dQ", "wait barrier", "dqkv save + projection^T"]
    d = np.diff(t, axis=2)
    for i, nm in enumerate(names):
        print(f"{nm:42s} mean {d[:, :, i].mean():9.0f}   max-wave {d[:, :, i].max(axis=1).mean():9.0f} cycles")
    print(f"{'last layer total':42s} {(t[:, :, 15] - t[:, :, 0]).mean():9.0f} cycles")
