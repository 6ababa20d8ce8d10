import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
WSEED, DSEED, RSEED = 7, 11, 1234  # must match tests/golden/make_golden.py


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """RF_SEGV_BT=1: native backtrace on a host SIGSEGV (tools/probes/segv_bt.c), installed after pytest's faulthandler."""
    if os.environ.get("RF_SEGV_BT") == "1":
        import ctypes
        lib = ctypes.CDLL(os.path.join(ROOT, "tools", "probes", "libsegv_bt.so"))
        path = os.environ.get("RF_SEGV_FILE")
        fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_APPEND, 0o644) if path else -1
        assert lib.rf_segv_bt_install(fd) == 0
    if os.environ.get("RF_GC") == "off":  # crash bisect: is the cyclic collector's timing part of it?
        import gc
        gc.disable()


def pytest_sessionfinish(session, exitstatus):
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        print(f"\n[gpu memory] peak reserved {torch.cuda.max_memory_reserved() / 2**30:.1f} GiB, reserved at exit "
              f"{torch.cuda.memory_reserved() / 2**30:.1f} GiB")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def draws(gold, prefix):
    """Recorded torch.randint tensors `<prefix>draw000..` as a list of int64 tensors."""
    out, i = [], 0
    while f"{prefix}draw{i:03d}" in gold.files:
        out.append(torch.from_numpy(gold[f"{prefix}draw{i:03d}"].astype(np.int64)))
        i += 1
    return out


def t(a):
    return torch.from_numpy(np.asarray(a))


def build_product_model(case_name, device="cpu", rf=None, gps=None):
    """The product model with the fixture's synthetic weights; returns (model, cfg, state_dict, case).
    ``rf`` / ``gps``: overrides of the case's RouteformerConfig / GPSBackboneConfig keywords (e.g. the dropouts)."""
    from routeformer_amd import presets, synthetic
    from routeformer_amd.models import Routeformer, RouteformerConfig
    from routeformer_amd.models.gps_backbone import GPSBackboneConfig, Informer
    from routeformer_amd.models.video_backbone import HRNet16Backbone, VideoBackboneConfig

    c = presets.case(case_name)
    if rf:
        c["rf"] = dict(c["rf"], **rf)
    if gps:
        c["gps"] = dict(c["gps"], **gps)
    _, cfg = presets.build_configs(c, GPSBackboneConfig, RouteformerConfig, VideoBackboneConfig)
    model = Routeformer(cfg, gps_backbone=Informer, video_backbone=HRNet16Backbone if cfg.with_video else None)
    sd = synthetic.synth_state_dict(model.state_dict(), WSEED)
    model.load_state_dict(sd)
    return model.to(device), cfg, sd, c


def case_item(c):
    from routeformer_amd import synthetic
    return synthetic.synth_item(c["B"], c["T"], c["P"], DSEED, c["H"], c["W"], streams=c["streams"], gaze=c["gaze"])


def rel_err(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / max(1.0, float(b.abs().max())))


def fro_err(a, b):
    """Relative Frobenius error: robust to a handful of ReLU-mask / top-u flips in bf16 mode."""
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / max(1e-12, float(b.norm())))


def masks(gold, prefix):
    """Recorded nn.Dropout keep-masks `<prefix>maskbits` / `<prefix>maskshapes` as a list of bool tensors (call order)."""
    shapes = gold[prefix + "maskshapes"]
    bits = np.unpackbits(gold[prefix + "maskbits"])
    out, off = [], 0
    for row in shapes:
        shape = tuple(int(d) for d in row if d > 0)
        n = int(np.prod(shape))
        out.append(torch.from_numpy(bits[off:off + n].astype(bool).reshape(shape)))
        off += n
    return out
