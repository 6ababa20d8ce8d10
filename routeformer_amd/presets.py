"""Hyper-parameter presets and the named workload cases.

Plain ``dict`` kwargs so that the SAME preset instantiates either this package's config classes or
(only inside ``tests/golden/make_golden.py``, in the build container) the reference's.

* ``GPS_PAPER`` / ``RF_PAPER``  = ``experiments/full_comparison.py:159-178`` / ``:264-282`` + ``:226``
  (dropouts forced to 0 for parity runs).
* ``GPS_DEFAULT`` / ``RF_DEFAULT`` = dataclass defaults (``models/config.py:14-81``,
  ``gps_backbone/config.py:11-26``) with 64-d embeddings when gaze is on (required).
"""
from __future__ import annotations

import copy

GPS_PAPER = dict(embed="timeF", freq="m", moving_avg=25, factor=4, distil=True, dropout=0.0,
                 activation="relu", individual=False, d_model=832, n_heads=8, e_layers=6,
                 d_layers=1, d_ff=832 * 4)
GPS_DEFAULT = dict(dropout=0.0)
GPS_TINY = dict(d_model=64, n_heads=4, e_layers=3, d_layers=1, d_ff=128, factor=2, dropout=0.0)

_RF_COMMON = dict(view_dropout=0.0, gaze_dropout=0.0, feature_dropout=0.0, motion_noise=0.0,
                  epsilon=1.0, visual_epsilon=0.3)
RF_PAPER = dict(_RF_COMMON, discount_factor={0: 0.97, 100: 0.98, 200: 0.99}, decoder_mode="smart",
                video_fps=1, gaze_fps=1, dense_prediction=True, dense_loss_ratio=0.5,
                image_embedding_size=64, encoder_hidden_size=64, encoder_heads=8,
                encoder_layers=8, encoder_d_ff=256, cross_modal_decoder_heads=8,
                cross_modal_decoder_layers=2, lr=1e-5, wd=1e-4, optimizer="AdamW")
RF_DEFAULT = dict(_RF_COMMON, image_embedding_size=64, encoder_hidden_size=64)

# name -> workload description.  "streams": video keys present in the batch.
CASES = {
    # BASELINE.json configs[0]: GPS-only plumbing case.
    "c1_default": dict(B=4, T=10, P=15, H=0, W=0, streams=(), gaze=False, gps=GPS_DEFAULT,
                       rf=dict(RF_DEFAULT, with_video=False)),
    "c1_paper": dict(B=4, T=10, P=15, H=0, W=0, streams=(), gaze=False, gps=GPS_PAPER,
                     rf=dict(_RF_COMMON, with_video=False, decoder_mode="smart",
                             discount_factor={0: 0.97})),
    # configs[1] at reduced size (parity) and at full size (bench / one golden forward).
    "c2_small": dict(B=2, T=20, P=10, H=64, W=64,
                     streams=("left_video", "right_video", "front_video"), gaze=True, gps=GPS_TINY,
                     rf=dict(RF_DEFAULT, with_video=True, with_gaze=True, dense_prediction=True,
                             dense_loss_ratio=0.5, decoder_mode="smart", encoder_layers=2,
                             cross_modal_decoder_layers=1)),
    # the PerceiveEncoder-based baseline (experiments/multimodal_transformer): every frame is encoded (T = 8)
    "mmt_small": dict(B=2, T=8, P=6, H=64, W=64,
                      streams=("left_video", "right_video", "front_video"), gaze=True, gps=GPS_TINY,
                      rf=dict(RF_DEFAULT, with_video=True, with_gaze=True, encoder_layers=2)),
    "c2_paper": dict(B=2, T=40, P=30, H=224, W=224,
                     streams=("left_video", "right_video", "front_video"), gaze=True, gps=GPS_PAPER,
                     rf=dict(RF_PAPER, with_video=True, with_gaze=True)),
    # configs[3]: DR(eye)VE-shaped, one video stream, no gaze, rotated motion.
    "c4_small": dict(B=2, T=20, P=10, H=64, W=64, streams=("left_video",), gaze=False,
                     gps=GPS_TINY,
                     rf=dict(RF_DEFAULT, with_video=True, with_gaze=False, dense_prediction=True,
                             rotate_motion=True, decoder_mode="smart", encoder_d_ff=128)),
    # recursive decoder + normalised, rotated motion (routeformer.py:243-250,286-287,368-371);
    # "recursive" with dense_prediction raises a shape error in the reference, so GPS-only.
    "c1_recursive": dict(B=3, T=10, P=15, H=0, W=0, streams=(), gaze=False, gps=GPS_TINY,
                         rf=dict(RF_DEFAULT, with_video=False, rotate_motion=True,
                                 decoder_mode="recursive", normalize_motion=True,
                                 motion_mean=1.8332362885457094, motion_std=0.9090128501056961)),
    # motion_noise > 0 (routeformer.py:281-282): torch.randn_like(gps) is the FIRST host draw of a train-mode forward
    "c1_noise": dict(B=3, T=10, P=15, H=0, W=0, streams=(), gaze=False, gps=GPS_TINY,
                     rf=dict(RF_DEFAULT, with_video=False, motion_noise=0.25, decoder_mode="smart")),
    # configs[4] at reduced size: long horizon (fusion length 320), GELU Informer.
    "c5_small": dict(B=1, T=80, P=25, H=96, W=96,
                     streams=("left_video", "right_video", "front_video"), gaze=True, gps=GPS_TINY,
                     rf=dict(RF_DEFAULT, with_video=True, with_gaze=True, dense_prediction=True,
                             decoder_mode="smart", cross_modal_decoder_layers=2)),
    # eval-time autoregressive loop (routeformer.py:164-197), vanilla decoder, no dense head.
    "ar_small": dict(B=2, T=20, P=10, H=64, W=64, streams=("left_video", "right_video"),
                     gaze=False, gps=GPS_TINY,
                     rf=dict(RF_DEFAULT, with_video=True, with_gaze=False, dense_prediction=True,
                             autoregressive=True, autoregressive_step_size=5,
                             decoder_mode="vanilla")),
}
# Full-size workloads for bench.py (BASELINE.json configs[1..4]); never run on the CPU oracle whole.
BENCH_CASES = {
    "C2": dict(CASES["c2_paper"], B=8),
    "C4": dict(B=16, T=40, P=30, H=224, W=224, streams=("left_video",), gaze=False, gps=GPS_PAPER,
               rf=dict(RF_PAPER, with_video=True, with_gaze=False, rotate_motion=True)),
    "C5": dict(B=4, T=80, P=25, H=448, W=448,
               streams=("left_video", "right_video", "front_video"), gaze=True, gps=GPS_PAPER,
               rf=dict(RF_PAPER, with_video=True, with_gaze=True)),
}


def case(name: str) -> dict:
    src = CASES if name in CASES else BENCH_CASES
    return copy.deepcopy(src[name])


def build_configs(c: dict, GPSBackboneConfig, RouteformerConfig, VideoBackboneConfig=None):
    """Instantiate (gps_cfg, rf_cfg) for a case with whichever config classes are passed in."""
    gps_cfg = GPSBackboneConfig(seq_len=c["T"], label_len=c["T"], pred_len=c["P"], **c["gps"])
    rf_kw = dict(c["rf"])
    if rf_kw.get("with_video", False) and VideoBackboneConfig is not None:
        rf_kw["video_backbone_config"] = VideoBackboneConfig(torchcache_enabled=False)
    return gps_cfg, RouteformerConfig(gps_backbone_config=gps_cfg, **rf_kw)
