#!/usr/bin/env python3
"""Logits error of the mixed mode (and of its fallback rungs) on HEAVY-TAILED weights draws (tests/_full_size.heavy_tailed_state_dict:
checkpoint-like BatchNorm scales, outlier channels) against the torch-CPU fp32 oracle.

    python tools/heavy_tail_sweep.py [seeds, e.g. 0,1,2,3] [H W]

prints, per weights draw and option set, max |dlogit| / max |logit| and the rms figure."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import _full_size as fs  # noqa: E402
from oracle import network_oracle as no  # noqa: E402
from vision_semantic_segmentation_amd.network import SegNet  # noqa: E402

seeds = [int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2,3").split(",")]
h, w = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (320, 416)
SETS = [("f32", "f32", {}), ("mixed (default)", "mixed", {}), ("mixed, three-launch layer1", "mixed", dict(fuse_block=False)),
        ("mixed + layer1 lo planes", "mixed", dict(layer1_lo=True)),
        ("mixed, no FP4 in the grouped conv", "mixed", dict(gconv_mx=False)),
        ("mixed, trunk lo as f16 planes", "mixed", dict(trunk_fp4=False)),
        ("split16 (no FP4 anywhere)", "mixed", dict(mx=False, gconv_mx=False, trunk_fp4=False, layer1_lo=True)),
        ("f16", "f16", {}), ("bf16", "bf16", {})]
if os.environ.get("SWEEP_SETS"):
    SETS = [s for s in SETS if any(k in s[0] for k in os.environ["SWEEP_SETS"].split(","))]
FAMILIES = {"spec": dict(), "mid": dict(gamma=(0.1, 5.0), outlier=(5.0, 10.0)), "mild": dict(gamma=(0.3, 3.0), outlier=(3.0, 6.0), outlier_frac=0.01),
            "noout": dict(gamma=(0.05, 8.0), outlier=(1.0, 1.0)), "damped": dict(residual_gain=0.25), "damped10": dict(residual_gain=0.1)}
fam = os.environ.get("SWEEP_FAMILY", "spec")
for ws in seeds:
    st = fs.heavy_tailed_state_dict(ws, **FAMILIES[fam])
    img = np.random.default_rng(50 + ws).integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    ref = no.forward_logits(st, img)[0]
    scale = float(ref.abs().max())
    for name, prec, opts in SETS:
        net = SegNet(st, h, w, precision=prec, device="cuda:0", **opts)
        net.forward(torch.from_numpy(img).cuda())
        got = net.logits.permute(2, 0, 1).float().cpu()
        err = (got - ref).abs()
        bad = net.nonfinite_counts() if prec != "f32" else {}
        print("heavy-tailed[" + fam + "] weights %d  %dx%d  %-36s max rel %.3e  rms rel %.3e  arg-max %.4f%s" % (
            ws, h, w, name, float(err.max()) / scale, float(err.pow(2).mean().sqrt()) / scale,
            float((got.argmax(0) == ref.argmax(0)).float().mean()), ("  Inf/NaN in " + ", ".join(sorted(bad)[:3])) if bad else ""), flush=True)
        del net
        torch.cuda.empty_cache()
