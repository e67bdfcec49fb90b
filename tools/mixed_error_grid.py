#!/usr/bin/env python3
"""Logits error (max and rms, relative to max|logit|) of the mixed-precision variants against the torch-CPU fp32 oracle.
    python tools/mixed_error_grid.py [H W]"""
import itertools
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from oracle import network_oracle as no  # noqa: E402
from vision_semantic_segmentation_amd.network import SegNet, random_state_dict  # noqa: E402

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (480, 640)
st = random_state_dict(0)
imgs = [np.random.default_rng(s).integers(0, 256, size=(H, W, 3), dtype=np.uint8) for s in (0, 1)]
refs = [no.forward_logits(st, im)[0] for im in imgs]
for mx, fp4, c2 in itertools.product((False, True), (False, True), (False, True)):
    if fp4 and not mx:
        continue
    net = SegNet(st, H, W, precision="mixed", device="cuda:0", mx=mx, trunk_fp4=fp4, conv2_split=c2)
    out = []
    for im, ref in zip(imgs, refs):
        net.forward(im)
        got = net.logits.permute(2, 0, 1).cpu()
        d = (got - ref) / ref.abs().max()
        out.append("max %.3e rms %.3e" % (float(d.abs().max()), float(d.pow(2).mean().sqrt())))
    print("mx %d trunk_fp4 %d conv2_split %d : %s" % (mx, fp4, c2, " | ".join(out)), flush=True)
    del net
    torch.cuda.empty_cache()
