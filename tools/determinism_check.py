#!/usr/bin/env python3
"""Runs the segmentation plan several times on one image and reports whether the logits repeat bit for bit,
with and without the fused depthwise+pointwise kernel (race hunting)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from vision_semantic_segmentation_amd.network import SegNet, random_state_dict  # noqa: E402

h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1440, 1920)
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 8
state = random_state_dict(0)
img = torch.from_numpy(np.random.default_rng(9).integers(0, 256, size=(h, w, 3), dtype=np.uint8)).cuda()
for fuse in (True, False):
    net = SegNet(state, h, w, precision="bf16", device="cuda:0", fuse_dwpw=fuse)
    net.forward(img)
    torch.cuda.synchronize()
    ref = net.logits.clone()
    bad = 0
    for i in range(REPS):
        net.forward(img)
        torch.cuda.synchronize()
        d = (net.logits != ref)
        if bool(d.any()):
            bad += 1
            idx = d.nonzero()
            print("  fuse=%s run %d: %d logits differ, rows %d..%d cols %d..%d" % (fuse, i, int(d.sum()), int(idx[:, 0].min()), int(idx[:, 0].max()),
                                                                                 int(idx[:, 1].min()), int(idx[:, 1].max())))
    print("fuse_dwpw=%s eager: %d of %d repeats differ" % (fuse, bad, REPS))
    net.capture_graph()
    bad = 0
    for i in range(REPS):
        net.forward(img)
        torch.cuda.synchronize()
        d = (net.logits != ref)
        if bool(d.any()):
            bad += 1
            idx = d.nonzero()
            print("  fuse=%s graph run %d: %d logits differ, rows %d..%d cols %d..%d" % (fuse, i, int(d.sum()), int(idx[:, 0].min()), int(idx[:, 0].max()),
                                                                                       int(idx[:, 1].min()), int(idx[:, 1].max())))
    print("fuse_dwpw=%s graph: %d of %d repeats differ" % (fuse, bad, REPS))
