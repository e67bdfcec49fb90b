#!/usr/bin/env python3
"""Which builder option moves the logits error of the worst weights draw?  (weight seed 2: the bench frame at 1080p and one 480 x 640
frame) x a list of mixed-mode option sets; the oracle runs once per frame.   usage: opt_sweep.py "k=v,k=v;k=v;..." """
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import _full_size as fs  # noqa: E402
from vision_semantic_segmentation_amd.network import SegNet  # noqa: E402

sets = [dict((kv.split("=")[0], bool(int(kv.split("=")[1]))) for kv in part.split(",") if kv) for part in (sys.argv[1] if len(sys.argv) > 1 else "").split(";")]
cases = [(2, 3, 480, 640), (2, "bench", 1080, 1920)]
if len(sys.argv) > 2:
    cases = [c for c in cases if str(c[2]) in sys.argv[2].split(",")]
for wseed, iseed, h, w in cases:
    ref = fs.oracle_logits(wseed, iseed, h, w)
    img = torch.from_numpy(fs.image_for(iseed, h, w)).cuda()
    for opts in sets:
        net = SegNet(fs.state_dict(wseed), h, w, precision="mixed", device="cuda:0", **opts)
        net.forward(img)
        got = net.logits.permute(2, 0, 1).float().cpu()
        err = (got - ref).abs()
        print("weights %d frame %s %dx%d %-40s max rel %.3e  rms rel %.3e" % (wseed, iseed, h, w, opts, float(err.max() / ref.abs().max()),
                                                                             float(err.pow(2).mean().sqrt() / ref.abs().max())), flush=True)
        del net
        torch.cuda.empty_cache()
