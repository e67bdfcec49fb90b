#!/usr/bin/env python3
"""Per-op HIP-event profile of the segmentation plan (default 1080x1920, bf16)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from vision_semantic_segmentation_amd.network import SegNet, random_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--h", type=int, default=1080)
ap.add_argument("--w", type=int, default=1920)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--top", type=int, default=25)
ap.add_argument("--kind", default="")
ap.add_argument("--no-conv2-split", action="store_true")
ap.add_argument("--mixed-opts", default="", help="comma list of key=0|1 for the mixed mode (SegNet.MIXED_OPTS)")
a = ap.parse_args()

opts = {kv.split("=")[0]: bool(int(kv.split("=")[1])) for kv in a.mixed_opts.split(",") if kv}
if a.no_conv2_split:
    opts["conv2_split"] = False
net = SegNet(random_state_dict(0), a.h, a.w, precision=a.precision, device="cuda:0", **opts)
img = torch.from_numpy(np.random.default_rng(1).integers(0, 256, size=(a.h, a.w, 3), dtype=np.uint8)).cuda()
net.forward(img)
torch.cuda.synchronize()
best = None
for _ in range(a.reps):
    prof = net.profile()
    if best is None:
        best = prof
    else:
        for b, p in zip(best, prof):
            b["ms"] = min(b["ms"], p["ms"])
tot = sum(p["ms"] for p in best)
fl = sum(p["flops"] for p in best)
print("total %.3f ms (sum of per-op events), %.1f GFLOP -> %.1f TFLOP/s, %.1f frames/s" % (tot, fl / 1e9, fl / tot / 1e9, 1e3 / tot))
kinds = {}
for p in best:
    k = kinds.setdefault(p["kind"], [0.0, 0.0, 0.0, 0])
    k[0] += p["ms"]; k[1] += p["flops"]; k[2] += p["bytes"]; k[3] += 1
print("%-10s %5s %9s %9s %9s %9s" % ("kind", "n", "ms", "GFLOP", "TFLOP/s", "GB/s"))
for kname, (ms, f, b, n) in sorted(kinds.items(), key=lambda kv: -kv[1][0]):
    print("%-10s %5d %9.3f %9.1f %9.1f %9.1f" % (kname, n, ms, f / 1e9, f / ms / 1e9 if ms else 0, b / ms / 1e6 if ms else 0))
print("--- top ops")
sel = [p for p in best if (not a.kind or p["kind"] in a.kind.split(","))]
for p in (sel if a.kind else sorted(sel, key=lambda p: -p["ms"])[:a.top]):
    print("%-46s %-8s %8.3f ms %8.1f GFLOP %7.1f TF/s %8.1f GB/s" % (p["name"], p["kind"], p["ms"], p["flops"] / 1e9,
                                                                   p["flops"] / p["ms"] / 1e9, p["bytes"] / p["ms"] / 1e6))
# whole-plan timing without per-op events
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    net.forward()
e0.record()
for _ in range(10):
    net.forward()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("plan run: %.3f ms/frame = %.1f frames/s, %.1f TFLOP/s" % (ms, 1e3 / ms, fl / ms / 1e9))
import time
ref = net.labels.clone()
net.capture_graph()
for _ in range(3):
    net.forward()
torch.cuda.synchronize()
t0 = time.perf_counter()
e0.record()
for _ in range(10):
    net.forward()
e1.record()
host = (time.perf_counter() - t0) / 10
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("hipGraph replay: %.3f ms/frame = %.1f frames/s (host %.1f us per forward), labels identical: %s"
      % (ms, 1e3 / ms, host * 1e6, bool(torch.equal(ref, net.labels))))
