#!/usr/bin/env python3
"""Micro-benchmark of the mapping kernels alone (fused project+vote+apply), configs C and E."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from vision_semantic_segmentation_amd import SemanticMapping, get_cfg_defaults, synthetic as syn  # noqa: E402
from vision_semantic_segmentation_amd.camera import camera_setup_1  # noqa: E402
from vision_semantic_segmentation_amd.mapping import PCD_ORIGIN_OFFSET  # noqa: E402
from vision_semantic_segmentation_amd.utils.logger import MyLogger  # noqa: E402


def run(n, res, half, layout, iters=200):
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(1)
    H, W = 1080, 1920
    cam = camera_setup_1().scaled(1.0, 1080 / 1440.0)
    pcd = syn.make_cloud(rng, n, cam.K, cam.R, cam.t, W, H)
    small = torch.from_numpy(syn.make_label_map(rng, 266, 476, tile=8)).to(dev)
    cfg = get_cfg_defaults()
    cfg.MAPPING.BOUNDARY = syn.centred_boundary(PCD_ORIGIN_OFFSET[:2], half)
    cfg.MAPPING.RESOLUTION = res
    sm = SemanticMapping(cfg, device=dev, logger=MyLogger("bench", quiet=True))
    sm.confusion_matrix = syn.log_confusion(5)
    if layout == "f32aos":
        pts = torch.from_numpy(np.ascontiguousarray(pcd.T.astype(np.float32))).to(dev)
    else:
        pts = torch.from_numpy(pcd).to(dev)
    for _ in range(10):
        sm.frame_device(pts, "velodyne", small, None, cam, src_kind="classmap", image_size=(H, W))
    sm.map = np.zeros((sm.map_height, sm.map_width, sm.map_depth))
    sm.frame_device(pts, "velodyne", small, None, cam, src_kind="classmap", image_size=(H, W))
    u = int((sm.map_dev != 0).any(dim=2).sum().item())           # touched cells of one frame (works in list and sweep mode)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(iters):
        sm.frame_device(pts, "velodyne", small, None, cam, src_kind="classmap", image_size=(H, W))
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / iters * 1e6
    us = e0.elapsed_time(e1) / iters * 1e3
    bpp = 16 if layout == "f32aos" else 32
    alg = n * bpp + n * 1 + u * (2 * 5 * 8) + u * 8
    print("n=%d res=%.2f %s: %.1f us/frame (wall %.1f us), U=%d, algorithmic %.2f MB -> %.1f GB/s"
          % (n, res, layout, us, wall, u, alg / 1e6, alg / us / 1e3))


if __name__ == "__main__":
    for layout in ("f64soa", "f32aos"):
        run(120000, 0.2, 200.0, layout)
        run(1000000, 0.05, 100.0, layout)
