"""Shared pieces of the full-size, oracle-anchored GPU tests: the bench's own workload (BASELINE configs[2]: a 1080 x 1920
frame + 120 k points, bench.py's seeds) and a cache of oracle forwards, so that the tests which look at the same
(weights seed, frame) pay for the 16 s torch-CPU fp32 forward once per session.  Test infrastructure."""
import functools

import numpy as np

H, W, NPTS = 1080, 1920, 120000


@functools.lru_cache(maxsize=None)
def bench_workload():
    """(image uint8 [H,W,3], cloud float64 [4,N], camera) exactly as bench.py rank 0 draws them"""
    from vision_semantic_segmentation_amd import synthetic as syn
    from vision_semantic_segmentation_amd.camera import camera_setup_1
    rng = np.random.default_rng(1)
    cam = camera_setup_1().scaled(1.0, H / 1440.0, imSize=[W, H])
    image = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    cloud = syn.make_cloud(rng, NPTS, cam.K, cam.R, cam.t, W, H)
    return image, cloud, cam


@functools.lru_cache(maxsize=None)
def state_dict(weight_seed):
    from vision_semantic_segmentation_amd.network import random_state_dict
    return random_state_dict(weight_seed)


def image_for(image_seed, h, w):
    """image_seed "bench": the bench frame (1080 x 1920 only); an int: default_rng(seed) uniform bytes"""
    if image_seed == "bench":
        assert (h, w) == (H, W)
        return bench_workload()[0]
    return np.random.default_rng(image_seed).integers(0, 256, size=(h, w, 3), dtype=np.uint8)


@functools.lru_cache(maxsize=None)
def oracle_logits(weight_seed, image_seed, h, w):
    """fp32 logits [19, h/4-4, w/4-4] of oracle/network_oracle.forward_logits (torch CPU)"""
    from oracle import network_oracle as no
    return no.forward_logits(state_dict(weight_seed), image_for(image_seed, h, w))[0]


@functools.lru_cache(maxsize=None)
def heavy_tailed_state_dict(seed, calib_hw=(160, 224), gamma=(0.05, 8.0), outlier=(20.0, 50.0), outlier_frac=0.015, residual_gain=1.0):
    """oracle/checkpoint_like.heavy_tailed_state_dict, cached per argument set for the session"""
    from oracle.checkpoint_like import heavy_tailed_state_dict as make
    return make(seed, calib_hw, gamma, outlier, outlier_frac, residual_gain)
