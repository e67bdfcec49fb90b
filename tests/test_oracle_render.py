"""Renderer oracle vs fixtures made by the reference's own src/renderer.py (pure NumPy functions)."""
import os

import numpy as np

from oracle import renderer_oracle as ro

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "render.npz"))


def test_render_bev_map_matches_reference():
    assert np.array_equal(ro.render_bev_map(G["grid"], G["colors"]), G["bev"])


def test_render_thresholds_match_reference():
    with np.errstate(all="ignore"):
        assert np.array_equal(ro.render_bev_map_with_thresholds(G["grid"], G["colors"], G["priority"], G["thresholds"]), G["bev_thr"])
        assert np.array_equal(ro.render_bev_map_with_thresholds(G["grid"], G["colors"]), G["bev_thr_default"])


def test_box_filter_properties():
    rng = np.random.default_rng(0)
    x = rng.normal(size=(9, 11, 5))
    y = ro.apply_filter(x)
    k = float(np.float32(1) / np.float32(9))
    assert np.isclose(y[4, 5, 2], k * x[3:6, 4:7, 2].sum())
    assert np.isclose(y[0, 0, 1], k * (x[0, 0, 1] + 2 * x[0, 1, 1] + 2 * x[1, 0, 1] + 4 * x[1, 1, 1]))      # reflect-101 corner
    assert np.allclose(ro.apply_filter(np.ones((6, 7, 3))), 9 * k)
