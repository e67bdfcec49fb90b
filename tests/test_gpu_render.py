"""GPU renderer (avl_render_* / avl_grid_box_filter) vs the reference-generated fixtures and the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "render.npz"))


def test_render_matches_reference_fixtures(cuda_device):
    import torch
    from vision_semantic_segmentation_amd import renderer as rr
    assert np.array_equal(rr.render_bev_map(G["grid"], G["colors"]), G["bev"])
    assert np.array_equal(rr.render_bev_map_with_thresholds(G["grid"], G["colors"], G["priority"].tolist(), G["thresholds"].tolist()), G["bev_thr"])
    assert np.array_equal(rr.render_bev_map_with_thresholds(G["grid"], G["colors"]), G["bev_thr_default"])
    dev = torch.from_numpy(G["grid"]).to(cuda_device)
    out = rr.render_bev_map(dev, G["colors"])
    assert out.is_cuda and np.array_equal(out.cpu().numpy(), G["bev"])
    with pytest.raises(ValueError):
        rr.render_bev_map(G["grid"], G["colors"][:3])


def test_box_filter_then_render_like_the_end_of_a_run(cuda_device):
    """mapping.py:332-334: apply_filter then render_bev_map, at full grid size (2000 x 2000 x 5)."""
    from oracle import renderer_oracle as ro
    from vision_semantic_segmentation_amd import renderer as rr
    small = ro.apply_filter(G["grid"])
    got = rr.apply_filter(G["grid"])
    assert np.max(np.abs(got - small)) <= 1e-12 * max(1.0, np.abs(small).max())       # 9-term double sum, same order
    rng = np.random.default_rng(2)
    big = np.zeros((2000, 2000, 5))
    idx = rng.integers(0, 2000, size=(200000, 2))
    big[idx[:, 0], idx[:, 1]] = rng.normal(size=(200000, 5)) * 3
    f = rr.apply_filter(big)
    assert np.allclose(f, ro.apply_filter(big), rtol=0, atol=1e-12)
    assert np.array_equal(rr.render_bev_map(f, G["colors"]), ro.render_bev_map(f, G["colors"]))
