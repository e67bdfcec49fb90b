"""The NumPy oracle of the mapping path against fixtures generated from the REFERENCE's own
project_pcd / update_map (oracle/gen_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest

from oracle import mapping_oracle as mo

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "mapping_*.npz")))


def dense(idx, val, shape):
    m = np.zeros(shape)
    m[idx[:, 0], idx[:, 1]] = val
    return m


def load_case(path):
    g = np.load(path)
    pose = g["pose7"] if g["pose7"].size else None
    boundary = g["boundary"].tolist()
    res = float(g["resolution"])
    return g, pose, boundary, res


def test_fixtures_present():
    assert len(CASES) >= 4


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[8:-4] for p in CASES])
def test_oracle_matches_reference_outputs(path):
    g, pose, boundary, res = load_case(path)
    with np.errstate(all="ignore"):
        mp, lab = mo.project_pcd(g["pcd"], str(g["frame"]), g["image"], pose, g["P"], float(g["range_max"]), g["T_v2b"])
    # bit-exact: the masked points are copies of the inputs, the labels are gathered bytes
    assert np.array_equal(mp, g["masked_pcd"], equal_nan=True)
    assert np.array_equal(lab, g["label"])
    mh, mw = mo.map_dims(boundary, res)
    grid = np.zeros((mh, mw, 5))
    mo.update_map(grid, mp, lab, boundary, res, mo.LABELS_NAMES, mo.LABEL_COLORS, g["cm"], bool(g["use_intensity"]))
    assert np.array_equal(grid, dense(g["map_idx"], g["map_val"], grid.shape))
    # second frame accumulated on the first (a9)
    pcd2 = g["pcd"].copy()
    pcd2[0:2] += 0.37
    with np.errstate(all="ignore"):
        mp2, lab2 = mo.project_pcd(pcd2, str(g["frame"]), g["image"], pose, g["P"], float(g["range_max"]), g["T_v2b"])
    assert np.array_equal(mp2, g["masked_pcd2"], equal_nan=True) and np.array_equal(lab2, g["label2"])
    mo.update_map(grid, mp2, lab2, boundary, res, mo.LABELS_NAMES, mo.LABEL_COLORS, g["cm"], bool(g["use_intensity"]))
    assert np.array_equal(grid, dense(g["map2_idx"], g["map2_val"], grid.shape))


def test_camera_restatement(golden_dir):
    cam = np.load(os.path.join(golden_dir, "camera.npz"))
    for cid in (1, 6):
        c = mo.camera_matrices(cid)
        assert np.array_equal(c["P"], cam["P%d" % cid])
        assert np.array_equal(c["T"], cam["T%d" % cid])


def test_quirks_pinned():
    """Q1 buffered add, Q2 blue ignored, Q3 truncation, Q4 INT_MIN casts -- on hand-made inputs."""
    boundary, res = [[0.0, 10.0], [0.0, 10.0]], 1.0
    off = np.array(mo.PCD_ORIGIN_OFFSET)
    def at(x, y, inten=5.0):
        return [x - off[0], y - off[1], 0.3, inten]
    pcd = np.array([at(2.5, 3.5), at(2.6, 3.4), at(2.7, 3.3, 1.0), at(-0.5, 4.5), at(np.nan, 1.0), at(3e9, 1.0), at(4.5, 4.5)]).T
    label = np.array([[128, 64, 128], [128, 64, 128], [255, 255, 7], [140, 140, 200], [128, 64, 128], [128, 64, 128], [1, 2, 3]],
                     dtype=np.uint8).T
    cm = np.arange(25, dtype=np.float64).reshape(5, 5) + 1
    grid = np.zeros((10, 10, 5))
    mo.update_map(grid, pcd, label, boundary, res, mo.LABELS_NAMES, mo.LABEL_COLORS, cm, True)
    expect = np.zeros_like(grid)
    expect[2, 3] += cm[:, 0]            # two road points, one add (Q1)
    expect[2, 3] += cm[:, 2]            # lane matched on R,G although B = 7 (Q2)
    expect[2, 3, 2] += 2                # intensity 1.0 < 2
    expect[0, 4] += cm[:, 1]            # x = -0.5 truncates to cell 0 (Q3)
    assert np.array_equal(grid, expect) # NaN / 3e9 rows rejected (Q4), unmatched colour ignored


def test_matmul_order_is_fma_chain():
    """The HIP kernel hard-codes OpenBLAS's K=4 order (p0*x0 then fused multiply-adds).  Check the
    host BLAS of this machine agrees; a different order would move results by <= 1 ulp only."""
    rng = np.random.default_rng(5)
    P = rng.normal(size=(3, 4)) * 1000
    X = np.vstack([rng.uniform(-80, 80, size=(3, 4096)), np.ones((1, 4096))])
    ref = P @ X
    # an FMA is the exactly rounded a*b+c: emulate with float128-free exact arithmetic via fractions on a sample
    from fractions import Fraction
    bad = 0
    for j in range(0, 4096, 64):
        for r in range(3):
            s = float(Fraction(P[r, 0]) * Fraction(X[0, j]))
            for k in (1, 2, 3):
                s = float(Fraction(P[r, k]) * Fraction(X[k, j]) + Fraction(s))
            bad += (s != ref[r, j])
    if bad:
        pytest.xfail("host BLAS sums K=4 products in a different order (%d/192 differ by an ulp)" % bad)


def test_resize_nearest_and_colour():
    lab = np.arange(12, dtype=np.uint8).reshape(3, 4) % 19
    big = mo.resize_nearest(lab, 9, 10)
    assert big.shape == (9, 10) and big[0, 0] == lab[0, 0] and big[8, 9] == lab[2, 3]
    assert np.array_equal(big[:, 5], lab[[0, 0, 0, 1, 1, 1, 2, 2, 2], 2])
    img = mo.semantic_image_from_labels(lab, 9, 10)
    assert tuple(img[0, 0]) == tuple(mo.PALETTE_19[lab[0, 0]])
