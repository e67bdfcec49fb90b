"""Seeded synthetic inputs of the shapes BASELINE.json names.

There is no dataset or rosbag offline, so tests, fixtures and bench.py all draw from here:
a back-projected LiDAR cloud that is ~95 % in-frame for the given camera plus a fixed 5 % of
adversarial points (behind the sensor, NaN, inf, far, sub-pixel-negative), a blob-structured
class-id map, and its palette-coloured RGB image.
"""
import numpy as np

from .labels import PALETTE_19


def make_label_map(rng, h, w, num_classes=19, tile=32):
    """uint8[h,w] class ids in coarse tiles so that neighbouring pixels correlate."""
    th, tw = (h + tile - 1) // tile, (w + tile - 1) // tile
    coarse = rng.integers(0, num_classes, size=(th, tw), dtype=np.uint8)
    return np.repeat(np.repeat(coarse, tile, axis=0), tile, axis=1)[:h, :w].copy()


def colorize(label_map, palette=PALETTE_19):
    """uint8[h,w] -> uint8[h,w,3] through the 19-entry palette (classes >= len(palette) -> black)."""
    lut = np.zeros((256, 3), dtype=np.uint8)
    lut[:len(palette)] = np.asarray(palette, dtype=np.uint8)
    return lut[label_map]


def make_cloud(rng, n, K, R, t, img_w, img_h, adversarial_frac=0.05, depth=(2.0, 80.0), dtype=np.float64):
    """float[4,n] (rows x,y,z,intensity) in the velodyne frame.

    Regular points: pixel (u,v) uniform in the image, depth d uniform, X_cam = d K^-1 [u,v,1],
    X_velo = R^T (X_cam - t)  (camera model of src/camera.py:22-35: x_cam = R X + t).
    """
    n_adv = int(round(n * adversarial_frac))
    n_reg = n - n_adv
    u = rng.uniform(0.0, img_w, n_reg)
    v = rng.uniform(0.0, img_h, n_reg)
    d = rng.uniform(depth[0], depth[1], n_reg)
    xc = np.linalg.inv(K) @ np.vstack([u, v, np.ones(n_reg)]) * d
    xv = R.T @ (xc - t.reshape(3, 1))
    pts = np.empty((4, n), dtype=np.float64)
    pts[0:3, :n_reg] = xv
    pts[3, :n_reg] = rng.uniform(0.0, 30.0, n_reg)
    if n_adv:
        adv = np.empty((4, n_adv))
        adv[0:3] = rng.uniform(-60.0, 60.0, (3, n_adv))       # many behind the sensor / out of frame
        adv[3] = rng.uniform(0.0, 30.0, n_adv)
        k = np.arange(n_adv)
        adv[0, k % 7 == 0] = np.nan
        adv[1, k % 7 == 1] = np.inf
        adv[2, k % 7 == 2] = -np.inf
        adv[0, k % 7 == 3] = 1.0e12
        adv[0, k % 7 == 4] = 0.0                               # x == 0 fails 0 < x
        adv[0, k % 7 == 5] = 150.0                             # beyond RANGE_MAX
        pts[:, n_reg:] = adv
    perm = rng.permutation(n)
    return np.ascontiguousarray(pts[:, perm]).astype(dtype)


def centred_boundary(offset_xy, half_extent):
    """BOUNDARY centred on the sensor origin in map coordinates (SURVEY.md 8d 'Grid')."""
    ox, oy = offset_xy
    return [[ox - half_extent, ox + half_extent], [oy - half_extent, oy + half_extent]]


def log_confusion(num=5, diag=0.75, floor=0.05):
    """A log-probability confusion matrix log(floor + diag*I) rows normalised (float parity case)."""
    m = np.full((num, num), floor) + diag * np.eye(num)
    m = m / m.sum(axis=1, keepdims=True)
    return np.log(m)
