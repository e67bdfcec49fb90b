"""NumPy restatement of the reference's planar (no-LiDAR) mode: src/mapping.py:446-488 + src/homography.py:22-76.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED: the arithmetic lives in OpenCV (`cv2.findHomography`, `cv2.warpPerspective`; requirements.txt:3, unpinned,
absent from this image) and the transform comes from a live ROS TF lookup (mapping.py:454-457); the reference has no test or
fixture for the mode.  Restated from the published algorithms:
  * findHomography(pts_src, pts_dst), method 0, four correspondences: the exact projective map (normalised DLT, h33 = 1);
  * warpPerspective(src, H, (w, h)): dst(x, y) = src(H^-1 (x, y, 1)), INTER_LINEAR, BORDER_CONSTANT 0.  OpenCV interpolates
    with 5-bit fixed-point coordinates; here (and in the kernel) plain float64 bilinear weights, rounded half to even.
What the reference then does with the warped image (mapping.py:473-481) is restated literally, including the comparison of a uint8
channel with the label NAME (`image_on_map[:, :, 0] == self.label_names[i]`), which NumPy evaluates to False everywhere: as written
the mode adds nothing and only clamps negative cells.  `match="colour"` is the evident intent (the R,G test of update_map, Q2).
"""
import numpy as np


def find_homography(pts_src, pts_dst):
    """pts_*: [n >= 4][2] -> 3x3 H with H (x, y, 1) ~ (x', y', 1), h33 = 1 (homography.py:37)."""
    src, dst = np.asarray(pts_src, dtype=np.float64), np.asarray(pts_dst, dtype=np.float64)

    def normalise(p):
        c = p.mean(axis=0)
        s = np.sqrt(2.0) / max(np.sqrt(((p - c) ** 2).sum(axis=1)).mean(), 1e-300)
        T = np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.0]])
        return (p - c) * s, T
    a, Ta = normalise(src)
    b, Tb = normalise(dst)
    rows = []
    for (x, y), (u, v) in zip(a, b):
        rows.append([-x, -y, -1, 0, 0, 0, u * x, u * y, u])
        rows.append([0, 0, 0, -x, -y, -1, v * x, v * y, v])
    _, _, vt = np.linalg.svd(np.array(rows))
    Hn = vt[-1].reshape(3, 3)
    H = np.linalg.inv(Tb) @ Hn @ Ta
    return H / H[2, 2]


def warp_perspective(src, H, out_w, out_h):
    """uint8 [h][w][ch] -> uint8 [out_h][out_w][ch] (homography.py:52)."""
    src = np.asarray(src)
    h, w = src.shape[:2]
    Hi = np.linalg.inv(np.asarray(H, dtype=np.float64))
    ys, xs = np.mgrid[0:out_h, 0:out_w].astype(np.float64)
    den = Hi[2, 0] * xs + Hi[2, 1] * ys + Hi[2, 2]
    with np.errstate(all="ignore"):
        sx = (Hi[0, 0] * xs + Hi[0, 1] * ys + Hi[0, 2]) / den
        sy = (Hi[1, 0] * xs + Hi[1, 1] * ys + Hi[1, 2]) / den
    ok = np.isfinite(sx) & np.isfinite(sy) & (np.abs(sx) < 1e9) & (np.abs(sy) < 1e9)
    sx, sy = np.where(ok, sx, -10.0), np.where(ok, sy, -10.0)
    x0, y0 = np.floor(sx), np.floor(sy)
    ax, ay = sx - x0, sy - y0
    x0, y0 = x0.astype(np.int64), y0.astype(np.int64)
    acc = np.zeros((out_h, out_w, src.shape[2]), dtype=np.float64)
    for dy in (0, 1):
        for dx in (0, 1):
            xx, yy = x0 + dx, y0 + dy
            inside = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
            wgt = np.where(dx, ax, 1.0 - ax) * np.where(dy, ay, 1.0 - ay)
            px = src[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.float64)
            acc = acc + np.where(inside, wgt, 0.0)[:, :, None] * px
    return np.clip(np.rint(acc), 0, 255).astype(np.uint8)


def anchor_points_2(map_width, map_height):
    """mapping.py:160-163"""
    return np.array([[map_width, map_width / 2, map_width / 2, map_width],
                     [map_height / 4, map_height / 4, map_height * 3 / 4, map_height * 3 / 4]], dtype=np.float64)


def planar_points_image(anchor, discretize_matrix_inv, T_local_to_base, T_velodyne_to_baselink, P):
    """mapping.py:449-463: anchor cells -> local metres -> velodyne -> image pixels ([2][4])."""
    pm = np.vstack([anchor, np.ones((1, anchor.shape[1]))])
    pl = np.matmul(discretize_matrix_inv, pm)
    pl[2, :] = 0
    pl = np.vstack([pl, np.ones((1, pl.shape[1]))])
    T = np.matmul(np.linalg.inv(T_velodyne_to_baselink), T_local_to_base)
    pv = np.matmul(T, pl)
    pi = np.matmul(P, pv)
    return pi[0:2] / pi[2]


def update_map_planar(map_local, image, points_image, anchor, map_boundary, resolution, label_names, label_colors, match="reference"):
    """mapping.py:465-488 on a warped image; mutates and returns map_local."""
    mh, mw = map_local.shape[:2]
    H = find_homography(points_image.T, anchor.T)
    image_on_map = warp_perspective(image, H, mw, mh)
    sep = int((8 - map_boundary[0][0]) / resolution)
    mask = np.ones((mh, mw), dtype=bool)
    mask[:, 0:sep] = False                            # as written (:470): a negative sep counts from the right edge
    for i in range(len(label_names)):
        if match == "reference":
            idx = np.zeros((mh, mw), dtype=bool)          # uint8 array == str  ->  False (mapping.py:474)
        else:
            idx = (image_on_map[:, :, 0] == label_colors[i][0]) & (image_on_map[:, :, 1] == label_colors[i][1])
        map_local[idx & mask, i] += 1
    map_local[map_local < 0] = 0
    return map_local
