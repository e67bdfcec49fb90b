"""NumPy restatement of the node's image pre-processing:
vision_semantic_segmentation_node.py:83-98 -- cv2.cvtColor(BGR2RGB), cv2.undistort(K, dist), cv2.resize(INTER_AREA).

TEST INFRASTRUCTURE ONLY.  OpenCV is neither in the reference tree nor installed: PARITY UNPINNED.  Restated from
OpenCV's documented algorithms:
  * undistort(src, K, dist) = remap(src, initUndistortRectifyMap(K, dist, I, K), INTER_LINEAR, BORDER_CONSTANT 0):
    for every destination pixel (u, v): x = (u-cx)/fx, y = (v-cy)/fy, r2 = x^2+y^2,
    radial = 1 + k1 r2 + k2 r2^2 + k3 r2^3, x' = x radial + 2 p1 x y + p2 (r2 + 2x^2), y' = y radial + p1 (r2 + 2y^2) + 2 p2 x y,
    source = (fx x' + cx, fy y' + cy), bilinear, zeros outside.  (OpenCV interpolates with 5-bit fixed-point weights;
    this restatement uses float weights and rounds to nearest, so it can differ from OpenCV by 1 grey level.)
  * resize(INTER_AREA) by an integer factor f: mean of each f x f box; for f = 2 OpenCV's integer path (s + 2) >> 2,
    otherwise round-half-even of the float mean.
  * resize(INTER_AREA) by any other shrink ratio (round 5): the area decimation table of computeResizeAreaTab, float accumulation
    (resize_area below).
"""
import numpy as np


def bgr_to_rgb(img):
    return img[:, :, ::-1]


def undistort(img, K, dist):
    h, w = img.shape[:2]
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    k1, k2, p1, p2, k3 = [float(v) for v in dist[:5]]
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    x, y = (u - cx) / fx, (v - cy) / fy
    r2 = x * x + y * y
    radial = 1.0 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2
    xd = x * radial + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
    yd = y * radial + p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y
    sx, sy = (fx * xd + cx).astype(np.float32), (fy * yd + cy).astype(np.float32)     # the maps are float32 in OpenCV
    x0, y0 = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
    ax, ay = (sx - x0).astype(np.float32), (sy - y0).astype(np.float32)
    out = np.zeros(img.shape, dtype=np.float32)
    for dy in (0, 1):
        for dx in (0, 1):
            xx, yy = x0 + dx, y0 + dy
            wgt = (ax if dx else 1 - ax) * (ay if dy else 1 - ay)
            ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
            px = img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.float32)
            out += (wgt * ok)[..., None] * px
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def resize_area_int(img, f):
    h, w = img.shape[0] // f, img.shape[1] // f
    s = img[:h * f, :w * f].reshape(h, f, w, f, -1).astype(np.int64).sum(axis=(1, 3))
    if f == 2:
        return ((s + 2) >> 2).astype(np.uint8)
    return np.clip(np.rint(s.astype(np.float32) * np.float32(1.0 / (f * f))), 0, 255).astype(np.uint8)


def _area_table(dsize, ssize):
    """OpenCV computeResizeAreaTab for one axis: [(source index, weight)] per destination index (float32 weights)"""
    scale = ssize / float(dsize)
    tab = []
    for d in range(dsize):
        f1 = d * scale
        f2 = f1 + scale
        cell = min(scale, ssize - f1)
        s1, s2 = int(np.ceil(f1)), int(np.floor(f2))
        s2 = min(s2, ssize - 1)
        s1 = min(s1, s2)
        ent = []
        if s1 - f1 > 1e-3:
            ent.append((s1 - 1, np.float32((s1 - f1) / cell)))
        ent += [(sx, np.float32(1.0 / cell)) for sx in range(s1, s2)]
        if f2 - s2 > 1e-3:
            ent.append((s2, np.float32(min(min(f2 - s2, 1.0), cell) / cell)))
        tab.append(ent)
    return tab


def resize_area(img, out_h, out_w):
    """cv2.resize(img, (out_w, out_h), interpolation=INTER_AREA) for a NON-integer shrink ratio (OpenCV's ResizeArea_<uchar, float>):
    every source row of a destination cell is reduced along x in float32 (value x weight, added in table order), the rows are then
    combined with the y weights the same way, and the sum is rounded half to even.  vision_semantic_segmentation_node.py:92-98."""
    h, w = img.shape[:2]
    xt, yt = _area_table(out_w, w), _area_table(out_h, h)
    src = img.astype(np.float32)
    out = np.zeros((out_h, out_w, img.shape[2]), dtype=np.uint8)
    for dy in range(out_h):
        for dx in range(out_w):
            total = np.zeros(img.shape[2], dtype=np.float32)
            for sy, beta in yt[dy]:
                buf = np.zeros(img.shape[2], dtype=np.float32)
                for sx, alpha in xt[dx]:
                    buf = (buf + src[sy, sx] * alpha).astype(np.float32)
                total = (total + buf * beta).astype(np.float32)
            out[dy, dx] = np.clip(np.rint(total), 0, 255).astype(np.uint8)
    return out


def preprocess_area(bgr, K, dist, out_h, out_w):
    """:83-98 with an IMAGE_SCALE that is not 1 / integer: BGR2RGB, undistort, general INTER_AREA"""
    img = bgr_to_rgb(bgr)
    if K is not None and dist is not None:
        img = undistort(img, K, dist)
    return np.ascontiguousarray(resize_area(np.ascontiguousarray(img), out_h, out_w))


def preprocess(bgr, K=None, dist=None, factor=1):
    """vision_semantic_segmentation_node.py:83-98 in order: BGR2RGB, undistort, INTER_AREA downscale."""
    img = bgr_to_rgb(bgr)
    if K is not None and dist is not None:
        img = undistort(img, K, dist)
    if factor > 1:
        img = resize_area_int(img, factor)
    return np.ascontiguousarray(img)
