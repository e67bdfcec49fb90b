"""NumPy restatement of the reference's LiDAR->image->BEV-grid mapping path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the reference
file:line it follows.  Parity status: pinned by tests/golden/mapping_*.npz, which
oracle/gen_golden.py produced by calling the reference's own
``SemanticMapping.project_pcd`` / ``update_map`` (src/mapping.py:357-444) in the build
container.  Pieces whose arithmetic lives in packages absent from the reference tree
(ROS ``tf.transformations``, OpenCV ``resize``) are restated from their published
algorithms and are marked "parity unpinned" where they appear.

The arithmetic deliberately stays literal (``np.matmul``, ``astype(np.int32)``, fancy-index
``+=``) so that NumPy's own quirks are inherited, not re-derived.
"""
import math

import numpy as np

# --------------------------------------------------------------------------------------
# Constants (data, not code) of the reference
# --------------------------------------------------------------------------------------

# src/mapping.py:404 (and the negated pose at :232-233): pcd origin w.r.t. the map origin
PCD_ORIGIN_OFFSET = (1369.0496826171875, 562.84814453125, 0.0)

# src/config/base_cfg.py:47-57
LABELS = [2, 1, 8, 10, 3]
LABELS_NAMES = ["road", "crosswalk", "lane", "vegetation", "sidewalk"]
LABEL_COLORS = [[128, 64, 128], [140, 140, 200], [255, 255, 255], [107, 142, 35], [244, 35, 232]]

# config/config_19.json "labels"[k]["color"], k = network class id (19 Mapillary classes)
PALETTE_19 = [
    [196, 196, 196], [140, 140, 200], [128, 64, 128], [244, 35, 232], [70, 70, 70],
    [220, 20, 60], [255, 0, 0], [255, 0, 100], [255, 255, 255], [70, 130, 180],
    [107, 142, 35], [100, 128, 160], [153, 153, 153], [220, 220, 0], [119, 11, 32],
    [0, 60, 100], [0, 0, 142], [0, 0, 230], [0, 0, 70],
]

# src/camera.py:102-135 (K, Rt, image size, plumb-bob distortion) for camera 1 and 6
_CAM = {
    1: dict(
        K=[[1826.998004, 0.0, 1174.548672], [0.0, 1802.603136, 776.028597], [0.0, 0.0, 1.0]],
        Rt=[[1.5426360183850896e-01, -6.8597082105982421e-02, 9.8564556584725482e-01, 4.7539938241243362e-02],
            [-9.8802970661938061e-01, -1.0912135033489312e-02, 1.5387730224640517e-01, 3.1389930844306946e-01],
            [1.9996357324159053e-04, -9.9758476614047986e-01, -6.9459300162133530e-02, -5.5608768016099930e-02],
            [0.0, 0.0, 0.0, 1.0]],
        dist=[-0.136981, 0.043159, 0.006235, 0.018954, 0.0]),
    6: dict(
        K=[[1790.634474, 0.0, 973.099292], [0.0, 1785.950534, 803.294457], [0.0, 0.0, 1.0]],
        Rt=[[-2.1022535018250471e-01, -9.2112145235168197e-02, 9.7330398891652492e-01, -1.4076865278184414e-02],
            [-9.7735897207277012e-01, -4.6117027185500481e-03, -2.1153763709301088e-01, -3.1732881069183350e-01],
            [2.3973774202277975e-02, -9.9573795995643932e-01, -8.9057134763516621e-02, -7.2184838354587555e-02],
            [0.0, 0.0, 0.0, 1.0]],
        dist=[-0.191070, 0.100324, 0.004250, -0.003317, 0.0]),
}


# --------------------------------------------------------------------------------------
# Small geometry helpers
# --------------------------------------------------------------------------------------

def homogenize(x):
    """src/utils/utils.py:68-70 -- append a row of ones."""
    return np.vstack((x, np.ones((1, x.shape[1]))))


def dehomogenize(x):
    """src/utils/utils.py:73-75 -- divide by the last row (no zero guard, Q4)."""
    return x[:-1] / x[-1]


def camera_matrices(cam_id):
    """src/camera.py:22-35 + :102-135 -> dict(K, R, t, P, T, dist).  P = K [R | t]."""
    c = _CAM[cam_id]
    K = np.array(c["K"])
    Rt = np.array(c["Rt"])
    R = Rt[0:3, 0:3].T                      # camera.py:111 / :129
    t = -np.matmul(R, Rt[0:3, 3:4])          # camera.py:112 / :130
    P_norm = np.concatenate([R, t], axis=1)  # camera.py:27
    P = np.matmul(K, P_norm)                 # camera.py:28
    T = np.vstack([P_norm, np.zeros((1, 4))])
    T[-1, -1] = 1
    return dict(K=K, R=R, t=t, P=P, T=T, dist=np.array(c["dist"]), imSize=[1920, 1440])


def quaternion_matrix(quaternion):
    """tf.transformations.quaternion_matrix (ROS tf, not in the reference tree; parity
    unpinned).  Published algorithm (C. Gohlke, transformations.py): quaternion is
    (x, y, z, w); returns the 4x4 homogeneous rotation."""
    q = np.array(quaternion[:4], dtype=np.float64, copy=True)
    nq = np.dot(q, q)
    if nq < np.finfo(float).eps * 4.0:
        return np.identity(4)
    q *= math.sqrt(2.0 / nq)
    q = np.outer(q, q)
    return np.array((
        (1.0 - q[1, 1] - q[2, 2], q[0, 1] - q[2, 3], q[0, 2] + q[1, 3], 0.0),
        (q[0, 1] + q[2, 3], 1.0 - q[0, 0] - q[2, 2], q[1, 2] - q[0, 3], 0.0),
        (q[0, 2] - q[1, 3], q[1, 2] + q[0, 3], 1.0 - q[0, 0] - q[1, 1], 0.0),
        (0.0, 0.0, 0.0, 1.0)), dtype=np.float64)


def transform_from_pose(pose7):
    """src/utils/utils_ros.py:104-111 -> TransformerROS.fromTranslationRotation =
    translation_matrix(t) . quaternion_matrix(q) (ROS tf; parity unpinned).
    pose7 = (tx, ty, tz, qx, qy, qz, qw)."""
    M = np.identity(4)
    M[:3, 3] = pose7[:3]
    return np.dot(M, quaternion_matrix(pose7[3:7]))


def velodyne_to_baselink():
    """src/mapping.py:165-170: euler_matrix(0, 0.140, 0) (static xyz = rotation about +y,
    tf.transformations; parity unpinned) with translation (2.64, 0, 1.98)."""
    c, s = math.cos(0.140), math.sin(0.140)
    T = np.array([[c, 0.0, s, 0.0], [0.0, 1.0, 0.0, 0.0], [-s, 0.0, c, 0.0], [0.0, 0.0, 0.0, 1.0]])
    T[0:3, -1::] = np.array([[2.64, 0, 1.98]]).T
    return T


def map_dims(boundary, resolution):
    """src/mapping.py:115-116: map_height indexes x (boundary[0]), map_width indexes y."""
    h = int((boundary[0][1] - boundary[0][0]) / resolution)
    w = int((boundary[1][1] - boundary[1][0]) / resolution)
    return h, w


def log_confusion_submatrix(cfn_mtx, indices):
    """src/data/confusion_matrix.py:25-48,59-63 with to_probability=True, use_log=True."""
    sub = np.asarray(cfn_mtx)[np.ix_(indices, indices)]
    sub = sub / np.sum(sub, axis=1)[:, np.newaxis]
    return np.log(sub)


# --------------------------------------------------------------------------------------
# a7: project_pcd  (src/mapping.py:357-389)
# --------------------------------------------------------------------------------------

def project_pcd(pcd, pcd_frame_id, image, pose7, P, range_max, T_velodyne_to_baselink=None,
                return_debug=False):
    """Restates SemanticMapping.project_pcd, src/mapping.py:357-389.

    pcd float64[4,N]; image uint8[H,W,3]; pose7 or None; P float64[3,4].
    Returns (masked_pcd float64[4,M], label uint8[3,M]); with return_debug also
    (IXY int32[2,N], mask bool[N])."""
    if pcd is None:                                                      # :366
        return None
    if pcd_frame_id != "velodyne":                                       # :367
        T_base_to_origin = transform_from_pose(pose7)                    # :368
        T_origin_to_velodyne = np.linalg.inv(np.matmul(T_base_to_origin, T_velodyne_to_baselink))  # :369
        with np.errstate(all="ignore"):
            pcd_velodyne = np.matmul(T_origin_to_velodyne, homogenize(pcd[0:3, :]))                # :371
    else:
        pcd_velodyne = homogenize(pcd[0:3, :])                           # :373

    with np.errstate(all="ignore"):
        IXY = dehomogenize(np.matmul(P, pcd_velodyne)).astype(np.int32)  # :375  (Q3, Q4)

        mask_positive = np.logical_and(0 < pcd_velodyne[0, :], pcd_velodyne[0, :] < range_max)  # :378 (Q5)

    mask = np.logical_and(np.logical_and(0 <= IXY[0, :], IXY[0, :] < image.shape[1]),       # :381
                          np.logical_and(0 <= IXY[1, :], IXY[1, :] < image.shape[0]))       # :382
    mask = np.logical_and(mask, mask_positive)                           # :383

    masked_pcd = pcd[:, mask]                                            # :385 (Q6: original frame)
    image_idx = IXY[:, mask]                                             # :386
    label = image[image_idx[1, :], image_idx[0, :]].T                    # :387

    if return_debug:
        return masked_pcd, label, IXY, mask
    return masked_pcd, label


# --------------------------------------------------------------------------------------
# a8: update_map  (src/mapping.py:391-444)
# --------------------------------------------------------------------------------------

def cell_indices(pcd, boundary, resolution, map_height, map_width):
    """src/mapping.py:403-411 -> (pcd_pixel int32[2,M], on_grid_mask bool[M])."""
    normal = np.array([[0.0, 0.0, 1.0]]).T                               # :403
    pcd_origin_offset = np.array([[PCD_ORIGIN_OFFSET[0]], [PCD_ORIGIN_OFFSET[1]], [PCD_ORIGIN_OFFSET[2]]])  # :404
    with np.errstate(all="ignore"):
        pcd_local = pcd[0:3] + pcd_origin_offset                         # :405
        pcd_on_map = pcd_local - np.matmul(normal, np.matmul(normal.T, pcd_local))  # :406
        pcd_pixel = ((pcd_on_map[0:2, :] - np.array([[boundary[0][0]], [boundary[1][0]]]))
                     / resolution).astype(np.int32)                      # :408-409 (Q3, Q4)
    on_grid_mask = np.logical_and(np.logical_and(0 <= pcd_pixel[0, :], pcd_pixel[0, :] < map_height),
                                  np.logical_and(0 <= pcd_pixel[1, :], pcd_pixel[1, :] < map_width))  # :410-411
    return pcd_pixel, on_grid_mask


def update_map(map, pcd, label, boundary, resolution, label_names, label_colors, confusion_matrix,
               use_pcd_intensity):
    """Restates SemanticMapping.update_map, src/mapping.py:391-444.  Mutates and returns map."""
    map_height, map_width = map.shape[0], map.shape[1]
    label_colors = np.asarray(label_colors)
    pcd_pixel, on_grid_mask = cell_indices(pcd, boundary, resolution, map_height, map_width)

    for i, label_name in enumerate(label_names):                         # :414
        # Q2: logical_and(*rows) == logical_and(R_match, G_match, out=B_match)
        idx = np.logical_and(*(label == label_colors[i].reshape(3, 1)))  # :419
        idx_mask = np.logical_and(idx, on_grid_mask)                     # :420
        # Q1: buffered fancy-index add -- one add per unique cell
        map[pcd_pixel[0, idx_mask], pcd_pixel[1, idx_mask], :] += confusion_matrix[:, i].reshape(1, -1)  # :424
        if not use_pcd_intensity:                                        # :427
            continue
        if label_name == "lane":                                         # :431
            with np.errstate(all="ignore"):
                intensity_mask = np.logical_or(pcd[3] < 2, pcd[3] > 14)  # :432
            intensity_mask = np.logical_and(intensity_mask, idx_mask)    # :433
            map[pcd_pixel[0, intensity_mask], pcd_pixel[1, intensity_mask], i] += 2  # :437
    return map


# --------------------------------------------------------------------------------------
# a6: label map -> full-resolution colour image  (vision_semantic_segmentation_node.py:101-116)
# --------------------------------------------------------------------------------------

def resize_nearest(src, dst_h, dst_w):
    """cv2.resize(src, (dst_w, dst_h), interpolation=INTER_NEAREST)
    (vision_semantic_segmentation_node.py:109-110).  OpenCV is not in the reference tree
    nor installed: parity unpinned.  Published rule (imgproc resizeNN, non-"exact"):
    sx = min(int(floor(dx * (src_w / dst_w))), src_w - 1) in double arithmetic."""
    sh, sw = src.shape[:2]
    fx = sw / float(dst_w)
    fy = sh / float(dst_h)
    sx = np.minimum(np.floor(np.arange(dst_w) * fx).astype(np.int64), sw - 1)
    sy = np.minimum(np.floor(np.arange(dst_h) * fy).astype(np.int64), sh - 1)
    return src[sy[:, None], sx[None, :]]


def apply_color_map(label_array, palette=PALETTE_19):
    """src/network/deeplab_v3_plus/data/utils/mapillary_visualization.py:70-89 (2-D case)."""
    color_array = np.zeros(label_array.shape + (3,), dtype=np.uint8)
    for label_id, color in enumerate(palette):
        color_array[label_array == label_id] = color
    return color_array


def semantic_image_from_labels(labels_small, out_h, out_w, palette=PALETTE_19):
    """vision_semantic_segmentation_node.py:102,109-116: uint8 cast, nearest upscale, colourise."""
    lab = np.asarray(labels_small).astype(np.uint8)
    return apply_color_map(resize_nearest(lab, out_h, out_w), palette)


# --------------------------------------------------------------------------------------
# a9: one frame of SemanticMapping.mapping  (src/mapping.py:292-321), arithmetic only
# --------------------------------------------------------------------------------------

def mapping_frame(map, pcd, pcd_frame_id, semantic_image, pose7, P, cfg):
    """project_pcd -> update_map as called at src/mapping.py:314,319.
    cfg: dict(range_max, T_velodyne_to_baselink, boundary, resolution, label_names, label_colors,
    confusion_matrix, use_pcd_intensity)."""
    pcd_in_range, pcd_label = project_pcd(pcd, pcd_frame_id, semantic_image, pose7, P, cfg["range_max"],
                                          cfg.get("T_velodyne_to_baselink"))
    return update_map(map, pcd_in_range, pcd_label, cfg["boundary"], cfg["resolution"], cfg["label_names"],
                      cfg["label_colors"], cfg["confusion_matrix"], cfg["use_pcd_intensity"])
