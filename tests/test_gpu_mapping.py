"""HIP mapping kernels (through the C ABI, via the reference-shaped SemanticMapping class) against
the NumPy oracle and the reference-generated golden fixtures.  Integer results are compared
bit-exactly; with a float64 grid the log-odds are compared bit-exactly too."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "mapping_*.npz")))


def make_sm(boundary, res, cm, use_intensity, device, grid_dtype="f64", range_max=100.0):
    from vision_semantic_segmentation_amd import SemanticMapping, get_cfg_defaults
    from vision_semantic_segmentation_amd.utils.logger import MyLogger
    cfg = get_cfg_defaults()
    cfg.MAPPING.BOUNDARY = boundary
    cfg.MAPPING.RESOLUTION = res
    cfg.MAPPING.PCD.USE_INTENSITY = bool(use_intensity)
    cfg.MAPPING.PCD.RANGE_MAX = range_max
    cfg.MAPPING.GRID_DTYPE = grid_dtype
    sm = SemanticMapping(cfg, device=device, logger=MyLogger("test", quiet=True))
    sm.confusion_matrix = np.array(cm)
    return sm


class _Cam(object):
    def __init__(self, P):
        self.P = P


def dense(idx, val, shape):
    m = np.zeros(shape)
    m[idx[:, 0], idx[:, 1]] = val
    return m


def pose_of(g):
    from vision_semantic_segmentation_amd.utils import Pose
    return Pose.from_array(g["pose7"]) if g["pose7"].size else None


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[8:-4] for p in CASES])
def test_project_pcd_and_update_map_match_golden(path, cuda_device):
    g = np.load(path)
    boundary, res = g["boundary"].tolist(), float(g["resolution"])
    sm = make_sm(boundary, res, g["cm"], g["use_intensity"], cuda_device)
    cam, pose, frame = _Cam(g["P"]), pose_of(g), str(g["frame"])
    mp, lab = sm.project_pcd(g["pcd"], frame, g["image"], pose, cam)
    assert mp.dtype == np.float64 and lab.dtype == np.uint8
    assert np.array_equal(mp, g["masked_pcd"], equal_nan=True)       # order-preserving compaction, exact copies
    assert np.array_equal(lab, g["label"])
    # NumPy grid (the reference's calling convention): touched rows travel, arithmetic on the GPU
    grid = np.zeros((sm.map_height, sm.map_width, sm.map_depth))
    out = sm.update_map(grid, mp, lab)
    assert out is grid
    assert np.array_equal(grid, dense(g["map_idx"], g["map_val"], grid.shape))
    out = sm.update_map(grid, g["masked_pcd2"], g["label2"])
    assert np.array_equal(grid, dense(g["map2_idx"], g["map2_val"], grid.shape))


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[8:-4] for p in CASES])
def test_fused_mapping_matches_golden(path, cuda_device):
    """mapping() = fused project+vote+apply on the device grid; two frames accumulate."""
    g = np.load(path)
    boundary, res = g["boundary"].tolist(), float(g["resolution"])
    sm = make_sm(boundary, res, g["cm"], g["use_intensity"], cuda_device)
    cam, pose, frame = _Cam(g["P"]), pose_of(g), str(g["frame"])
    sm.pcd, sm.pcd_frame_id = g["pcd"], frame
    assert sm.map is None
    sm.mapping(g["image"], pose, cam)
    assert np.array_equal(sm.map, dense(g["map_idx"], g["map_val"], sm.map.shape))
    pcd2 = g["pcd"].copy()
    pcd2[0:2] += 0.37
    sm.pcd = pcd2
    sm.mapping(g["image"], pose, cam)
    assert np.array_equal(sm.map, dense(g["map2_idx"], g["map2_val"], sm.map.shape))
    # scratch contract: the vote masks are all zero again
    assert int(sm.grid.cell_mask.abs().sum().item()) == 0


def test_device_grid_f32_and_device_tensors(cuda_device):
    """float32 grid + CUDA-tensor inputs: identity CM gives exact small integers in fp32."""
    import torch
    g = np.load(CASES[0])
    boundary, res = g["boundary"].tolist(), float(g["resolution"])
    sm = make_sm(boundary, res, g["cm"], g["use_intensity"], cuda_device, grid_dtype="f32")
    mp = torch.from_numpy(g["masked_pcd"]).to(cuda_device)
    lab = torch.from_numpy(g["label"]).to(cuda_device)
    out = sm.update_map(sm.map_dev, mp, lab)
    assert out.dtype == torch.float32 and out.is_cuda
    ref = dense(g["map_idx"], g["map_val"], tuple(out.shape))
    tol = 1e-3 * max(1.0, np.abs(ref).max())          # north_star: grid log-odds within 1e-3
    assert np.max(np.abs(out.cpu().numpy().astype(np.float64) - ref)) <= tol


def test_classmap_source_equals_colour_image_path(cuda_device):
    """The fused class-map source (argmax map + LUT + nearest index map) must equal the reference
    route: uint8 cast -> INTER_NEAREST upscale -> palette -> project_pcd -> update_map."""
    import torch
    from oracle import mapping_oracle as mo
    from vision_semantic_segmentation_amd import synthetic as syn
    from vision_semantic_segmentation_amd.camera import camera_setup_1
    rng = np.random.default_rng(11)
    H, W, lh, lw = 1080, 1920, 266, 476
    cam = camera_setup_1()
    pcd = syn.make_cloud(rng, 50000, cam.K, cam.R, cam.t, W, H)
    small = syn.make_label_map(rng, lh, lw, tile=7)
    boundary = syn.centred_boundary(mo.PCD_ORIGIN_OFFSET[:2], 100.0)
    cm = syn.log_confusion(5)
    sm = make_sm(boundary, 0.2, cm, True, cuda_device)
    sm.frame_device(pcd, "velodyne", torch.from_numpy(small).to(cuda_device), None, cam, src_kind="classmap",
                    image_size=(H, W))
    image = mo.semantic_image_from_labels(small, H, W)
    grid = np.zeros((sm.map_height, sm.map_width, 5))
    cfg = dict(range_max=100.0, boundary=boundary, resolution=0.2, label_names=mo.LABELS_NAMES,
               label_colors=mo.LABEL_COLORS, confusion_matrix=cm, use_pcd_intensity=True)
    mo.mapping_frame(grid, pcd, "velodyne", image, None, cam.P, cfg)
    assert np.count_nonzero(grid) > 1000
    assert np.array_equal(sm.map, grid)
    # and the colourised image the node would publish
    from vision_semantic_segmentation_amd.vision_semantic_segmentation_node import colorize_labels_device
    col = colorize_labels_device(torch.from_numpy(small).to(cuda_device), H, W).cpu().numpy()
    assert np.array_equal(col, image)


def test_projection_indices_bit_exact_float32_aos(cuda_device):
    """avl_project_points on a float32 [N,4] cloud: int32 pixel indices and mask vs the oracle."""
    import ctypes as C
    import torch
    from oracle import mapping_oracle as mo
    from vision_semantic_segmentation_amd import _lib, synthetic as syn
    from vision_semantic_segmentation_amd.camera import camera_setup_6
    rng = np.random.default_rng(3)
    cam = camera_setup_6()
    H, W, n = 1440, 1920, 200000
    pts32 = syn.make_cloud(rng, n, cam.K, cam.R, cam.t, W, H, dtype=np.float32)      # [4,N] f32
    aos = torch.from_numpy(np.ascontiguousarray(pts32.T)).to(cuda_device)            # [N,4]
    ixy = torch.empty((2, n), dtype=torch.int32, device=cuda_device)
    mask = torch.empty(n, dtype=torch.uint8, device=cuda_device)
    P = (C.c_double * 12)(*cam.P.ravel().tolist())
    rc = _lib.lib().avl_project_points(C.c_void_p(aos.data_ptr()), n, _lib.AVL_F32, 16, 4, P, None, 100.0, W, H,
                                       C.c_void_p(ixy.data_ptr()), C.c_void_p(mask.data_ptr()), None)
    _lib.check(rc)
    torch.cuda.synchronize()
    image = np.zeros((H, W, 3), dtype=np.uint8)
    with np.errstate(all="ignore"):
        _, _, IXY, m = mo.project_pcd(pts32.astype(np.float64), "velodyne", image, None, cam.P, 100.0, return_debug=True)
    assert np.array_equal(ixy.cpu().numpy(), IXY)
    assert np.array_equal(mask.cpu().numpy().astype(bool), m)


def test_truncating_divisions_on_integer_boundaries(cuda_device):
    """The kernels replace the IEEE division of `(P X)[0:2] / (P X)[2]` and `(p - b) / res` (mapping.py:375, :408-409) by a
    reciprocal-multiply + residual correction and fall back to the exact division for lanes whose quotient is within 2^-44 of an
    integer (csrc/mapping.hip, div_i32_numpy).  Adversarial inputs: quotients that are EXACTLY integers, one ulp either side of one,
    zero, tiny, huge, negative, zero / tiny / infinite denominators -- int32 pixel indices and grid cells must equal NumPy's."""
    import ctypes as C
    import torch
    from oracle import mapping_oracle as mo
    from vision_semantic_segmentation_amd import _lib
    rng = np.random.default_rng(11)
    # --- pixels: P = [I | 0] so that (P X)[0:2] / (P X)[2] = (x / z, y / z)
    P = np.zeros((3, 4)); P[0, 0] = P[1, 1] = P[2, 2] = 1.0
    zs = np.concatenate([rng.uniform(0.1, 50.0, 4000), [1.0, 0.5, 3.0, 0.1, 1e-320, 1e-300, 1e300, 0.0, np.inf, 7.0, 1e-3, 2.0 ** -20]])
    ks = rng.integers(0, 1900, size=zs.size).astype(np.float64)
    xs = ks * zs                                                   # x / z is k up to one rounding: exactly k for many z
    pts = []
    for dx in (0, 1, -1, 3):                                       # ... and a few ulps either side
        x = xs.copy()
        for _ in range(abs(dx)):
            x = np.nextafter(x, np.inf if dx > 0 else -np.inf)
        pts.append(np.stack([x, 0.25 * x, zs, np.ones_like(zs)]))
    pts.append(np.stack([rng.uniform(-50, 50, 20000), rng.uniform(-50, 50, 20000), rng.uniform(-5, 50, 20000), np.ones(20000)]))
    pcd = np.ascontiguousarray(np.concatenate(pts, axis=1))        # float64 [4, N]
    n = pcd.shape[1]
    dev = torch.from_numpy(pcd).to(cuda_device)
    ixy = torch.empty((2, n), dtype=torch.int32, device=cuda_device)
    mask = torch.empty(n, dtype=torch.uint8, device=cuda_device)
    Pc = (C.c_double * 12)(*P.ravel().tolist())
    W, H = 1920, 1440
    rc = _lib.lib().avl_project_points(C.c_void_p(dev.data_ptr()), n, _lib.AVL_F64, 8, 8 * n, Pc, None, 1e9, W, H,
                                       C.c_void_p(ixy.data_ptr()), C.c_void_p(mask.data_ptr()), None)
    _lib.check(rc)
    torch.cuda.synchronize()
    with np.errstate(all="ignore"):
        _, _, IXY, m = mo.project_pcd(pcd, "velodyne", np.zeros((H, W, 3), dtype=np.uint8), None, P, 1e9, return_debug=True)
    assert np.array_equal(ixy.cpu().numpy(), IXY)
    assert np.array_equal(mask.cpu().numpy().astype(bool), m)
    # --- grid cells: points ON cell boundaries (b + k res, and one ulp either side), resolutions that are and are not binary fractions
    for res in (0.2, 0.05, 0.25, 0.3):
        half = 400 * res
        boundary = [[mo.PCD_ORIGIN_OFFSET[0] - half, mo.PCD_ORIGIN_OFFSET[0] + half], [mo.PCD_ORIGIN_OFFSET[1] - half, mo.PCD_ORIGIN_OFFSET[1] + half]]
        sm = make_sm(boundary, res, np.eye(5), False, cuda_device)
        k = rng.integers(-3, sm.map_height + 3, size=6000).astype(np.float64)
        l = rng.integers(-3, sm.map_width + 3, size=6000).astype(np.float64)
        xw = (boundary[0][0] + k * res) - mo.PCD_ORIGIN_OFFSET[0]       # update_map adds the offset back: (x + off - b00) / res ~ k
        yw = (boundary[1][0] + l * res) - mo.PCD_ORIGIN_OFFSET[1]
        cols = []
        for d in (0, 1, -1):
            x, y = xw.copy(), yw.copy()
            if d:
                x = np.nextafter(x, np.inf * d)
                y = np.nextafter(y, -np.inf * d)
            cols.append(np.stack([x, y, np.zeros_like(x), np.ones_like(x)]))
        pc = np.ascontiguousarray(np.concatenate(cols, axis=1))
        lab = np.tile(np.array(mo.LABEL_COLORS[0], dtype=np.uint8).reshape(3, 1), (1, pc.shape[1]))
        grid = np.zeros((sm.map_height, sm.map_width, sm.map_depth))
        sm.update_map(grid, pc, lab)
        want = np.zeros_like(grid)
        mo.update_map(want, pc, lab, boundary, res, mo.LABELS_NAMES, mo.LABEL_COLORS, np.eye(5), False)
        assert np.array_equal(grid, want), "resolution %s" % res
        assert want.sum() > 1000


def test_edge_cases(cuda_device):
    """Empty cloud, a cloud with no survivor, and argument errors reported through avl_last_error."""
    g = np.load(CASES[0])
    boundary, res = g["boundary"].tolist(), float(g["resolution"])
    sm = make_sm(boundary, res, g["cm"], True, cuda_device)
    cam = _Cam(g["P"])
    mp, lab = sm.project_pcd(np.zeros((4, 0)), "velodyne", g["image"], None, cam)
    assert mp.shape == (4, 0) and lab.shape == (3, 0)
    behind = np.array([[-5.0, 0.0, 0.0, 1.0], [np.nan, 0, 0, 0]]).T
    mp, lab = sm.project_pcd(behind, "velodyne", g["image"], None, cam)
    assert mp.shape == (4, 0)
    grid = np.zeros((sm.map_height, sm.map_width, 5))
    assert sm.update_map(grid, mp, lab) is grid and not grid.any()
    assert sm.project_pcd(None, "velodyne", g["image"], None, cam) is None
    with pytest.raises(ValueError):
        sm.project_pcd(np.zeros((3, 5)), "velodyne", g["image"], None, cam)


def test_full_size_properties(cuda_device):
    """BASELINE sizes (config C: 120k points, 2000x2000 grid @0.2 m; config E: 1M points, 4000x4000
    @0.05 m) through properties that need no oracle at full size plus the oracle itself (it
    finishes in < 1 s at these sizes): idempotence under point duplication (Q1), permutation
    invariance, additivity of frames."""
    import torch
    from oracle import mapping_oracle as mo
    from vision_semantic_segmentation_amd import synthetic as syn
    from vision_semantic_segmentation_amd.camera import camera_setup_1
    cam = camera_setup_1().scaled(1.0, 1080 / 1440.0)
    for n, res, half in ((120000, 0.2, 200.0), (1000000, 0.05, 100.0)):
        rng = np.random.default_rng(n)
        H, W = 1080, 1920
        pcd = syn.make_cloud(rng, n, cam.K, cam.R, cam.t, W, H)
        image = syn.colorize(syn.make_label_map(rng, H, W))
        boundary = syn.centred_boundary(mo.PCD_ORIGIN_OFFSET[:2], half)
        cm = syn.log_confusion(5)
        sm = make_sm(boundary, res, cm, True, cuda_device)
        img_d = torch.from_numpy(image).to(cuda_device)
        sm.frame_device(pcd, "velodyne", img_d, None, cam)
        one = sm.map_dev.clone()
        # oracle at full size
        grid = np.zeros((sm.map_height, sm.map_width, 5))
        cfg = dict(range_max=100.0, boundary=boundary, resolution=res, label_names=mo.LABELS_NAMES,
                   label_colors=mo.LABEL_COLORS, confusion_matrix=cm, use_pcd_intensity=True)
        mo.mapping_frame(grid, pcd, "velodyne", image, None, cam.P, cfg)
        assert np.array_equal(one.cpu().numpy(), grid)
        # scratch is clean again: vote mask all zero, list cursors and tickets of the partitioned lists back to zero
        assert not bool(sm.grid.cell_mask.any()) and not bool(sm.grid.counter[4:].any())
        # duplicated + permuted cloud gives the same single-frame delta
        sm.map = np.zeros_like(grid)
        perm = rng.permutation(2 * n)
        sm.frame_device(np.concatenate([pcd, pcd], axis=1)[:, perm], "velodyne", img_d, None, cam)
        assert torch.equal(sm.map_dev, one)
        # frames accumulate: a second frame on top of the first, against the oracle doing the same
        sm.frame_device(pcd, "velodyne", img_d, None, cam)
        mo.mapping_frame(grid, pcd, "velodyne", image, None, cam.P, cfg)
        assert np.array_equal(sm.map, grid)


def test_dense_votes_in_one_sweep_round(cuda_device):
    """A coarse grid under a dense cloud: 3377 voted cells land in ONE 16384-cell round of the byte-mask sweep (its LDS list is
    sized for a fully voted round); the grid must still equal the oracle's bit for bit, frame after frame."""
    import torch
    from oracle import mapping_oracle as mo
    from vision_semantic_segmentation_amd import synthetic as syn
    from vision_semantic_segmentation_amd.camera import camera_setup_1
    cam = camera_setup_1().scaled(1.0, 1080 / 1440.0)
    rng = np.random.default_rng(7)
    H, W = 1080, 1920
    pcd = syn.make_cloud(rng, 400000, cam.K, cam.R, cam.t, W, H)
    image = syn.colorize(syn.make_label_map(rng, H, W))
    boundary = syn.centred_boundary(mo.PCD_ORIGIN_OFFSET[:2], 40.0)          # 80 m x 80 m at 0.5 m: 160 x 160 cells
    cm = syn.log_confusion(5)
    sm = make_sm(boundary, 0.5, cm, True, cuda_device)
    img_d = torch.from_numpy(image).to(cuda_device)
    grid = np.zeros((sm.map_height, sm.map_width, 5))
    cfg = dict(range_max=100.0, boundary=boundary, resolution=0.5, label_names=mo.LABELS_NAMES,
               label_colors=mo.LABEL_COLORS, confusion_matrix=cm, use_pcd_intensity=True)
    for _ in range(2):
        sm.frame_device(pcd, "velodyne", img_d, None, cam)
        mo.mapping_frame(grid, pcd, "velodyne", image, None, cam.P, cfg)
    voted = (grid != 0).any(axis=2).reshape(-1)
    per_round = [int(voted[i:i + 16384].sum()) for i in range(0, voted.size, 16384)]
    assert max(per_round) > 2048, "the cloud must crowd one round (voted cells per round: %r)" % (per_round,)
    assert np.array_equal(sm.map, grid)


def test_callbacks_and_replay(cuda_device, tmp_path):
    """The ROS-shaped path: pcd/pose/image callbacks with nearest-stamp matching (mapping.py:172-290), input
    recording, and the offline replay driver (mapping_replay.py:175-192) reproducing the same grid."""
    from vision_semantic_segmentation_amd import replay
    from vision_semantic_segmentation_amd.utils import Header, Message, Pose, Stamp
    g = np.load([c for c in CASES if "W_world" in c][0])
    boundary, res = g["boundary"].tolist(), float(g["resolution"])
    sm = make_sm(boundary, res, g["cm"], g["use_intensity"], cuda_device)
    sm.record_inputs = True
    sm.cam6.P = g["P"]                      # the fixture's camera (cam6 intrinsics at this image size)
    pose = Pose.from_array(g["pose7"])
    pcd2 = g["pcd"].copy()
    pcd2[0:2] += 0.37
    # two clouds and poses around two image stamps; the nearest one must be picked
    sm.pcd_callback(Message(Header(Stamp(10, 0), "map"), points=g["pcd"]))
    sm.pcd_callback(Message(Header(Stamp(11, 0), "map"), points=pcd2.T))            # [N,4] accepted too
    sm.pose_callback(Message(Header(Stamp(10, 0)), pose=pose))
    sm.pose_callback(Message(Header(Stamp(11, 0)), pose=pose))
    sm.image_callback(Message(Header(Stamp(10, 100), "camera6"), data=g["image"]))
    assert np.array_equal(sm.map, dense(g["map_idx"], g["map_val"], sm.map.shape))
    sm.image_callback(Message(Header(Stamp(10, 900000000), "camera6"), data=g["image"]))
    assert np.array_equal(sm.map, dense(g["map2_idx"], g["map2_val"], sm.map.shape))
    with pytest.raises(ValueError):
        sm.image_callback(Message(Header(Stamp(12, 0), "camera9"), data=g["image"]))
    # record -> save -> replay
    d = str(tmp_path)
    sm.save_inputs(d)
    frames = replay.load_frames(d)
    assert len(frames) == 2 and frames[0]["pcd_frame_id"] == "map"
    sm2 = make_sm(boundary, res, g["cm"], g["use_intensity"], cuda_device)
    grid = replay.mapping_replay(sm2, frames, _Cam(g["P"]))
    assert np.array_equal(grid, sm.map)


def test_semantic_point_cloud_records(cuda_device):
    """create_point_cloud (utils_ros.py:31-59): GPU records == the reference's per-point struct.pack, == the host helper."""
    import struct
    from vision_semantic_segmentation_amd.utils import create_point_cloud
    g = np.load(CASES[0])
    sm = make_sm(g["boundary"].tolist(), float(g["resolution"]), g["cm"], True, cuda_device)
    rec, m = sm.semantic_cloud_device(g["pcd"], str(g["frame"]), g["image"], None, _Cam(g["P"]))
    assert m == g["masked_pcd"].shape[1]
    data = rec.cpu().numpy().tobytes()
    xyz, rgb = g["masked_pcd"][0:3].T, g["label"].T
    for k in (0, 1, m // 2, m - 1):
        rgba = struct.unpack("I", struct.Struct("BBBB").pack(int(rgb[k, 0]), int(rgb[k, 1]), int(rgb[k, 2]), 255))[0]   # utils_ros.py:51
        expect = struct.pack("<fffI", xyz[k, 0], xyz[k, 1], xyz[k, 2], rgba)
        assert data[16 * k:16 * k + 16] == expect
    host = create_point_cloud(xyz, rgb, frame_id="velodyne")
    assert host["point_step"] == 16 and host["width"] == m and host["data"] == data


def test_pointcloud2_unpack_on_device(cuda_device):
    """avl_unpack_pointcloud2 (pcd_callback, mapping.py:172-183): a hand-packed velodyne-style payload (32-byte records, padding
    bytes that are not zero, NaN in each of the four fields) unpacked on the GPU == the host unpack, slot for slot; and a
    frame mapped from the device points == the frame mapped from the host-unpacked float64 [4,M] array, bit for bit."""
    import torch
    from test_fixtures_misc import _pointcloud2
    from oracle import mapping_oracle as mo
    from vision_semantic_segmentation_amd import synthetic as syn
    from vision_semantic_segmentation_amd.camera import camera_setup_1
    from vision_semantic_segmentation_amd.mapping import unpack_pointcloud2
    rng = np.random.default_rng(17)
    H, W = 480, 640
    cam = camera_setup_1().scaled(W / 1920.0, H / 1440.0)
    cloud = syn.make_cloud(rng, 30000, cam.K, cam.R, cam.t, W, H).astype(np.float32)       # has NaN / inf / far points
    cloud[3, 5::97] = np.nan                                                                 # NaN intensity only: read_points skips those too
    cloud[1, 11::101] = np.nan
    pts = [tuple(float(v) for v in cloud[:, k]) for k in range(cloud.shape[1])]
    msg = _pointcloud2(pts)
    host = unpack_pointcloud2(msg)                                                           # float64 [4, M]
    boundary = syn.centred_boundary(mo.PCD_ORIGIN_OFFSET[:2], 100.0)
    a = make_sm(boundary, 0.2, syn.log_confusion(5), True, cuda_device)
    dev_pts, count = a.unpack_pointcloud2_device(msg)
    assert tuple(dev_pts.shape) == (30000, 4) and dev_pts.dtype == torch.float32
    got = dev_pts.cpu().numpy()
    bad = np.isnan(cloud).any(axis=0)
    assert int(count.item()) == int((~bad).sum()) == host.shape[1]
    assert np.array_equal(got[~bad].T.astype(np.float64), host, equal_nan=True)
    assert np.isnan(got[bad, 0]).all() and np.array_equal(got[bad, 1:], cloud[1:, bad].T, equal_nan=True)
    image = torch.from_numpy(syn.colorize(syn.make_label_map(rng, H, W))).to(cuda_device)
    a.frame_device(dev_pts, "velodyne", image, None, cam, src_kind="rgb")
    b = make_sm(boundary, 0.2, syn.log_confusion(5), True, cuda_device)
    b.frame_device(host, "velodyne", image, None, cam, src_kind="rgb")
    assert torch.equal(a.map_dev, b.map_dev) and float(a.map_dev.abs().sum()) > 0
    # through the callback: UNPACK_ON_DEVICE makes pcd_callback queue the device tensor
    a.unpack_on_device = True
    msg.header = type("H", (), {"frame_id": "velodyne", "stamp": 1.0})()
    a.pcd_callback(msg)
    assert a.pcd_queue[-1].is_cuda
    assert torch.equal(torch.nan_to_num(a.pcd_queue[-1]), torch.nan_to_num(dev_pts))
    # a different record layout and an empty cloud
    msg2 = _pointcloud2(pts[:100], point_step=48, offsets=(24, 28, 32, 4))
    d2, c2 = a.unpack_pointcloud2_device(msg2)
    assert torch.equal(torch.nan_to_num(d2), torch.nan_to_num(dev_pts[:100])) and int(c2.item()) == int((~bad[:100]).sum())
    d0, c0 = a.unpack_pointcloud2_device(_pointcloud2([]))
    assert d0.shape[0] == 0 and int(c0.item()) == 0
    # an explicit (side) stream, alternating with the current one: zero-fill, upload and kernel all go on the stream handed in
    # (ADVICE r3), the shared staging buffers are ordered across the two streams by events
    side = torch.cuda.Stream(device=cuda_device)
    for rep in range(6):
        if rep % 2 == 0:
            d3, c3 = a.unpack_pointcloud2_device(msg, stream=side.cuda_stream)
            side.synchronize()
        else:
            d3, c3 = a.unpack_pointcloud2_device(msg2)
        want_n, want = (30000, dev_pts) if rep % 2 == 0 else (100, dev_pts[:100])
        assert d3.shape[0] == want_n and int(c3.item()) == int((~bad[:want_n]).sum())
        assert torch.equal(torch.nan_to_num(d3), torch.nan_to_num(want))


@pytest.mark.parametrize("match", ["reference", "colour"])
@pytest.mark.parametrize("grid_dtype,x0", [("f64", -4.0), ("f32", -4.0), ("f64", 9.5)])
def test_planar_mode_matches_the_restated_oracle(match, grid_dtype, x0, cuda_device):
    """SURVEY 8f row 2, update_map_planar (src/mapping.py:446-488; PARITY UNPINNED: OpenCV / ROS TF absent): the warp + class test +
    clamp kernel against oracle/planar_oracle.py -- float64 arithmetic in the same order on both sides, so the grids are identical."""
    import torch
    from oracle import mapping_oracle as mo
    from oracle import planar_oracle as po
    from vision_semantic_segmentation_amd import synthetic as syn
    from vision_semantic_segmentation_amd.camera import camera_setup_1
    rng = np.random.default_rng(23)
    H, W = 360, 480
    cam = camera_setup_1().scaled(W / 1920.0, H / 1440.0)
    # the reference discretises local y as resolution * row + BOUNDARY[1][1] (mapping.py:148-152): this boundary puts the anchor
    # rectangle 16 .. 36 m ahead of the car and +-10 m to its sides, in view of camera 1
    # x0 = 9.5: the grid starts beyond x = 8 m, `sep` is negative and the reference's `mask[:, 0:sep] = 0` masks all but the last 15 columns
    boundary = [[x0, x0 + 40.0], [-60.0, -20.0]]
    sm = make_sm(boundary, 0.1, np.eye(5), True, cuda_device, grid_dtype=grid_dtype)
    sm.planar_match = match
    image = syn.colorize(syn.make_label_map(rng, H, W, tile=16))
    image[100:140, 200:260] = [128, 64, 7]                    # R,G of "road" with another blue: still a match (Q2)
    T_local_to_base = np.eye(4)
    T_local_to_base[:3, 3] = [-1.0, 0.5, 0.0]
    start = rng.normal(scale=2.0, size=(sm.map_height, sm.map_width, sm.map_depth)).astype(np.float64 if grid_dtype == "f64" else np.float32)
    want = start.astype(np.float64).copy()
    pts_img = po.planar_points_image(po.anchor_points_2(sm.map_width, sm.map_height), sm.discretize_matrix_inv, T_local_to_base,
                                     mo.velodyne_to_baselink(), cam.P)
    po.update_map_planar(want, image, pts_img, po.anchor_points_2(sm.map_width, sm.map_height), boundary, 0.1, mo.LABELS_NAMES,
                         mo.LABEL_COLORS, match=match)
    # CUDA-tensor grid, in place
    grid_dev = torch.from_numpy(start).to(cuda_device)
    out = sm.update_map_planar(grid_dev, image, cam, T_local_to_base=T_local_to_base)
    assert out is grid_dev
    got = grid_dev.cpu().numpy().astype(np.float64)
    assert np.array_equal(got, want.astype(start.dtype).astype(np.float64))
    if match == "colour":
        votes = want - np.maximum(start, 0)
        if x0 < 8:
            assert votes.sum() > 1000                             # the warped image does land on the grid
        else:
            assert votes[:, :sm.map_width - 15].max() <= 0 and votes[:, sm.map_width - 15:].sum() > 10
    # NumPy grid (the reference's calling convention), and through mapping() with a TF stand-in
    grid_np = start.copy()
    assert sm.update_map_planar(grid_np, image, cam, T_local_to_base=T_local_to_base) is grid_np
    assert np.array_equal(grid_np.astype(np.float64), got)
    sm.depth_method = "planar"
    sm.local_to_base_lookup = lambda stamp: T_local_to_base
    sm.map = start
    sm.mapping(image, None, cam)
    assert np.array_equal(sm.map.astype(np.float64), got)
