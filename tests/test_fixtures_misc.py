"""Pins for the small reference-held pieces of rows a6 / a10 (fixtures written by `python oracle/gen_golden.py misc` from the
reference's own mapillary_visualization.py, src/data/confusion_matrix.py and src/config/base_cfg.py), plus the host-side
wire-format helpers that have no GPU in them: unpack_pointcloud2 and the node's sensor_msgs/Image decoding."""
import json
import os
import struct
import types

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def misc():
    return np.load(os.path.join(GOLD, "misc.npz"))


def test_palette_and_get_labels_equal_the_references(misc):
    """labels.PALETTE_19 / get_labels() == get_labels(config/config_19.json) of the reference (:9-18)."""
    from oracle import mapping_oracle as mo
    from vision_semantic_segmentation_amd import labels
    assert np.array_equal(np.array(labels.PALETTE_19), misc["palette"])
    assert [d["name"] for d in labels.get_labels()] == list(misc["palette_names"])
    assert [d["color"] for d in labels.get_labels()] == misc["palette"].tolist()
    assert np.array_equal(np.array(mo.PALETTE_19), misc["palette"])


def test_oracle_apply_color_map_equals_the_references(misc):
    """oracle.mapping_oracle.apply_color_map == mapillary_visualization.apply_color_map (:70-89), 2-D and batched,
    ids outside the table stay black."""
    from oracle import mapping_oracle as mo
    assert np.array_equal(mo.apply_color_map(misc["labels_2d"]), misc["colors_2d"])
    assert np.array_equal(mo.apply_color_map(misc["labels_3d"]), misc["colors_3d"])


def test_vote_lut_matches_colour_matching_on_the_reference_palette(misc):
    """The class-map fast path votes through a LUT built from the palette; it must mark exactly the classes whose colour the
    reference's R,G comparison (mapping.py:419) would match on a colourised image."""
    from vision_semantic_segmentation_amd import labels
    lut = labels.vote_lut(misc["palette"], labels.LABEL_COLORS)
    for k, col in enumerate(misc["palette"]):
        want = 0
        for i, lc in enumerate(labels.LABEL_COLORS):
            if col[0] == lc[0] and col[1] == lc[1]:
                want |= 1 << i
        assert lut[k] == want
    assert [int(np.log2(lut[k])) for k in labels.LABELS] == [0, 1, 2, 3, 4]      # base_cfg LABELS -> map class index
    assert int((lut != 0).sum()) == 5


def test_confusion_matrix_equals_the_references(misc, tmp_path):
    from vision_semantic_segmentation_amd.data.confusion_matrix import ConfusionMatrix
    path = os.path.join(str(tmp_path), "cfn_mtx.npy")
    np.save(path, misc["cfn_mtx"])
    cm = ConfusionMatrix(load_path=path)
    idx = misc["cfn_indices"].tolist()
    assert np.array_equal(cm.get_submatrix(idx), misc["cfn_sub"])
    assert np.array_equal(cm.get_submatrix(idx, to_probability=True), misc["cfn_sub_prob"])
    assert np.array_equal(cm.get_submatrix(idx, to_probability=True, use_log=True), misc["cfn_sub_log"])       # bit for bit
    assert np.array_equal(cm.get_submatrix(idx, use_log=True), misc["cfn_sub_log_without_prob"])
    assert np.array_equal(cm.get_submatrix([18, 0, 7], True, True), misc["cfn_sub_perm"])
    assert len(cm) == int(misc["cfn_len"]) and np.array_equal(cm[3], misc["cfn_row3"])
    assert cm.get_submatrix([]) == []
    errs = []
    for bad in ([0, 19], [-1, 2], list(range(19)) + [0]):
        with pytest.raises(ValueError) as e:
            cm.get_submatrix(bad)
        errs.append(str(e.value.args[0]))
    assert errs == list(misc["cfn_errors"])


def test_config_defaults_equal_base_cfg():
    """config.get_cfg_defaults() carries the reference's default for every key base_cfg.py defines.  Two values differ on
    purpose and are checked to be the documented substitutes: the paths on the authors' NAS (DATASET_CONFIG, MODEL.WEIGHT)."""
    from vision_semantic_segmentation_amd.config import get_cfg_defaults
    ref = json.load(open(os.path.join(GOLD, "base_cfg.json")))
    cfg = get_cfg_defaults()
    substitutes = {"VISION_SEM_SEG.SEM_SEG_NETWORK.DATASET_CONFIG": "", "VISION_SEM_SEG.SEM_SEG_NETWORK.MODEL.WEIGHT": ""}
    seen = []

    def walk(r, c, path):
        for k, v in r.items():
            p = path + k
            assert hasattr(c, k), "missing config key %s" % p
            cv = getattr(c, k)
            if isinstance(v, dict):
                walk(v, cv, p + ".")
            elif p in substitutes:
                assert v.startswith("/mnt/avl_shared/") and cv == substitutes[p]
                seen.append(p)
            else:
                got = [list(x) if isinstance(x, (list, tuple)) else x for x in cv] if isinstance(cv, (list, tuple)) else cv
                assert got == v, "%s: %r != reference default %r" % (p, got, v)
    walk(ref, cfg, "")
    assert sorted(seen) == sorted(substitutes)


def _pointcloud2(points, point_step=32, offsets=(0, 4, 8, 16)):
    """hand-packed sensor_msgs/PointCloud2 (velodyne layout: x y z f32, 4 bytes of padding, intensity f32, ring u16, padding)"""
    n = len(points)
    buf = bytearray(b"\xAB" * (n * point_step))              # padding bytes are NOT zero
    for i, p in enumerate(points):
        for o, v in zip(offsets, p):
            struct.pack_into("<f", buf, i * point_step + o, v)
        struct.pack_into("<H", buf, i * point_step + 20, i % 64)
    fields = [types.SimpleNamespace(name=nm, offset=o, datatype=7, count=1) for nm, o in zip(("x", "y", "z", "intensity"), offsets)]
    fields.append(types.SimpleNamespace(name="ring", offset=20, datatype=4, count=1))
    return types.SimpleNamespace(data=bytes(buf), fields=fields, point_step=point_step, row_step=n * point_step, width=n, height=1,
                                 is_bigendian=False, is_dense=False, header=types.SimpleNamespace(frame_id="velodyne", stamp=None))


def test_unpack_pointcloud2_hand_packed_buffer():
    """mapping.py:172-183 (read_points(skip_nans=True) + the fill loop): x, y, z, intensity as float64 columns in message
    order, points with a NaN in any of the four dropped, no garbage tail."""
    from vision_semantic_segmentation_amd.mapping import unpack_pointcloud2
    nan = float("nan")
    pts = [(1.5, -2.25, 0.125, 7.0), (nan, 1.0, 1.0, 1.0), (3.0, 4.0, 5.0, 14.5), (1.0, nan, 0.0, 2.0), (9.0, 8.0, 7.0, nan),
           (-0.0, 1e-30, 3.4e38, 0.0), (float("inf"), 1.0, 2.0, 3.0)]
    got = unpack_pointcloud2(_pointcloud2(pts))
    keep = [p for p in pts if not any(np.isnan(v) for v in p)]
    assert got.dtype == np.float64 and got.shape == (4, len(keep)) and got.flags["C_CONTIGUOUS"]
    want = np.array(keep, dtype=np.float32).astype(np.float64).T          # float32 on the wire, widened exactly
    assert np.array_equal(got, want)
    assert np.isinf(got[0, -1])                                            # inf is not NaN: kept, as read_points keeps it
    # a different field order / stride
    got2 = unpack_pointcloud2(_pointcloud2(pts, point_step=48, offsets=(24, 28, 32, 4)))
    assert np.array_equal(got2, want)
    empty = unpack_pointcloud2(_pointcloud2([]))
    assert empty.shape == (4, 0)
    # an organised cloud (height 2) whose rows are padded to row_step, and a big-endian payload (refused, not misread)
    flat = _pointcloud2(pts[:6])
    rows = np.frombuffer(flat.data, dtype=np.uint8).reshape(2, 3 * 32)
    padded = np.full((2, 3 * 32 + 20), 0xCD, dtype=np.uint8)
    padded[:, :3 * 32] = rows
    org = types.SimpleNamespace(**dict(vars(flat), data=padded.tobytes(), width=3, height=2, row_step=3 * 32 + 20))
    assert np.array_equal(unpack_pointcloud2(org), unpack_pointcloud2(flat))
    import pytest
    with pytest.raises(NotImplementedError):
        unpack_pointcloud2(types.SimpleNamespace(**dict(vars(flat), is_bigendian=True)))


def test_imgmsg_decoding_handles_row_padding():
    """sensor_msgs/Image -> ndarray (what cv_bridge's imgmsg_to_cv2(passthrough) gives the reference at
    vision_semantic_segmentation_node.py:76-80 / mapping.py:266): bytes payload, `step` may exceed width * channels."""
    from vision_semantic_segmentation_amd.mapping import _imgmsg_to_array
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(5, 7, 3), dtype=np.uint8)
    step = 7 * 3 + 11
    rows = np.full((5, step), 0xEE, dtype=np.uint8)
    rows[:, :21] = img.reshape(5, 21)
    msg = types.SimpleNamespace(data=rows.tobytes(), height=5, width=7, step=step, encoding="bgr8")
    assert np.array_equal(_imgmsg_to_array(msg), img)
    mono = types.SimpleNamespace(data=rows.tobytes(), height=5, width=7, step=step, encoding="mono8")
    assert np.array_equal(_imgmsg_to_array(mono), rows[:, :7])
