"""SURVEY 8f row 3, evaluation part: avl_eval_map (through vision_semantic_segmentation_amd.evaluation) against
outputs of the reference's own convert_labels / Test.iou (tests/golden/eval.npz) and the CPU restatement.
Counting is integer work: label maps and IoU / missing-rate values must be bit-equal."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "eval.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_convert_labels_bit_exact(gold, tag, cuda_device):
    import torch
    from vision_semantic_segmentation_amd import evaluation as ev
    cmap, mask = gold[tag + "_cmap"], gold[tag + "_mask"]
    got = ev.convert_labels(cmap)
    assert got.dtype == np.float64 and np.array_equal(got, gold[tag + "_labels"])
    assert np.array_equal(ev.convert_labels(cmap, mask), gold[tag + "_labels_masked"])
    dev = ev.convert_labels(torch.from_numpy(cmap).to(cuda_device), torch.from_numpy(mask).to(cuda_device))
    assert dev.is_cuda and dev.dtype == torch.uint8
    assert np.array_equal(dev.cpu().numpy().astype(np.float64), gold[tag + "_labels_masked"])


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("shift", ["s0", "s1"])
def test_iou_bit_equal_to_the_reference(gold, tag, shift, cuda_device, tmp_path, capsys):
    from oracle import evaluation_oracle as eo
    from vision_semantic_segmentation_amd import evaluation as ev
    sw, sh = [int(v) for v in gold["%s_%s_shift" % (tag, shift)]]
    np.save(os.path.join(str(tmp_path), "truth.npy"), gold[tag + "_truth"])
    t = ev.Test(ground_truth_dir=str(tmp_path), shift_w=sw, shift_h=sh)
    lab = gold[tag + "_labels_masked"]
    gm = gold[tag + "_truth"][sw:lab.shape[0] + sw, sh:lab.shape[1] + sh]
    iou_lists, miss = t.iou(gm, lab, verbose=True)
    assert np.array_equal(np.array(iou_lists), gold["%s_%s_iou" % (tag, shift)])
    assert miss == float(gold["%s_%s_miss" % (tag, shift)])
    out = capsys.readouterr().out
    assert "IOU for road:" in out and "Overall Missing rate: {}".format(miss) in out
    # test_single_map: colour map in, same numbers when no mask is involved
    cmap = gold[tag + "_cmap"]
    iou2, miss2 = t.test_single_map(cmap)
    ref = eo.iou(gm, eo.convert_labels(cmap))
    assert iou2 == ref[0] and miss2 == ref[2]


def test_edge_cases(cuda_device, tmp_path):
    from oracle import evaluation_oracle as eo
    from vision_semantic_segmentation_amd import evaluation as ev
    rng = np.random.default_rng(3)
    # ground truth with values the reference never special-cases: negatives, fractions, large ints
    h, w = 33, 1025
    truth = rng.choice(np.array([0.0, 1.0, 2.0, 3.0, -1.0, 2.5, 9.0, 300.0]), size=(h, w))
    lab = rng.integers(0, 6, size=(h, w)).astype(np.float64)
    np.save(os.path.join(str(tmp_path), "truth.npy"), truth)
    t = ev.Test(ground_truth_dir=str(tmp_path))
    iou_lists, miss = t.iou(truth, lab)
    ref = eo.iou(truth, lab)
    assert iou_lists == ref[0] and miss == ref[2]
    # an absent class divides 0.0 by 0.0 exactly like the reference
    z = np.zeros((8, 8))
    z[0, 0] = 1
    with pytest.raises(ZeroDivisionError):
        t.iou(z, np.zeros((8, 8)))
    # shape mismatch is the reference's broadcast error
    with pytest.raises(ValueError):
        t.iou(np.zeros((8, 9)), np.zeros((8, 8)))
    with pytest.raises(FileNotFoundError):
        ev.Test(ground_truth_dir=os.path.join(str(tmp_path), "nope"))


def test_finish_run_filters_renders_saves_and_evaluates(cuda_device, tmp_path, capsys):
    """mapping.py:323-345 end to end on the GPU: apply_filter -> render_bev_map -> global_map.png -> Test.test_single_map."""
    import torch
    from PIL import Image
    from oracle import evaluation_oracle as eo
    from oracle import mapping_oracle as mo
    from oracle import renderer_oracle as ro
    from vision_semantic_segmentation_amd import SemanticMapping, get_cfg_defaults
    from vision_semantic_segmentation_amd import synthetic as syn
    from vision_semantic_segmentation_amd.camera import camera_setup_1
    from vision_semantic_segmentation_amd.utils.logger import MyLogger
    cfg = get_cfg_defaults()
    cfg.MAPPING.BOUNDARY = syn.centred_boundary(mo.PCD_ORIGIN_OFFSET[:2], 60.0)
    cfg.MAPPING.RESOLUTION = 0.5
    cfg.OUTPUT_DIR = str(tmp_path)
    cfg.GROUND_TRUTH_DIR = str(tmp_path)
    H, W = 240, 320
    rng = np.random.default_rng(5)
    cam = camera_setup_1().scaled(W / 1920.0, H / 1440.0)
    sm = SemanticMapping(cfg, device=cuda_device, logger=MyLogger("t", quiet=True))
    for _ in range(3):
        classmap = syn.make_label_map(rng, H, W)
        sm.pcd, sm.pcd_frame_id = syn.make_cloud(rng, 30000, cam.K, cam.R, cam.t, W, H), "velodyne"
        sm.mapping(syn.colorize(classmap), None, cam)
    grid_before = sm.map.copy()
    truth = rng.integers(0, 4, size=(sm.map_height, sm.map_width)).astype(np.float64)
    np.save(os.path.join(str(tmp_path), "truth.npy"), truth)
    color = sm.finish_run()
    out = capsys.readouterr().out
    # the same sequence on the CPU restatements
    smooth = ro.apply_filter(grid_before)
    ref_color = ro.render_bev_map(smooth, sm.label_colors)
    assert np.allclose(sm.map, smooth, rtol=0, atol=1e-12)
    assert np.array_equal(color, ref_color)
    png = np.asarray(Image.open(os.path.join(sm.output_dir, "global_map.png")))
    assert np.array_equal(png[:, :, ::-1], color)                 # stored the way cv2.imwrite stores it
    ref = eo.iou(truth, eo.convert_labels(ref_color))
    assert "Overall Missing rate: {}".format(ref[2]) in out
    assert "IOU for road: {}".format(ref[0][0]) in out
