"""End-of-run map evaluation on the GPU -- counterpart of the reference's test/test_semantic_mapping.py
(`convert_labels` :6-19, `Test` :29-161), which `SemanticMapping.mapping` calls once at shutdown when a ground-truth
directory is configured (src/mapping.py:341-344).  Same names, arguments and printed lines; the per-pixel work
(colour -> label, joint ground-truth/label histogram) is one HIP kernel (`avl_eval_map`), the IoU / accuracy /
missing-rate ratios are formed from its integer counts with the reference's own float expressions.

NumPy in -> NumPy out (float64 label maps like the reference); CUDA tensors in -> results stay on the GPU.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib

CLASS_NAMES = {0: "road", 1: "crosswalk", 2: "lane"}          # test_semantic_mapping.py:77
CLASS_LISTS = [1, 2, 3]                                        # :78


def _dev_u8(a, device=None):
    if isinstance(a, torch.Tensor):
        return a.to(torch.uint8).contiguous() if a.dtype != torch.uint8 else a.contiguous()
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint8)).to(device or "cuda")


def _bin_truth(gmap):
    """ground truth of any numeric type -> uint8 bins the kernel counts: 1..6 exact, 7 = any other positive value
    (never equal to a generated label, but counted by `gmap > 0`), 0 = zero / negative."""
    if isinstance(gmap, torch.Tensor):
        g = gmap
        out = torch.zeros(g.shape, dtype=torch.uint8, device=g.device)
        pos = g > 0
        out[pos] = 7
        for v in range(1, 7):
            out[g == v] = v
        return out
    g = np.asarray(gmap)
    out = np.zeros(g.shape, dtype=np.uint8)
    out[g > 0] = 7
    for v in range(1, 7):
        out[g == v] = v
    return out


def _run(color, mask, gt):
    """-> (labels uint8 CUDA [H,W], counts int64 numpy [8,8] or None)"""
    color = _dev_u8(color)
    if color.dim() != 3 or color.shape[2] != 3:
        raise ValueError("colour map must be [H, W, 3]")
    dev = color.device
    h, w = int(color.shape[0]), int(color.shape[1])
    mask_t = gt_t = counts = None
    if mask is not None:
        m = mask.to(dev) if isinstance(mask, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(mask)).to(dev)
        mask_t = (m != 0).to(torch.uint8).contiguous()
        if mask_t.shape[0] < h or mask_t.shape[1] < w:
            raise ValueError("mask %s is smaller than the map %s" % (tuple(mask_t.shape), (h, w)))
    if gt is not None:
        gt_t = _dev_u8(gt, dev)
        if tuple(gt_t.shape) != (h, w):
            # the reference slices truth[shift_w:H+shift_w, shift_h:W+shift_h]; a short slice fails at `gmap_layer * map_layer`
            raise ValueError("operands could not be broadcast together with shapes %s %s" % (tuple(gt_t.shape), (h, w)))
        counts = torch.zeros(64, dtype=torch.int64, device=dev)
    labels = torch.empty((h, w), dtype=torch.uint8, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    _lib.check(_lib.lib().avl_eval_map(p(color), h, w, p(mask_t), int(mask_t.shape[1]) if mask_t is not None else 0, p(gt_t),
                                       w if gt_t is not None else 0, p(labels), p(counts),
                                       C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "avl_eval_map")
    return labels, (counts.cpu().numpy().reshape(8, 8) if counts is not None else None)


def convert_labels(gmap, mask=None):
    """test_semantic_mapping.py:6-19 -- covert colors to labels."""
    labels, _ = _run(gmap, mask, None)
    if isinstance(gmap, torch.Tensor):
        return labels
    return labels.cpu().numpy().astype(np.float64)


def read_img(global_map_path, mask=None):
    """:22-27.  The reference reads with cv2.imread (BGR order of a file that cv2.imwrite wrote from the RGB-valued
    array, i.e. the array comes back as it was rendered); without OpenCV the PNG is read with PIL and reversed to that order."""
    from PIL import Image
    gmap = np.asarray(Image.open(global_map_path).convert("RGB"))[:, :, ::-1].copy()
    return gmap, convert_labels(gmap, mask)


def stats_from_counts(J, class_lists=CLASS_LISTS):
    """Test.iou's ratios (:136-148) from the joint histogram J[gt][label] -- the same float expressions on the same
    integer sums, so the results are bit-equal to the reference's."""
    J = np.asarray(J, dtype=np.int64)
    iou_lists, acc_lists = [], []
    for cls in class_lists:
        intersection = float(J[cls, cls])
        n_gt, n_map = J[cls, :].sum(), J[:, cls].sum()
        union = float(n_gt + n_map - intersection)
        iou_lists.append(intersection / union)                  # ZeroDivisionError for an absent class, as in the reference
        with np.errstate(divide="ignore", invalid="ignore"):
            acc_lists.append(intersection / n_gt)
    n_pos = J[1:, :].sum()
    with np.errstate(divide="ignore", invalid="ignore"):
        miss = 1 - J[1:, 1:].sum() / n_pos
        accuracy = sum(J[v, v] for v in range(1, 6)) / n_pos
    return iou_lists, acc_lists, miss, accuracy


class Test(object):
    def __init__(self, ground_truth_dir="./", shift_h=0, shift_w=0, logger=None):
        """:30-81.  Loads <ground_truth_dir>/truth.npy (and mask.npy when present).  The reference can also build
        truth.npy from four BEV JPEGs with OpenCV (:44-66); OpenCV is not available offline, so that branch raises."""
        truth_file_path = os.path.join(ground_truth_dir, "truth.npy")
        if not os.path.exists(truth_file_path):
            raise FileNotFoundError("%s does not exist (building it from the bev-5cm-*.jpg images needs OpenCV)" % truth_file_path)
        self.ground_truth_mask = np.load(truth_file_path, allow_pickle=False)
        mask_path = os.path.join(ground_truth_dir, "mask.npy")
        self.mask = np.load(mask_path, allow_pickle=False) if os.path.exists(mask_path) else None
        self.d = dict(CLASS_NAMES)
        self.class_lists = list(CLASS_LISTS)
        self.shift_w = shift_w
        self.shift_h = shift_h
        self.logger = logger
        self._truth_bins = None

    def _print(self, text):
        print(text)                                   # the reference prints (its logger argument is stored, never used)

    def _truth_window(self, shape):
        """:125-126 -- ground_truth_mask[shift_w:H+shift_w, shift_h:W+shift_h]"""
        return self.ground_truth_mask[self.shift_w:shape[0] + self.shift_w, self.shift_h:shape[1] + self.shift_h]

    def test_single_map(self, global_map):
        """:119-127 -- IoU, accuracy and missing rate of the rendered global map against the ground truth.
        Colour conversion and counting run as ONE kernel over the colour map."""
        h, w = int(global_map.shape[0]), int(global_map.shape[1])
        gmap = self._truth_window((h, w))
        _, J = _run(global_map, None, _bin_truth(gmap))
        return self._report(J, latex_mode=False, verbose=True)

    def iou(self, gmap, generate_map, latex_mode=False, verbose=False):
        """:128-161 -- gmap: ground-truth label map, generate_map: generated label map (both integer-valued)."""
        g, m = _bin_truth(gmap), _bin_truth(generate_map)
        if tuple(g.shape) != tuple(m.shape):
            raise ValueError("operands could not be broadcast together with shapes %s %s" % (tuple(g.shape), tuple(m.shape)))
        # the kernel recognises colours; feed it the label map through its palette so the same pass does the counting
        pal = np.zeros((8, 3), dtype=np.uint8)
        pal[1:6] = [(128, 64, 128), (140, 140, 200), (255, 255, 255), (244, 35, 232), (107, 142, 35)]
        if int((m >= 6).sum()):
            raise ValueError("generated label map holds values outside 0..5")
        color = torch.from_numpy(pal).to(m.device)[m.long()] if isinstance(m, torch.Tensor) else pal[m]
        _, J = _run(color, None, g)
        return self._report(J, latex_mode=latex_mode, verbose=verbose)

    def _report(self, J, latex_mode, verbose):
        iou_lists, acc_lists, miss, accuracy = stats_from_counts(J, self.class_lists)
        if verbose:
            if not latex_mode:
                self._print("IOU for {}: {}\t{}: {}\t{}:{}\tmIOU: {}".format(self.d[0], iou_lists[0], self.d[1], iou_lists[1],
                                                                          self.d[2], iou_lists[2], np.mean(iou_lists)))
                self._print("Accuracy for {}: {}\t{}: {}\t{}:{}\tmean Accuracy: {}".format(self.d[0], acc_lists[0], self.d[1],
                                                                                        acc_lists[1], self.d[2], acc_lists[2], accuracy))
                self._print("Overall Missing rate: {}".format(miss))
            else:
                miss_percent = miss * 100
                self._print(f"&{iou_lists[0]:.3f}&{iou_lists[1]:.3f}&{iou_lists[2]:.3f}&{np.mean(iou_lists):.3f}&{miss_percent:.3g}\\\\ \\hline")
        return iou_lists, miss

    def full_test(self, dir_path="./global_maps", visualize=False, latex_mode=False, verbose=False):
        """:83-117 -- every *.png of dir_path; prints the batch averages.  (visualize needs matplotlib windows: ignored.)"""
        paths = [os.path.join(dir_path, x) for x in os.listdir(dir_path) if ".png" in x]
        iou_array, miss_array = [], []
        for path in paths:
            self._print("You are testing\t" + path.split("/")[-1])
            _, generate_map = read_img(path, self.mask)
            gmap = self._truth_window(generate_map.shape)
            iou_lists, miss = self.iou(gmap, generate_map, latex_mode=latex_mode, verbose=verbose)
            iou_array.append(np.array(iou_lists).reshape(1, -1))
            miss_array.append(miss)
        miss = np.mean(miss_array)
        iou_lists = np.mean(np.concatenate(iou_array, axis=0), axis=0)
        self._print("Final Batch evaluation")
        self._print("IOU for {}: {}\t{}: {}\t{}:{}\tmIOU: {}".format(self.d[0], iou_lists[0], self.d[1], iou_lists[1], self.d[2],
                                                                  iou_lists[2], np.mean(iou_lists)))
        self._print("Overall Missing rate: {}".format(miss))
        if latex_mode:
            self._print(f"&{iou_lists[0]:.3f}&{iou_lists[1]:.3f}&{iou_lists[2]:.3f}&{np.mean(iou_lists):.3f}&{miss * 100:.3g}\\\\ \\hline")
        return iou_lists, miss
