"""Times AVL_OP_BOTTLENECK alone at layer1's 1080p shape (270 x 480 pixels): the three block variants the network runs.

    python tools/bench_bottleneck.py [--reps 50]
prints microseconds per launch (HIP events around `reps` back-to-back launches), the algorithmic TFLOP/s and GB/s."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vision_semantic_segmentation_amd import _lib  # noqa: E402
from vision_semantic_segmentation_amd.network import OP_BOTTLENECK, AvlSegOp, pack_bottleneck  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--hw", type=int, nargs=2, default=(270, 480))
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    H, W = a.hw
    rows = (H * W + 255) // 256 * 256
    g = torch.Generator().manual_seed(0)
    for name, cin, ds, t1lo, xlo, olo in (("layer1.0 (64 -> 256, downsample folded, t1 hi + lo)", 64, 1, 1, 0, 0),
                                          ("layer1.0 (64 -> 256, downsample folded, t1 hi)", 64, 1, 0, 0, 0),
                                          ("layer1.1 (256 -> 256, single planes)", 256, 0, 0, 0, 0),
                                          ("layer1.2 (256 -> 256, split output)", 256, 0, 0, 0, 1),
                                          ("layer1.x (256 -> 256, split input and output)", 256, 0, 0, 1, 1)):
        x = (torch.randn((2, rows, cin), generator=g) * 0.5).to(torch.float16).to(dev)
        y = torch.zeros((2, rows, 256), dtype=torch.float16, device=dev)
        w1 = torch.randn((128, cin), generator=g, dtype=torch.float64) * (2.0 / cin) ** 0.5
        w2 = torch.randn((128, 4, 3, 3), generator=g, dtype=torch.float64) * (2.0 / 36) ** 0.5
        w3 = torch.randn((256, 128), generator=g, dtype=torch.float64) * (1.0 / 128) ** 0.5
        wd = torch.randn((256, cin), generator=g, dtype=torch.float64) * (1.0 / cin) ** 0.5 if ds else None
        p1, p2, p3 = (t.to(dev) for t in pack_bottleneck(w1, w2, w3, wd, 32))
        b = (torch.randn(512, generator=g) * 0.1).to(dev)
        op = AvlSegOp()
        op.kind, op.dtype = OP_BOTTLENECK, _lib.AVL_F16
        op.in_, op.out, op.weight, op.in2, op.in3, op.bias = x[0].data_ptr(), y[0].data_ptr(), p1.data_ptr(), p2.data_ptr(), p3.data_ptr(), b.data_ptr()
        op.in_lo = x[1].data_ptr() if xlo else 0
        op.out_lo = y[1].data_ptr() if olo else 0
        op.in_h, op.in_w, op.in_c, op.in_ld, op.in_rows = H, W, cin, cin, rows
        op.out_h, op.out_w, op.out_c, op.out_ld, op.out_rows = H, W, 256, 256, rows
        op.in3_c, op.ksize, op.stride, op.pad, op.dil, op.groups, op.relu, op.w_layout, op.w_split = 128, 3, 1, 1, 1, 32, 1, ds, t1lo
        plan = C.c_void_p()
        _lib.check(_lib.lib().avl_seg_plan_create((AvlSegOp * 1)(op), 1, C.byref(plan)), "avl_seg_plan_create")
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(5):
            _lib.check(_lib.lib().avl_seg_plan_run(plan, s), "run")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            _lib.check(_lib.lib().avl_seg_plan_run(plan, s), "run")
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.reps
        macs = cin * 128 + 128 * 4 * 9 + 128 * 256 + (cin * 256 if ds else 0)
        fl = 2.0 * H * W * macs
        by = H * W * 2.0 * (cin * (2 if xlo else 1) + 256 * (2 if olo else 1))
        print("%-58s %7.1f us   %6.1f TFLOP/s algorithmic   %6.0f GB/s (in + out planes)" % (name, us, fl / us * 1e-6, by / us * 1e-3))
        _lib.lib().avl_seg_plan_destroy(plan)


if __name__ == "__main__":
    main()
