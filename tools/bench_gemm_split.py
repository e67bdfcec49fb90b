#!/usr/bin/env python3
"""Micro-benchmark of the multi-pass ("mixed" precision) ring GEMM on the network's layer shapes.
    python tools/bench_gemm_split.py [--reps 20]     (AVL_GEMM_DEEP=0/1 selects the ring depth of the 2-pass 256x256 kernel)"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from vision_semantic_segmentation_amd import _lib  # noqa: E402
from vision_semantic_segmentation_amd.network import AvlSegOp, OP_GEMM  # noqa: E402

SHAPES = [  # (name, M, K, N, A split, residual split, out split)
    ("layer4.conv1 x3", 32400, 2048, 1024, True, None, False), ("layer4.conv1 x2", 32400, 2048, 1024, False, None, False),
    ("layer4.conv3 x2", 32400, 1024, 2048, False, True, True), ("layer4.conv3 x3", 32400, 1024, 2048, True, True, True),
    ("layer4.ds x3", 32400, 1024, 2048, True, None, True),
    ("layer3.conv1 x3", 32400, 1024, 512, True, None, False), ("layer3.conv1 x2", 32400, 1024, 512, False, None, False),
    ("layer3.conv3 x2", 32400, 512, 1024, False, True, True), ("layer3.conv3 x3", 32400, 512, 1024, True, True, True),
    ("aspp.b0 x3", 32400, 2048, 256, True, None, True), ("aspp.proj x3", 32400, 1024, 256, True, None, True),
    ("layer2.conv3 x2", 32400, 256, 512, False, True, True), ("layer1.conv1 x3", 129600, 256, 128, True, None, False),
    ("layer1.conv3 x2", 129600, 128, 256, False, True, True), ("dec.pw0 x3", 128104, 512, 256, True, None, True),
]


def run(name, M, K, N, a_split, r_split, o_split, reps):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    Mp, Np = (M + 255) // 256 * 256, (N + 255) // 256 * 256
    a = torch.randn(2, Mp, K, device=dev).to(torch.float16)
    a[1] *= 2 ** -11
    nsub = 3 if a_split else 2
    w = (torch.randn(Np, K * nsub, device=dev) / K ** 0.5).to(torch.float16)
    b = torch.randn(Np, device=dev)
    r = torch.randn(2, Mp, N, device=dev).to(torch.float16)
    out = torch.zeros(2, Mp, N, device=dev, dtype=torch.float16)
    op = AvlSegOp()
    op.kind, op.dtype = OP_GEMM, _lib.AVL_F16
    op.in_, op.out, op.weight, op.bias = a[0].data_ptr(), out[0].data_ptr(), w.data_ptr(), b.data_ptr()
    op.in_h, op.in_w, op.in_c, op.in_ld, op.in_rows = 1, M, K, K, Mp
    op.out_h, op.out_w, op.out_c, op.out_ld, op.out_rows = 1, M, N, N, Mp
    op.relu, op.w_rows, op.ksize, op.stride, op.dil, op.groups, op.w_split = 1, Np, 1, 1, 1, 1, 1
    if a_split:
        op.in_lo = a[1].data_ptr()
    if r_split is not None:
        op.in2, op.in2_ld = r[0].data_ptr(), N
        if r_split:
            op.in2_lo = r[1].data_ptr()
    if o_split:
        op.out_lo = out[1].data_ptr()
    plan = C.c_void_p()
    _lib.check(_lib.lib().avl_seg_plan_create((AvlSegOp * 1)(op), 1, C.byref(plan)))
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        _lib.check(_lib.lib().avl_seg_plan_run(plan, s))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib.lib().avl_seg_plan_run(plan, s)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * M * K * N
    by = 2.0 * (Mp * K * (2 if a_split else 1) + N * K * nsub + M * N * ((2 if r_split else 1) if r_split is not None else 0) + M * N * (2 if o_split else 1))
    print("%-18s M=%6d K=%4d N=%4d  %8.1f us  %7.1f TF/s algorithmic  %7.1f TF/s executed  %6.0f GB/s" %
          (name, M, K, N, ms * 1e3, fl / ms / 1e9, nsub * fl / ms / 1e9, by / ms / 1e6), flush=True)
    _lib.lib().avl_seg_plan_destroy(plan)
    return ms


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    tot = sum(run(*sh, reps=a.reps) for sh in SHAPES)
    print("sum over shapes: %.4f ms" % tot)
