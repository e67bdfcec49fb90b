#!/usr/bin/env python3
"""Micro-benchmark + correctness check of the 1x1-conv GEMM on the network's layer shapes."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from vision_semantic_segmentation_amd import _lib  # noqa: E402
from vision_semantic_segmentation_amd.network import AvlSegOp, OP_GEMM  # noqa: E402

SHAPES = [  # (name, M, K, N, residual)
    ("layer4.conv1", 32400, 2048, 1024, False), ("layer4.conv3", 32400, 1024, 2048, True),
    ("layer4.ds", 32400, 1024, 2048, False), ("layer3.conv1", 32400, 1024, 512, False),
    ("layer3.conv3", 32400, 512, 1024, True), ("aspp.pw", 32400, 2048, 256, False),
    ("layer2.conv3", 32400, 256, 512, True), ("layer1.conv1", 129600, 256, 128, False),
    ("layer1.conv3", 129600, 128, 256, True), ("dec.pw0", 128104, 512, 256, False),
]


def run(name, M, K, N, res, variant, reps, check, pad=0):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    Mp = (M + 255) // 256 * 256
    Np = (N + 255) // 256 * 256
    a = torch.randn(Mp, K + pad, device=dev).to(torch.bfloat16)[:, :K]         # pad: row pitch of the activations (L2 channel experiment)
    if os.environ.get("AVL_ZERO"):          # zero operands draw less power: separates clock limits from schedule limits
        a.zero_()
    w = (torch.randn(Np, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(Np, device=dev)
    r = torch.randn(Mp, N, device=dev).to(torch.bfloat16) if res else None
    out = torch.zeros(Mp, N, device=dev, dtype=torch.bfloat16)
    op = AvlSegOp()
    op.kind, op.dtype = OP_GEMM, _lib.AVL_BF16
    op.in_, op.out, op.weight, op.bias = a.data_ptr(), out.data_ptr(), w.data_ptr(), b.data_ptr()
    op.in_h, op.in_w, op.in_c, op.in_ld, op.in_rows = 1, M, K, K + pad, Mp
    op.out_h, op.out_w, op.out_c, op.out_ld, op.out_rows = 1, M, N, N, Mp
    op.relu, op.w_rows, op.ksize, op.stride, op.dil, op.groups = 1, Np, 1, 1, 1, 1
    op.w_layout = variant
    if res:
        op.in2, op.in2_ld = r.data_ptr(), N
    plan = C.c_void_p()
    arr = (AvlSegOp * 1)(op)
    _lib.check(_lib.lib().avl_seg_plan_create(arr, 1, C.byref(plan)))
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
    err = -1.0
    if check:
        ref = a[:M].float() @ w[:N].float().t() + b[:N]
        if res:
            ref = ref + r[:M].float()
        ref = torch.relu(ref)
        err = float((out[:M].float() - ref).abs().max() / ref.abs().max())
    fl = 2.0 * M * K * N
    print("%-14s v%d M=%6d K=%4d N=%4d res=%d  %8.1f us  %7.1f TF/s  relerr %.2e" % (name, variant, M, K, N, res, ms * 1e3, fl / ms / 1e9, err))
    _lib.lib().avl_seg_plan_destroy(plan)
    return ms


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="1,0")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--pad", type=int, default=0, help="extra elements per activation row (in_ld = K + pad)")
    a = ap.parse_args()
    tot = {}
    for sh in SHAPES:
        for v in [int(x) for x in a.variants.split(",")]:
            if v == 5 and sh[3] % 256:          # k_gemm_w4 (one wave per SIMD, experiment): N % 256 == 0 only
                continue
            tot[v] = tot.get(v, 0.0) + run(*sh, variant=v, reps=a.reps, check=not a.no_check, pad=a.pad)
    print("sum over shapes (ms):", {k: round(v, 4) for k, v in tot.items()})
