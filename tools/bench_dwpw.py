#!/usr/bin/env python3
"""Times AVL_OP_DWPW alone at the frame's shapes: the three ASPP branches (k_dwpw_x: 135 x 240 x 2048 -> 256, dilation 12 / 24 / 36) and the
decoder's two refine blocks (k_dwpw_xs, split input: 270 x 480 x 512 -> 256 and 268 x 478 x 256 -> 256, pad 0).  Random operands,
`reps` launches between two events on the launch stream.  With a DW_EXP library (tools/ab_dwpw.sh) the numbers are phase ablations."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from vision_semantic_segmentation_amd import _lib  # noqa: E402
from vision_semantic_segmentation_amd.network import OP_DWPW, AvlSegOp, dwpw_block_order, dwpw_tile_order, pack_dw_f32, pack_split_rows  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
CASES = (("aspp d12", 135, 240, 2048, 256, 12, 12, False), ("aspp d36", 135, 240, 2048, 256, 36, 36, False),
         ("decoder.0", 270, 480, 512, 256, 1, 0, True), ("decoder.1", 268, 478, 256, 256, 1, 0, True), ("aspp d12 split in", 135, 240, 2048, 256, 12, 12, True),
         ("decoder.0, 246 tiles", 134, 240, 512, 256, 1, 0, True), ("decoder.0, 502 tiles", 270, 240, 512, 256, 1, 0, True))
for (name, H, W, K, N, d, pad, split_in) in CASES:
    g = torch.Generator().manual_seed(1)
    OH, OW = H + 2 * pad - 2 * d, W + 2 * pad - 2 * d
    M, Mi = OH * OW, H * W
    Mp, Mip, Np = (M + 255) // 256 * 256, (Mi + 255) // 256 * 256, 256
    x = torch.zeros((2, Mip, K), dtype=torch.float16)
    x[0, :Mi] = torch.randn((Mi, K), generator=g).to(torch.float16)
    x[1, :Mi] = (torch.randn((Mi, K), generator=g) * 2.0 ** -12).to(torch.float16)
    w1, b1 = torch.randn((K, 1, 3, 3), generator=g).double() * 0.3, torch.randn(K, generator=g).double() * 0.1
    w2 = torch.zeros((Np, K), dtype=torch.float64)
    w2[:N] = torch.randn((N, K), generator=g).double() / K ** 0.5
    b2 = torch.randn(Np, generator=g)
    xd, w2d, b2d = x.to(dev), pack_split_rows(w2, 2).to(dev), b2.to(dev)
    blocks = split_in and d == 1 and not os.environ.get("DWPW_ROW_TILES")       # 8 x 16-pixel tiles, as the network's decoder uses them
    params = torch.cat([pack_dw_f32(w1, b1), dwpw_block_order(OH, OW) if blocks else dwpw_tile_order(OH, OW, d)]).to(dev)
    out = torch.zeros((2, Mp, N), dtype=torch.float16, device=dev)
    op = AvlSegOp()
    op.kind, op.dtype = OP_DWPW, _lib.AVL_F16
    op.in_, op.in2, op.out, op.out_lo, op.weight, op.bias = xd[0].data_ptr(), params.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), w2d.data_ptr(), b2d.data_ptr()
    if split_in:
        op.in_lo = xd[1].data_ptr()
    op.in_h, op.in_w, op.in_c, op.in_ld, op.in_rows = H, W, K, K, Mip
    op.out_h, op.out_w, op.out_c, op.out_ld, op.out_rows = OH, OW, N, N, Mp
    op.relu, op.w_rows, op.ksize, op.stride, op.pad, op.dil, op.groups, op.w_split, op.w_layout = 1, Np, 3, 1, pad, d, K, 3, int(blocks)
    plan = C.c_void_p()
    _lib.check(_lib.lib().avl_seg_plan_create((AvlSegOp * 1)(op), 1, C.byref(plan)))
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        _lib.lib().avl_seg_plan_run(plan, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib.lib().avl_seg_plan_run(plan, s)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    ntiles = ((OH + 7) // 8) * ((OW + 15) // 16) if blocks else (M + 127) // 128
    nk, rounds = K // 64, -(-ntiles // 256)
    flop = 2.0 * M * K * (9 + N)
    print("%-22s %.4f ms  %6.1f TFLOP/s  %5.2f us = %5.0f cycles per K-step and tile (%d steps x %d rounds, 2.4 GHz)"
          % (name, ms, flop / ms * 1e-9, ms * 1e3 / (nk * rounds), ms * 1e3 / (nk * rounds) * 2400, nk, rounds))
    _lib.lib().avl_seg_plan_destroy(plan)
