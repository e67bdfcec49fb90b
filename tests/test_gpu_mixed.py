"""MODEL.PRECISION = "mixed": f16 MFMA with split (hi + lo) operands.

A split tensor is two float16 planes whose sum carries ~22 significant bits; split weights are the pair
hi = f16(w), lo = f16(w - hi).  Every kernel that takes them is run ALONE through avl_seg_plan_* and compared with a
float64 evaluation of the same operator on the values the planes actually hold (hi + lo), so only the fp32
accumulation order and the dropped lo x lo products (2^-22) differ: tolerance 3e-6 of max|ref|.
Then the whole network: logits within 1e-3 of the torch-CPU fp32 oracle (north_star's bar) -- the mode bench.py times."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 3e-6


def _run_plan(ops):
    from test_gpu_ops import _run_plan as run
    run(ops)


def _split(x64):
    import torch
    hi = x64.to(torch.float16)
    lo = (x64 - hi.double()).to(torch.float16)
    return hi, lo


def _pad_rows(t, rows):
    import torch
    out = torch.zeros((rows,) + tuple(t.shape[1:]), dtype=t.dtype)
    out[:t.shape[0]] = t
    return out


@pytest.mark.parametrize("case", [  # (M, K, N, A split, residual, residual split, out split, relu)
    (2600, 128, 256, False, False, False, False, True),       # conv1-like, weights split only (2 passes), 256x128 ring
    (2600, 128, 256, True, False, False, True, True),         # 3 passes, split output
    (70000, 256, 512, False, True, True, True, True),         # conv3-like on the 256x256 ring: split residual + output
    (70000, 512, 256, True, True, True, True, False),         # downsample-like (no ReLU), 3 passes
    (4097, 1024, 512, True, True, False, True, True),         # single-plane residual, split output
    (300, 2048, 1024, False, True, True, False, True),        # split residual, single-plane output
    (777, 256, 19, True, False, False, False, False),         # the classifier: 3 passes in the small-N kernel, fp32 out
    (1000, 64, 128, True, False, False, True, True),          # K = one block
])
def test_split_gemm(case, cuda_device):
    import torch
    from vision_semantic_segmentation_amd import _lib
    from vision_semantic_segmentation_amd.network import OP_GEMM, AvlSegOp, pack_split_rows
    M, K, N, a_split, res, r_split, o_split, relu = case
    g = torch.Generator().manual_seed(M + K + N + int(a_split))
    Mp, Np = (M + 255) // 256 * 256, (N + 255) // 256 * 256
    a64 = torch.randn((M, K), generator=g, dtype=torch.float64)
    a_hi, a_lo = _split(a64)
    w64 = torch.zeros((Np, K), dtype=torch.float64)
    w64[:N] = torch.randn((N, K), generator=g, dtype=torch.float64) / K ** 0.5
    w_hi, w_lo = _split(w64)
    b = torch.zeros(Np)
    b[:N] = torch.randn(N, generator=g)
    out_f32 = N == 19
    a_val = a_hi.double() + (a_lo.double() if a_split else 0)
    ref = a_val @ (w_hi.double() + w_lo.double())[:N].t() + b[:N].double()
    if res:
        r64 = torch.randn((M, N), generator=g, dtype=torch.float64)
        r_hi, r_lo = _split(r64)
        ref = ref + r_hi.double() + (r_lo.double() if r_split else 0)
    if relu:
        ref = torch.relu(ref)
    planes = torch.stack([_pad_rows(a_hi, Mp), _pad_rows(a_lo, Mp)]).to(cuda_device)
    wd = pack_split_rows(w64, 3 if a_split else 2).to(cuda_device)
    bd = b.to(cuda_device)
    out = torch.full((2, Mp, N), 7.0, dtype=torch.float32 if out_f32 else torch.float16, device=cuda_device)
    op = AvlSegOp()
    op.kind, op.dtype = OP_GEMM, _lib.AVL_F16
    op.in_, op.out, op.weight, op.bias = planes[0].data_ptr(), out[0].data_ptr(), wd.data_ptr(), bd.data_ptr()
    op.in_h, op.in_w, op.in_c, op.in_ld, op.in_rows = 1, M, K, K, Mp
    op.out_h, op.out_w, op.out_c, op.out_ld, op.out_rows = 1, M, N, N, Mp
    op.relu, op.out_f32, op.w_rows, op.ksize, op.stride, op.dil, op.groups = int(relu), int(out_f32), Np, 1, 1, 1, 1
    op.w_split = 1
    if a_split:
        op.in_lo = planes[1].data_ptr()
    if o_split:
        op.out_lo = out[1].data_ptr()
    if res:
        rd = torch.stack([_pad_rows(r_hi, Mp), _pad_rows(r_lo, Mp)]).to(cuda_device)
        op.in2, op.in2_ld = rd[0].data_ptr(), N
        if r_split:
            op.in2_lo = rd[1].data_ptr()
    _run_plan([op])
    got = out[0, :M].cpu().double() + (out[1, :M].cpu().double() if o_split else 0)
    err = float((got - ref).abs().max() / ref.abs().max())
    bar = TOL if (o_split or out_f32) else 2 ** -11 * 1.5              # a single f16 output plane rounds to 11 bits
    assert err <= bar, "split gemm %s: %.3e" % (case, err)
    assert torch.all(out[0, M:] == 7.0)
    if o_split:
        assert torch.all(out[1, M:] == 7.0)
        # the planes are a proper split: hi is the rounded value, |lo| <= half an ulp of hi
        assert float((out[1, :M].float().abs() - out[0, :M].float().abs() * 2 ** -11).max()) <= 1e-7
    else:
        assert torch.all(out[1] == 7.0)                                  # the low plane is not touched


@pytest.mark.parametrize("case", [(23, 45, 128, 1, 1), (23, 45, 256, 2, 1), (30, 41, 512, 1, 2), (19, 67, 1024, 1, 4)])
@pytest.mark.parametrize("out_split", [False, True])
def test_split_grouped_conv(case, out_split, cuda_device):
    import torch
    import torch.nn.functional as F
    from test_gpu_ops import _from_rows, _nhwc_rows, _spatial_op
    from vision_semantic_segmentation_amd import _lib
    from vision_semantic_segmentation_amd.network import OP_GCONV, pack_gconv_windows
    H, W, width, s, d = case
    G = 32
    cg = width // G
    g = torch.Generator().manual_seed(H * 77 + W + width)
    x = torch.randn((1, width, H, W), generator=g).to(torch.float16)
    w64 = torch.randn((width, cg, 3, 3), generator=g, dtype=torch.float64) * (2.0 / (cg * 9)) ** 0.5
    b = torch.randn(width, generator=g) * 0.1
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    w_hi, w_lo = _split(w64)
    ref = F.relu(F.conv2d(x.double(), w_hi.double() + w_lo.double(), b.double(), stride=s, padding=d, dilation=d, groups=G))
    src = _nhwc_rows(x).to(cuda_device)
    dst = torch.full((2, (OH * OW + 255) // 256 * 256, width), 7.0, dtype=torch.float16, device=cuda_device)
    nwin = width // 32
    wd = torch.cat([pack_gconv_windows(w_hi.double(), G).reshape(nwin, 2, 9, 16, 32),
                    pack_gconv_windows(w_lo.double(), G).reshape(nwin, 2, 9, 16, 32)], dim=2).reshape(-1).to(torch.float16).to(cuda_device)
    bd = b.to(cuda_device)
    _run_plan([_spatial_op(OP_GCONV, _lib.AVL_F16, src, (H, W), width, dst[0], (OH, OW), width, weight=wd.data_ptr(), bias=bd.data_ptr(),
                           ksize=3, stride=s, pad=d, dil=d, groups=G, relu=1, w_layout=1, w_split=1,
                           out_lo=dst[1].data_ptr() if out_split else 0)])
    got = _from_rows(dst[0].cpu().double(), OH, OW, width)
    if out_split:
        got = got + _from_rows(dst[1].cpu().double(), OH, OW, width)
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err <= (TOL if out_split else 2 ** -11 * 1.5), "split grouped conv %s: %.3e" % (case, err)
    assert torch.all(dst[0, OH * OW:] == 7.0)


@pytest.mark.parametrize("case", [(37, 53, 128, 256, 12), (20, 31, 2048, 256, 24), (16, 16, 256, 512, 1)])
def test_split_fused_depthwise_pointwise(case, cuda_device):
    """AVL_OP_DWPW with split 1x1 weights and a split output (the ASPP branches of the mixed mode): the depthwise slice
    is still rounded to f16 (as in the f16 mode), the 1x1 runs two passes and nothing is rounded away at the output."""
    import torch
    import torch.nn.functional as F
    from test_gpu_ops import _from_rows, _nhwc_rows
    from vision_semantic_segmentation_amd import _lib
    from vision_semantic_segmentation_amd.network import OP_DWPW, AvlSegOp, dwpw_tile_order, pack_dw_pairs, pack_split_rows
    H, W, K, N, d = case
    g = torch.Generator().manual_seed(H * 31 + W + K + d)
    x = torch.randn((1, K, H, W), generator=g).to(torch.float16)
    w1 = torch.randn((K, 1, 3, 3), generator=g) * 0.3
    b1 = torch.randn(K, generator=g) * 0.1
    w2 = torch.randn((N, K), generator=g, dtype=torch.float64) / K ** 0.5
    b2 = torch.randn(N, generator=g) * 0.1
    M = H * W
    Mp, Np = (M + 255) // 256 * 256, (N + 255) // 256 * 256
    src = _nhwc_rows(x).to(cuda_device)
    w2p = torch.zeros((Np, K), dtype=torch.float64)
    w2p[:N] = w2
    b2p = torch.zeros(Np)
    b2p[:N] = b2
    w2d, b2d = pack_split_rows(w2p, 2).to(cuda_device), b2p.to(cuda_device)
    out = torch.full((2, Mp, N), 7.0, dtype=torch.float16, device=cuda_device)
    params = torch.cat([pack_dw_pairs(w1.double(), b1.double(), torch.float16), dwpw_tile_order(H, W, d)]).to(cuda_device)
    op = AvlSegOp()
    op.kind, op.dtype = OP_DWPW, _lib.AVL_F16
    op.in_, op.in2, op.out, op.out_lo = src.data_ptr(), params.data_ptr(), out[0].data_ptr(), out[1].data_ptr()
    op.weight, op.bias, op.w_split = w2d.data_ptr(), b2d.data_ptr(), 1
    op.in_h, op.in_w, op.in_c, op.in_ld, op.in_rows = H, W, K, K, src.shape[0]
    op.out_h, op.out_w, op.out_c, op.out_ld, op.out_rows = H, W, N, N, Mp
    op.relu, op.w_rows, op.ksize, op.stride, op.pad, op.dil, op.groups = 1, Np, 3, 1, d, d, K
    _run_plan([op])
    a = F.relu(F.conv2d(x.float(), w1.to(torch.float16).float(), b1, padding=d, dilation=d, groups=K)).to(torch.float16).double()
    w_hi, w_lo = _split(w2)
    ref = F.relu(F.conv2d(a, (w_hi.double() + w_lo.double()).view(N, K, 1, 1), b2.double()))
    got = _from_rows(out[0].cpu().double() + out[1].cpu().double(), H, W, N)
    # the depthwise slice goes through fp32 dot2 chains before its f16 rounding: a one-ulp flip of an `a` element
    # (2^-11 relative, one of K terms of comparable size) moves an output by ~2^-11 / sqrt(K)
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err <= 4 * 2 ** -11 / K ** 0.5, "split dwpw %s: %.3e" % (case, err)
    assert torch.all(out[:, M:] == 7.0)


@pytest.mark.parametrize("case", [(37, 53, 128, 256, 12), (20, 31, 2048, 256, 24), (16, 16, 256, 512, 1), (45, 80, 2048, 256, 36)])
def test_exact_fused_depthwise_pointwise(case, cuda_device):
    """AVL_OP_DWPW with w_split = 3 (k_dwpw_x, the ASPP branches of the default mixed mode): fp32 depthwise weights, the
    depthwise result as an f16 hi + lo tile, three MFMA passes -- against a float64 evaluation of the SAME operands (the f16
    input plane; the fp32 depthwise weights; the 1x1 weights hi + lo): nothing but fp32 accumulation is rounded away."""
    import torch
    import torch.nn.functional as F
    from test_gpu_ops import _from_rows, _nhwc_rows
    from vision_semantic_segmentation_amd import _lib
    from vision_semantic_segmentation_amd.network import OP_DWPW, AvlSegOp, dwpw_tile_order, pack_dw_f32, pack_split_rows, split_f16
    H, W, K, N, d = case
    g = torch.Generator().manual_seed(H * 31 + W + K + d + 1)
    x = torch.randn((1, K, H, W), generator=g).to(torch.float16)
    w1 = (torch.randn((K, 1, 3, 3), generator=g) * 0.3).double()
    b1 = (torch.randn(K, generator=g) * 0.1).double()
    w2 = torch.randn((N, K), generator=g, dtype=torch.float64) / K ** 0.5
    b2 = torch.randn(N, generator=g) * 0.1
    M = H * W
    Mp, Np = (M + 255) // 256 * 256, (N + 255) // 256 * 256
    src = _nhwc_rows(x).to(cuda_device)
    w2p = torch.zeros((Np, K), dtype=torch.float64)
    w2p[:N] = w2
    b2p = torch.zeros(Np)
    b2p[:N] = b2
    w2d, b2d = pack_split_rows(w2p, 2).to(cuda_device), b2p.to(cuda_device)
    out = torch.full((2, Mp, N), 7.0, dtype=torch.float16, device=cuda_device)
    params = torch.cat([pack_dw_f32(w1, b1), dwpw_tile_order(H, W, d)]).to(cuda_device)
    op = AvlSegOp()
    op.kind, op.dtype = OP_DWPW, _lib.AVL_F16
    op.in_, op.in2, op.out, op.out_lo = src.data_ptr(), params.data_ptr(), out[0].data_ptr(), out[1].data_ptr()
    op.weight, op.bias, op.w_split = w2d.data_ptr(), b2d.data_ptr(), 3
    op.in_h, op.in_w, op.in_c, op.in_ld, op.in_rows = H, W, K, K, src.shape[0]
    op.out_h, op.out_w, op.out_c, op.out_ld, op.out_rows = H, W, N, N, Mp
    op.relu, op.w_rows, op.ksize, op.stride, op.pad, op.dil, op.groups = 1, Np, 3, 1, d, d, K
    _run_plan([op])
    w1s = w1.float().double()                                                # what the kernel multiplies with: the fp32 weights
    b1s = b1.to(torch.float32).double()
    a64 = F.relu(F.conv2d(x.double(), w1s, b1s, padding=d, dilation=d, groups=K))
    ah, al = _split(a64)                                                      # the tile the kernel keeps: f16 hi + f16 lo
    w_hi, w_lo = _split(w2)
    ref = F.relu(F.conv2d(ah.double() + al.double(), (w_hi.double() + w_lo.double()).view(N, K, 1, 1), b2.double()))
    got = _from_rows(out[0].cpu().double() + out[1].cpu().double(), H, W, N)
    err = float((got - ref).abs().max() / ref.abs().max())
    print("exact dwpw %s: %.3e" % (case, err))
    # fp32 accumulation of the depthwise sums before the hi / lo split (2^-24 relative each) and of K products: TOL-sized
    assert err <= 2 * TOL, "exact dwpw %s: %.3e" % (case, err)
    assert torch.all(out[:, M:] == 7.0)


@pytest.mark.parametrize("case", [(37, 53, 512, 256, 1, 0, 0), (37, 53, 512, 256, 1, 0, 1), (20, 31, 256, 256, 1, 0, 1), (30, 40, 1024, 256, 12, 12, 0), (16, 16, 64, 512, 1, 1, 0),
                                  (16, 16, 64, 512, 1, 1, 1), (9, 140, 128, 64, 2, 0, 0), (11, 70, 128, 64, 1, 0, 1)])
def test_exact_fused_depthwise_pointwise_with_a_split_input(case, cuda_device):
    """round 5: AVL_OP_DWPW with w_split = 3 AND in_lo (k_dwpw_xs: the mixed decoder's refine blocks decoder.py:33-43 -- pad 0 --, the
    split16 plan's ASPP branches): input hi + lo planes, everything else as
    k_dwpw_x (fp32 depthwise weights, one v_fma_mix_f32 per tap, channel and plane) -- against a float64 evaluation of the same operands.
    Cases (last number: w_layout, 1 = 8 x 16-pixel tiles as the decoder
    uses them, block counts that do not divide the output): the decoder's two shapes, a dilated padded branch, two output-channel
    tiles, a tile spanning several image rows with K = 2 steps."""
    import torch
    import torch.nn.functional as F
    from test_gpu_ops import _from_rows, _nhwc_rows
    from vision_semantic_segmentation_amd import _lib
    from vision_semantic_segmentation_amd.network import (OP_DWPW, AvlSegOp, dwpw_block_order, dwpw_tile_order, pack_dw_f32, pack_split_rows,
                                                          split_f16)
    H, W, K, N, d, pad, blocks = case
    OH, OW = H + 2 * pad - 2 * d, W + 2 * pad - 2 * d
    g = torch.Generator().manual_seed(H * 31 + W + K + d + 7)
    x = torch.randn((1, K, H, W), generator=g, dtype=torch.float64)
    xh, xl = _split(x)
    w1 = (torch.randn((K, 1, 3, 3), generator=g) * 0.3).double()
    b1 = (torch.randn(K, generator=g) * 0.1).double()
    w2 = torch.randn((N, K), generator=g, dtype=torch.float64) / K ** 0.5
    b2 = torch.randn(N, generator=g) * 0.1
    M = OH * OW
    Mp, Np = (M + 255) // 256 * 256, (N + 255) // 256 * 256
    src = torch.stack([_nhwc_rows(xh), _nhwc_rows(xl)]).to(cuda_device)
    w2p = torch.zeros((Np, K), dtype=torch.float64)
    w2p[:N] = w2
    b2p = torch.zeros(Np)
    b2p[:N] = b2
    w2d, b2d = pack_split_rows(w2p, 2).to(cuda_device), b2p.to(cuda_device)
    out = torch.full((2, Mp, N), 7.0, dtype=torch.float16, device=cuda_device)
    params = torch.cat([pack_dw_f32(w1, b1), dwpw_block_order(OH, OW) if blocks else dwpw_tile_order(OH, OW, d)]).to(cuda_device)
    op = AvlSegOp()
    op.kind, op.dtype = OP_DWPW, _lib.AVL_F16
    op.in_, op.in_lo, op.in2, op.out, op.out_lo = src[0].data_ptr(), src[1].data_ptr(), params.data_ptr(), out[0].data_ptr(), out[1].data_ptr()
    op.weight, op.bias, op.w_split, op.w_layout = w2d.data_ptr(), b2d.data_ptr(), 3, blocks
    op.in_h, op.in_w, op.in_c, op.in_ld, op.in_rows = H, W, K, K, src.shape[1]
    op.out_h, op.out_w, op.out_c, op.out_ld, op.out_rows = OH, OW, N, N, Mp
    op.relu, op.w_rows, op.ksize, op.stride, op.pad, op.dil, op.groups = 1, Np, 3, 1, pad, d, K
    _run_plan([op])
    w1s = w1.float().double()                                                # what the kernel multiplies with
    b1s = b1.to(torch.float32).double()
    a64 = F.relu(F.conv2d(xh.double() + xl.double(), w1s, b1s, padding=pad, dilation=d, groups=K))
    ah, al = _split(a64)
    w_hi, w_lo = _split(w2)
    ref = F.relu(F.conv2d(ah.double() + al.double(), (w_hi.double() + w_lo.double()).view(N, K, 1, 1), b2.double()))
    got = _from_rows(out[0].cpu().double() + out[1].cpu().double(), OH, OW, N)
    err = float((got - ref).abs().max() / ref.abs().max())
    print("exact dwpw, split input %s: %.3e" % (case, err))
    assert err <= 2 * TOL, "exact dwpw, split input %s: %.3e" % (case, err)
    assert torch.all(out[:, M:] == 7.0)
    # and the lo plane matters: the same launch on the hi plane alone is off by the input's rounding
    a_hi_only = F.relu(F.conv2d(xh.double(), w1s, b1s, padding=pad, dilation=d, groups=K))
    ref_hi = F.relu(F.conv2d(a_hi_only, (w_hi.double() + w_lo.double()).view(N, K, 1, 1), b2.double()))
    assert float((got - ref_hi).abs().max() / ref.abs().max()) > 10 * err


@pytest.mark.parametrize("case", [(37, 53, 256, 19, 1), (20, 31, 512, 19, 0), (11, 70, 64, 32, 1), (9, 140, 128, 5, 0)])
def test_fused_depthwise_pointwise_with_the_classifier_in_its_epilogue(case, cuda_device):
    """round 5: AVL_OP_DWPW with out_f32 (k_dwpw_xs<CLS>): the decoder's last refine block (decoder.py:38-41), the classifier (decoder.py:42-43:
    1x1 conv with bias, no BN / ReLU) and torch.argmax (semantic_segmentation.py:56) in ONE launch -- the block's 256-channel result exists only
    in LDS, as f16 hi + lo tiles.  Logits against a float64 evaluation of the same operands; labels = the arg-max of the logits the kernel wrote,
    first maximal index on ties (classes 1 and 3 get identical weights and win everywhere they can); with five classes every logit is negative, so
    a lane that holds no real class (classes 8 .., 12 ..) would win with its padding zeros if it took part."""
    import torch
    import torch.nn.functional as F
    from test_gpu_ops import _nhwc_rows
    from vision_semantic_segmentation_amd import _lib
    from vision_semantic_segmentation_amd.network import OP_DWPW, AvlSegOp, dwpw_block_order, dwpw_tile_order, pack_dw_f32, pack_split_rows, split_f16
    H, W, K, ncls, blocks = case
    N, d, pad = 256, 1, 0
    OH, OW = H - 2, W - 2
    g = torch.Generator().manual_seed(H * 31 + W + K + ncls)
    x = torch.randn((1, K, H, W), generator=g, dtype=torch.float64)
    xh, xl = _split(x)
    w1 = (torch.randn((K, 1, 3, 3), generator=g) * 0.3).double()
    b1 = (torch.randn(K, generator=g) * 0.1).double()
    w2 = torch.randn((N, K), generator=g, dtype=torch.float64) / K ** 0.5
    b2 = torch.randn(N, generator=g) * 0.1
    wc = torch.randn((ncls, N), generator=g, dtype=torch.float64) / N ** 0.5
    bc = torch.randn(ncls, generator=g, dtype=torch.float64) * 0.1
    if ncls > 3:
        wc[3], bc[3] = wc[1], bc[1]
        bc[1] = bc[3] = 3.0                       # the tied pair is the maximum on most pixels
    if ncls == 5:
        bc -= 9.0                                 # every real logit negative: the zero rows that pad the weights to 32 classes must not win
    M = OH * OW
    Mp = (M + 255) // 256 * 256
    src = torch.stack([_nhwc_rows(xh), _nhwc_rows(xl)]).to(cuda_device)
    w2d, b2d = pack_split_rows(w2, 2).to(cuda_device), b2.to(cuda_device)
    wc32 = torch.zeros((32, N), dtype=torch.float64)
    wc32[:ncls] = wc
    bc32 = torch.zeros(32)
    bc32[:ncls] = bc.float()
    wch, wcl = split_f16(wc32)
    wcd, bcd = torch.stack([wch, wcl]).to(cuda_device), bc32.to(cuda_device)
    logits = torch.full((Mp, ncls), -7.0, dtype=torch.float32, device=cuda_device)
    labels = torch.full((Mp,), 99, dtype=torch.uint8, device=cuda_device)
    params = torch.cat([pack_dw_f32(w1, b1), dwpw_block_order(OH, OW) if blocks else dwpw_tile_order(OH, OW, d)]).to(cuda_device)
    op = AvlSegOp()
    op.kind, op.dtype = OP_DWPW, _lib.AVL_F16
    op.in_, op.in_lo, op.in2, op.in2_lo, op.in3, op.in3_c = src[0].data_ptr(), src[1].data_ptr(), params.data_ptr(), bcd.data_ptr(), wcd.data_ptr(), ncls
    op.out, op.out_mx, op.out_f32 = logits.data_ptr(), labels.data_ptr(), 1
    op.weight, op.bias, op.w_split, op.w_layout = w2d.data_ptr(), b2d.data_ptr(), 3, blocks
    op.in_h, op.in_w, op.in_c, op.in_ld, op.in_rows = H, W, K, K, src.shape[1]
    op.out_h, op.out_w, op.out_c, op.out_ld, op.out_rows = OH, OW, N, ncls, Mp
    op.relu, op.w_rows, op.ksize, op.stride, op.pad, op.dil, op.groups = 1, 256, 3, 1, pad, d, K
    _run_plan([op])
    a64 = F.relu(F.conv2d(xh.double() + xl.double(), w1.float().double(), b1.float().double(), groups=K))
    ah, al = _split(a64)
    w_hi, w_lo = _split(w2)
    y64 = F.relu(F.conv2d(ah.double() + al.double(), (w_hi.double() + w_lo.double()).view(N, K, 1, 1), b2.double()))
    yh, yl = _split(y64)
    ref = F.conv2d(yh.double() + yl.double(), (wch.double() + wcl.double())[:ncls].view(ncls, N, 1, 1), bc.float().double())[0]      # [ncls, OH, OW]
    got = logits[:M].cpu().double().t().reshape(ncls, OH, OW)
    err = float((got - ref).abs().max() / ref.abs().max())
    print("dwpw + classifier %s: %.3e" % (case, err))
    assert err <= 2 * TOL, "dwpw + classifier %s: %.3e" % (case, err)
    lab = labels[:M].cpu().long()
    assert torch.equal(lab, torch.argmax(logits[:M].cpu(), dim=1))
    if ncls > 3:
        assert int((lab == 1).sum()) > M // 2 and int((lab == 3).sum()) == 0 and bool((logits[:M, 1] == logits[:M, 3]).all())
    assert torch.all(labels[M:] == 99) and torch.all(logits[M:] == -7.0)


@pytest.mark.parametrize("cls", [0, 1])
def test_split_input_fused_depthwise_kernel_repeats_under_load(cls, cuda_device):
    """Race screen for k_dwpw_xs (weights and depthwise parameters by inline-asm LDS-DMA, awaited with hand-counted `s_waitcnt vmcnt(18)` next to
    the compiler's own waits for the tap loads; with cls the classifier epilogue re-uses the weight ring and the tiles as its staging area): a
    decoder-sized launch (several tiles per CU), back to back with copies in between, must give the same bytes every time."""
    import torch
    from vision_semantic_segmentation_amd import _lib
    from vision_semantic_segmentation_amd.network import OP_DWPW, AvlSegOp, dwpw_block_order, pack_dw_f32, pack_split_rows, split_f16
    H, W, K, N, ncls = 200, 330, 256, 256, 19
    OH, OW = H - 2, W - 2
    M, Mi = OH * OW, H * W
    Mp, Mip = (M + 255) // 256 * 256, (Mi + 255) // 256 * 256
    g = torch.Generator().manual_seed(21)
    x = torch.zeros((2, Mip, K), dtype=torch.float16)
    x[0, :Mi] = torch.randn((Mi, K), generator=g).to(torch.float16)
    x[1, :Mi] = (torch.randn((Mi, K), generator=g) * 2.0 ** -12).to(torch.float16)
    w1, b1 = torch.randn((K, 1, 3, 3), generator=g).double() * 0.3, torch.randn(K, generator=g).double() * 0.1
    w2 = torch.randn((N, K), generator=g, dtype=torch.float64) / K ** 0.5
    xd, w2d, b2d = x.to(cuda_device), pack_split_rows(w2, 2).to(cuda_device), torch.randn(N, generator=g).to(cuda_device)
    params = torch.cat([pack_dw_f32(w1, b1), dwpw_block_order(OH, OW)]).to(cuda_device)
    op = AvlSegOp()
    op.kind, op.dtype = OP_DWPW, _lib.AVL_F16
    op.in_, op.in_lo, op.in2, op.weight, op.bias = xd[0].data_ptr(), xd[1].data_ptr(), params.data_ptr(), w2d.data_ptr(), b2d.data_ptr()
    op.in_h, op.in_w, op.in_c, op.in_ld, op.in_rows = H, W, K, K, Mip
    op.out_h, op.out_w, op.out_c, op.out_rows = OH, OW, N, Mp
    op.relu, op.w_rows, op.ksize, op.stride, op.pad, op.dil, op.groups, op.w_split, op.w_layout = 1, 256, 3, 1, 0, 1, K, 3, 1
    if cls:
        wc32 = torch.zeros((32, N), dtype=torch.float64)
        wc32[:ncls] = torch.randn((ncls, N), generator=g, dtype=torch.float64) / N ** 0.5
        wcd, bcd = torch.stack(split_f16(wc32)).to(cuda_device), torch.zeros(32, device=cuda_device)
        out = torch.zeros((Mp, ncls), dtype=torch.float32, device=cuda_device)
        labels = torch.zeros(Mp, dtype=torch.uint8, device=cuda_device)
        op.in2_lo, op.in3, op.in3_c, op.out, op.out_mx, op.out_f32, op.out_ld = bcd.data_ptr(), wcd.data_ptr(), ncls, out.data_ptr(), labels.data_ptr(), 1, ncls
    else:
        out = torch.zeros((2, Mp, N), dtype=torch.float16, device=cuda_device)
        op.out, op.out_lo, op.out_ld = out[0].data_ptr(), out[1].data_ptr(), N
    plan = C.c_void_p()
    _lib.check(_lib.lib().avl_seg_plan_create((AvlSegOp * 1)(op), 1, C.byref(plan)), "avl_seg_plan_create")
    try:
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(_lib.lib().avl_seg_plan_run(plan, s), "avl_seg_plan_run")
        torch.cuda.synchronize()
        ref = out.clone()
        assert bool(torch.isfinite(ref.float()).all()) and float(ref.float().abs().max()) > 0
        copies = [torch.zeros_like(out) for _ in range(4)]
        bad = 0
        for i in range(120):
            _lib.lib().avl_seg_plan_run(plan, s)
            copies[i % 4].copy_(out)
            if i % 4 == 3:
                torch.cuda.synchronize()
                bad += sum(0 if torch.equal(c, ref) else 1 for c in copies)
        assert bad == 0, "%d of 120 launches differ from the first" % bad
    finally:
        _lib.lib().avl_seg_plan_destroy(plan)


@pytest.mark.parametrize("case", [(37, 53, 64, 1, 0), (20, 31, 512, 1, 0), (9, 9, 64, 2, 2)])
def test_split_depthwise_and_bilinear(case, cuda_device):
    import torch
    import torch.nn.functional as F
    from test_gpu_ops import _from_rows, _nhwc_rows, _spatial_op
    from vision_semantic_segmentation_amd import _lib
    from vision_semantic_segmentation_amd.network import OP_BILINEAR, OP_DWCONV
    H, W, Cc, d, pad = case
    g = torch.Generator().manual_seed(H * 1000 + W + d)
    x64 = torch.randn((1, Cc, H, W), generator=g, dtype=torch.float64)
    x_hi, x_lo = _split(x64)
    xv = x_hi.double() + x_lo.double()
    w = torch.randn((Cc, 1, 3, 3), generator=g) * 0.3
    b = torch.randn(Cc, generator=g) * 0.1
    OH, OW = H + 2 * pad - 2 * d, W + 2 * pad - 2 * d
    ref = F.relu(F.conv2d(xv, w.double(), b.double(), padding=pad, dilation=d, groups=Cc))
    src = torch.stack([_nhwc_rows(x_hi), _nhwc_rows(x_lo)]).to(cuda_device)
    dst = torch.full((2, (OH * OW + 255) // 256 * 256, Cc), 7.0, dtype=torch.float16, device=cuda_device)
    wd = w.reshape(Cc, 9).t().contiguous().reshape(-1).to(cuda_device)
    bd = b.to(cuda_device)
    zero = torch.zeros(64, dtype=torch.uint8, device=cuda_device)
    _run_plan([_spatial_op(OP_DWCONV, _lib.AVL_F16, src[0], (H, W), Cc, dst[0], (OH, OW), Cc, weight=wd.data_ptr(), bias=bd.data_ptr(),
                           in2=zero.data_ptr(), ksize=3, stride=1, pad=pad, dil=d, groups=Cc, relu=1,
                           in_lo=src[1].data_ptr(), out_lo=dst[1].data_ptr())])
    got = _from_rows(dst[0].cpu().double() + dst[1].cpu().double(), OH, OW, Cc)
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err <= TOL, "split depthwise %s: %.3e" % (case, err)
    assert torch.all(dst[:, OH * OW:] == 7.0)
    # bilinear x2 (align_corners) on the same split input
    oh, ow = 2 * H, 2 * W
    refb = F.interpolate(xv.float(), size=(oh, ow), mode="bilinear", align_corners=True).double()
    dstb = torch.zeros((2, (oh * ow + 255) // 256 * 256, Cc), dtype=torch.float16, device=cuda_device)
    _run_plan([_spatial_op(OP_BILINEAR, _lib.AVL_F16, src[0], (H, W), Cc, dstb[0], (oh, ow), Cc, in_lo=src[1].data_ptr(), out_lo=dstb[1].data_ptr())])
    gotb = _from_rows(dstb[0].cpu().double() + dstb[1].cpu().double(), oh, ow, Cc)
    errb = float((gotb - refb).abs().max() / refb.abs().max())
    assert errb <= 3e-5, "split bilinear %s: %.3e" % (case, errb)          # fp32 interpolation weights (see test_bilinear_align_corners)


@pytest.mark.parametrize("split", [False, True])
def test_ring_gemm_with_residual_repeats_under_load(split, cuda_device):
    """Race screen for k_gemm_ring (hand-counted vmcnt around inline-asm LDS-DMA next to compiler-visible residual loads
    and stores): a conv3 + residual shape with more tiles than CUs, launched back to back with copies in between, must
    give the same bytes every time -- plain f16 and the split (3-pass, split residual and output) variant."""
    import torch
    from vision_semantic_segmentation_amd import _lib
    from vision_semantic_segmentation_amd.network import OP_GEMM, AvlSegOp, pack_split_rows
    M, K, N = 256 * 300, 256, 512
    g = torch.Generator().manual_seed(3)
    a = torch.randn((2, M, K), generator=g).to(torch.float16).to(cuda_device)
    a[1] *= 2 ** -11
    r = torch.randn((2, M, N), generator=g).to(torch.float16).to(cuda_device)
    r[1] *= 2 ** -11
    w64 = torch.randn((N, K), generator=g, dtype=torch.float64) / K ** 0.5
    wd = (pack_split_rows(w64, 3) if split else w64.to(torch.float16)).to(cuda_device)
    bd = torch.randn(N, generator=g).to(cuda_device)
    out = torch.zeros((2, M, N), dtype=torch.float16, device=cuda_device)
    op = AvlSegOp()
    op.kind, op.dtype = OP_GEMM, _lib.AVL_F16
    op.in_, op.out, op.weight, op.bias = a[0].data_ptr(), out[0].data_ptr(), wd.data_ptr(), bd.data_ptr()
    op.in_h, op.in_w, op.in_c, op.in_ld, op.in_rows = 1, M, K, K, M
    op.out_h, op.out_w, op.out_c, op.out_ld, op.out_rows = 1, M, N, N, M
    op.relu, op.w_rows, op.ksize, op.stride, op.dil, op.groups = 1, N, 1, 1, 1, 1
    op.in2, op.in2_ld = r[0].data_ptr(), N
    if split:
        op.w_split, op.in_lo, op.in2_lo, op.out_lo = 1, a[1].data_ptr(), r[1].data_ptr(), out[1].data_ptr()
    plan = C.c_void_p()
    _lib.check(_lib.lib().avl_seg_plan_create((AvlSegOp * 1)(op), 1, C.byref(plan)), "avl_seg_plan_create")
    try:
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(_lib.lib().avl_seg_plan_run(plan, s), "avl_seg_plan_run")
        torch.cuda.synchronize()
        ref = out.clone()
        outs = [torch.zeros_like(out) for _ in range(4)]
        for i in range(80):
            _lib.lib().avl_seg_plan_run(plan, s)
            outs[i % 4].copy_(out)
            if i % 4 == 3:
                torch.cuda.synchronize()
                for o in outs:
                    assert torch.equal(o, ref), "launch ~%d differs from the first one" % i
    finally:
        _lib.lib().avl_seg_plan_destroy(plan)


def _cfg(precision):
    from vision_semantic_segmentation_amd.config import get_network_cfg_defaults
    cfg = get_network_cfg_defaults()
    cfg.MODEL.PRECISION = precision
    return cfg


@pytest.fixture(scope="module")
def state():
    from vision_semantic_segmentation_amd.network import random_state_dict
    return random_state_dict(0)


@pytest.mark.parametrize("hw", [(96, 128), (320, 416), (480, 640)])
def test_mixed_logits_within_1e3_of_oracle(state, hw, cuda_device):
    """north_star: segmentation logits within 1e-3 (relative to max|logit|) of the reference's fp32 forward, in the mode
    bench.py times (MODEL.PRECISION = "mixed")."""
    from test_gpu_seg import _compare
    rel, agree = _compare(state, "mixed", hw[0], hw[1], cuda_device)
    print("mixed %dx%d: max rel err %.3e, argmax agreement %.5f" % (hw[0], hw[1], rel, agree))
    assert rel <= 1e-3
    assert agree >= 0.998


# weight seeds 0-3 x one frame at 480 x 640 (two frames each until round 5: the suite took 526 s on one of the pool's slow boxes; the 28-combination
# sweep is tools/seed_sweep.py, profiles/r04/seed_sweep.log), weight seeds 0 and 2 (the worst draw of that sweep) on the bench frame
SEED_CASES = [(ws, 30 + 2 * ws + 1, 480, 640) for ws in range(4)] + [(0, "bench", 1080, 1920), (2, "bench", 1080, 1920)]


@pytest.mark.parametrize("wseed,iseed,h,w", SEED_CASES)
def test_mixed_logits_across_weight_seeds(wseed, iseed, h, w, cuda_device):
    """The logits error of the mixed mode follows the WEIGHTS draw, so the 1e-3 of north_star is asserted
    with margin (9e-4) over four weight seeds, two frames each, and at the bench's full size for the bench's own weights and
    for the worst draw -- in the DEFAULT configuration (what SemanticSegmentation builds from get_cfg_defaults())."""
    import torch
    import _full_size as fs
    from vision_semantic_segmentation_amd import SemanticSegmentation
    seg = SemanticSegmentation(_cfg("mixed"), device=cuda_device, state_dict=fs.state_dict(wseed))
    img = fs.image_for(iseed, h, w)
    got = seg.logits(img).float().cpu()
    ref = fs.oracle_logits(wseed, iseed, h, w)
    rel = float((got - ref).abs().max() / ref.abs().max())
    agree = float((got.argmax(0) == ref.argmax(0)).float().mean())
    print("mixed (default), weights seed %d, frame %s, %dx%d: max rel err %.3e, argmax agreement %.5f" % (wseed, iseed, h, w, rel, agree))
    del seg
    torch.cuda.empty_cache()
    assert rel <= 9e-4
    assert agree >= 0.998


def test_mixed_logits_without_the_mx_grouped_conv(state, cuda_device):
    """MODEL.MIXED_GCONV_MX = False (the round-2 default: conv1's output as one f16 plane, the grouped 3x3 with split weights only):
    still inside 1e-3 on these weights (not on every draw: that is why it is no longer the default), through the cfg switch."""
    import numpy as np
    from oracle import network_oracle as no
    from test_gpu_seg import _cfg
    from vision_semantic_segmentation_amd import SemanticSegmentation
    from vision_semantic_segmentation_amd.network import OP_GCONV
    cfg = _cfg("mixed")
    assert cfg.MODEL.MIXED_GCONV_MX is True                                # the default: FP4 corrections inside the grouped conv
    seg_default = SemanticSegmentation(cfg, device=cuda_device, state_dict=state)
    assert any(op.w_split == 2 and op.kind == OP_GCONV for op in seg_default.net_for(320, 416).ops)
    cfg.MODEL.MIXED_GCONV_MX = False
    seg = SemanticSegmentation(cfg, device=cuda_device, state_dict=state)
    img = np.random.default_rng(0).integers(0, 256, size=(320, 416, 3), dtype=np.uint8)
    got = seg.logits(img).cpu()
    assert not any(op.w_split == 2 and op.kind == OP_GCONV for op in seg.net_for(320, 416).ops)
    ref = no.forward_logits(state, img)[0]
    rel = float((got - ref).abs().max() / ref.abs().max())
    print("mixed without the MX grouped conv 320x416: max rel err %.3e" % rel)
    assert rel <= 1e-3


def test_mixed_logits_with_layer1_lo_planes(state, cuda_device):
    """MODEL.MIXED_LAYER1_LO = True (every block of layer1 keeps a lo plane of its output; the default keeps it for the last block
    only): more HBM traffic, a smaller error -- through the cfg switch, and the plan really differs."""
    import numpy as np
    from oracle import network_oracle as no
    from test_gpu_seg import _cfg
    from vision_semantic_segmentation_amd import SemanticSegmentation
    cfg = _cfg("mixed")
    assert cfg.MODEL.MIXED_LAYER1_LO is True
    img = np.random.default_rng(0).integers(0, 256, size=(320, 416, 3), dtype=np.uint8)
    ref = no.forward_logits(state, img)[0]
    errs, los = [], []
    for flag in (False, True):
        cfg.MODEL.MIXED_LAYER1_LO = flag
        seg = SemanticSegmentation(cfg, device=cuda_device, state_dict=state)
        got = seg.logits(img).cpu()
        net = seg.net_for(320, 416)
        # (layer1's blocks are one fused op each -- AVL_OP_BOTTLENECK -- named after the block; the three-launch form names conv3)
        los.append(sum(1 for n, op in zip(net.op_names, net.ops)
                       if n.startswith("backbone.layer1.") and (n.endswith("conv3") or n.count(".") == 2) and op.out_lo))
        errs.append(float((got - ref).abs().max() / ref.abs().max()))
    print("mixed 320x416: layer1 lo planes off / on: max rel err %.3e / %.3e" % tuple(errs))
    assert los == [1, 3]                       # conv3 outputs of layer1 that carry a lo plane
    assert errs[0] <= 9e-4 and errs[1] <= errs[0] * 1.15      # (max-norm errors of two plans 2^-12 apart in one tensor class are not ordered; the fused blocks' conv1 never reads the lo plane: it enters the residual sum only)


def _bundle(hi64, lo64, rows_pad):
    """MX bundle [Q4(hi) | scales | Q4(lo) | scales] of a split tensor, rows padded with zeros"""
    import torch
    from vision_semantic_segmentation_amd.network import mx_quant_fp4
    parts = []
    for t in (hi64, lo64):
        q, s = mx_quant_fp4(_pad_rows(t, rows_pad))
        parts += [q.reshape(-1), s.reshape(-1)]
    return torch.cat(parts)


def _unbundle(buf, rows, c, half):
    from vision_semantic_segmentation_amd.network import mx_bundle_bytes, mx_dequant_fp4
    hb = mx_bundle_bytes(rows, c)
    b = buf[half * hb:(half + 1) * hb]
    q = b[:rows * (c // 2)].reshape(rows, c // 2)
    s = b[rows * (c // 2):].reshape(c // 256, rows, 8)
    return q, s, mx_dequant_fp4(q, s)


@pytest.mark.parametrize("case", [  # (M, K, N, A split, residual split, out split, quantise output, relu)
    (2600, 256, 512, False, None, False, False, True),
    (2600, 256, 512, True, None, True, True, True),
    (17000, 512, 1024, False, True, True, True, True),        # conv3-like: split residual and output, FP4 copy of the output
    (34000, 512, 256, True, None, False, False, True),        # conv1-like: both corrections (128-row tiles: 266 of them)
    (300, 2048, 2048, True, False, True, True, False),        # fewer rows than one tile, single-plane residual
    (9000, 512, 1024, "fp4", "fp4", "fp4", True, True),       # the trunk form: every lo part only as FP4 (mx_flags)
    # K >= 1024 runs the software-pipelined stream (k_gemm_mx_pipe), K < 1024 the two-barrier kernel (k_gemm_ring_mx): the same forms again
    (17000, 1024, 1024, "fp4", "fp4", "fp4", True, True),     # trunk form, more tiles than CUs, 256-row tiles
    (34000, 1024, 256, True, None, False, False, True),       # both corrections, 128-row tiles
    (9000, 1280, 512, False, True, True, True, True),         # odd number of K macro-blocks, split residual and output
])
def test_mx_gemm(case, cuda_device):
    """w_split = 2: main product on f16 hi parts, corrections Q4(W lo) x Q4(x hi) [+ Q4(W hi) x Q4(x lo)] on the block-scaled
    matrix cores.  Reference: float64 evaluation of exactly that sum on the operands the kernel is given."""
    import torch
    from vision_semantic_segmentation_amd import _lib
    from vision_semantic_segmentation_amd.network import OP_GEMM, AvlSegOp, mx_dequant_fp4, mx_quant_fp4, pack_mx_weights
    M, K, N, a_split, r_split, o_split, quant_out, relu = case
    fp4_only = a_split == "fp4"
    if fp4_only:
        a_split, r_split, o_split = True, True, False
    g = torch.Generator().manual_seed(M + K + N)
    Mp = (M + 255) // 256 * 256
    a64 = torch.randn((M, K), generator=g, dtype=torch.float64) * torch.exp2(torch.randint(-3, 4, (M, 1), generator=g).double())
    a_hi, a_lo = _split(a64)
    if not a_split:
        a_lo = torch.zeros_like(a_lo)
    w64 = torch.randn((N, K), generator=g, dtype=torch.float64) / K ** 0.5
    w_hi16, wbundle = pack_mx_weights(w64)
    w_hi, w_lo = _split(w64)
    b = torch.randn(N, generator=g)
    deq = lambda t: mx_dequant_fp4(*mx_quant_fp4(t.double()))            # noqa: E731
    ref = a_hi.double() @ w_hi.double().t() + deq(_pad_rows(a_hi, Mp))[:M] @ deq(w_lo).t() + b.double()
    if a_split:
        ref = ref + deq(_pad_rows(a_lo, Mp))[:M] @ deq(w_hi).t()
    if r_split is not None:
        r64 = torch.randn((M, N), generator=g, dtype=torch.float64)
        r_hi, r_lo = _split(r64)
        ref = ref + r_hi.double() + ((deq(_pad_rows(r_lo, Mp))[:M] if fp4_only else r_lo.double()) if r_split else 0)
    if relu:
        ref = torch.relu(ref)
    planes = torch.stack([_pad_rows(a_hi, Mp), _pad_rows(a_lo, Mp)]).to(cuda_device)
    in_mx = _bundle(a_hi, a_lo, Mp).to(cuda_device)
    wd, wmx, bd = w_hi16.to(cuda_device), wbundle.to(cuda_device), b.to(cuda_device)
    out = torch.full((2, Mp, N), 7.0, dtype=torch.float16, device=cuda_device)
    from vision_semantic_segmentation_amd.network import mx_bundle_bytes
    out_mx = torch.full((2 * mx_bundle_bytes(Mp, N),), 0xEE, dtype=torch.uint8, device=cuda_device)
    op = AvlSegOp()
    op.kind, op.dtype = OP_GEMM, _lib.AVL_F16
    op.in_, op.out, op.weight, op.bias = planes[0].data_ptr(), out[0].data_ptr(), wd.data_ptr(), bd.data_ptr()
    op.in_h, op.in_w, op.in_c, op.in_ld, op.in_rows = 1, M, K, K, Mp
    op.out_h, op.out_w, op.out_c, op.out_ld, op.out_rows = 1, M, N, N, Mp
    op.relu, op.w_rows, op.ksize, op.stride, op.dil, op.groups = int(relu), N, 1, 1, 1, 1
    op.w_split, op.w_mx, op.in_mx = 2, wmx.data_ptr(), in_mx.data_ptr()
    if a_split and not fp4_only:
        op.in_lo = planes[1].data_ptr()
    if o_split:
        op.out_lo = out[1].data_ptr()
    if quant_out:
        op.out_mx = out_mx.data_ptr()
    if r_split is not None:
        rd = torch.stack([_pad_rows(r_hi, Mp), _pad_rows(r_lo, Mp)]).to(cuda_device)
        op.in2, op.in2_ld = rd[0].data_ptr(), N
        if r_split and not fp4_only:
            op.in2_lo = rd[1].data_ptr()
    if fp4_only:
        from vision_semantic_segmentation_amd.network import AVL_MX_IN_LO, AVL_MX_OUT_LO, AVL_MX_RES_LO
        r_mx = _bundle(r_hi, r_lo, Mp).to(cuda_device)
        op.in2_mx, op.mx_flags = r_mx.data_ptr(), AVL_MX_IN_LO | AVL_MX_RES_LO | AVL_MX_OUT_LO
    _run_plan([op])
    got = out[0, :M].cpu().double() + (out[1, :M].cpu().double() if o_split else 0)
    if fp4_only:
        # value = f16 plane + FP4 lo half of the bundle: the lo part carries FP4's ~12 % error, i.e. 2^-11 * 0.12 of the value
        v_hi, v_lo = _unbundle(out_mx.cpu(), Mp, N, 0)[2], _unbundle(out_mx.cpu(), Mp, N, 1)[2]
        want_lo = ref - got
        assert float((v_lo[:M] - want_lo).abs().max() / ref.abs().max()) <= 2 ** -11 * 0.3
        assert torch.all(out[1] == 7.0)                                  # no f16 lo plane is written
        err = float((got + v_lo[:M] - ref).abs().max() / ref.abs().max())
        assert err <= 2 ** -11 * 0.3, "mx gemm %s: %.3e" % (case, err)
        assert float((v_hi[:M] - got).abs().max() / ref.abs().max()) <= 0.3      # the FP4 copy of the hi plane is a 2-3 bit image of it
        return
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err <= (TOL if o_split else 2 ** -11 * 1.5), "mx gemm %s: %.3e" % (case, err)
    assert torch.all(out[0, M:] == 7.0)
    if quant_out:
        # the FP4 copy of the output equals the host quantisation of the planes the kernel wrote, byte for byte
        for half, plane in ((0, out[0]),) + (((1, out[1]),) if o_split else ()):
            q_dev, s_dev, v_dev = _unbundle(out_mx.cpu(), Mp, N, half)
            q_ref, s_ref = mx_quant_fp4(plane[:M].cpu().double())
            assert torch.equal(s_dev[:, :M], s_ref), "scales of plane %d differ" % half
            assert torch.equal(v_dev[:M], mx_dequant_fp4(q_ref, s_ref)), "FP4 plane %d differs" % half      # values: -0 == +0
        assert torch.all(_unbundle(out_mx.cpu(), Mp, N, 0)[0][M:] == 0xEE)          # rows past M untouched


@pytest.mark.parametrize("shape", [  # (M, K, N, correction passes, K of a second input appended along K or 0)
    # (round 5: sizes cut to ~40 % -- the host-side FP4 packing of the operands was most of this test's 85 s; every case still gives
    # each CU several tiles)
    (256 * 300, 1024, 512, 2, 0),            # 256-row tiles, 2.3 tiles per CU
    (256 * 140 + 77, 1024, 256, 1, 0),       # 128-row tiles (k_gemm_mx_pipe<., 4, ...>), weights-only correction, ragged last tile
    (256 * 180, 512, 512, 2, 1024),          # conv3 + downsample form: the stream switches inputs at K macro-block 2 of 6 (set_input)
    (256 * 140, 1024, 256, 2, 512),          # the same on 128-row tiles
    (256 * 300, 512, 512, 2, 0),             # K < 1024: the two-barrier kernel k_gemm_ring_mx (kept: it runs the short-K layers)
])
def test_mx_gemm_repeats_under_load(shape, cuda_device):
    """Race screen for the software-pipelined MX GEMM (k_gemm_mx_pipe, every case with K >= 1024: LDS slots re-filled by inline-asm
    LDS-DMA behind a barrier that sits in the middle of the MFMA stream, fragments read ahead across it): conv3-like shapes with
    several tiles per CU, 256- and 128-row tiles, one and two correction passes, one and two inputs along K (no unit
    case used to drive the pipe kernel's input switch), FP4 residual and output, launched back to back with copies in between,
    must give the same bytes every time -- and the first launch is checked against float64."""
    import torch
    from vision_semantic_segmentation_amd import _lib
    from vision_semantic_segmentation_amd.network import (AVL_MX_IN_LO, AVL_MX_OUT_LO, AVL_MX_RES_LO, OP_GEMM, AvlSegOp, mx_bundle_bytes,
                                                          mx_dequant_fp4, mx_quant_fp4, pack_mx_weights)
    M, K, N, nmx, K2 = shape
    Mp = (M + 255) // 256 * 256
    g = torch.Generator().manual_seed(11)
    a_hi = torch.randn((Mp, K), generator=g).to(torch.float16)
    a_lo = (torch.randn((Mp, K), generator=g) * 2 ** -11).to(torch.float16)
    if K2:
        assert nmx == 2
        b_hi = torch.randn((Mp, K2), generator=g).to(torch.float16)
        b_lo = (torch.randn((Mp, K2), generator=g) * 2 ** -11).to(torch.float16)
    r_hi = torch.randn((Mp, N), generator=g).to(torch.float16)
    r_lo = (torch.randn((Mp, N), generator=g) * 2 ** -11).to(torch.float16)
    w64 = torch.randn((N, K + K2), generator=g, dtype=torch.float64) / (K + K2) ** 0.5
    w_hi16, wbundle = pack_mx_weights(w64)
    bd = torch.randn(N, generator=g).to(cuda_device)
    a_d, r_d = a_hi.to(cuda_device), r_hi.to(cuda_device)
    in_mx, r_mx = _bundle(a_hi, a_lo, Mp).to(cuda_device), _bundle(r_hi, r_lo, Mp).to(cuda_device)
    if K2:
        b_d, in3_mx = b_hi.to(cuda_device), _bundle(b_hi, b_lo, Mp).to(cuda_device)
    wd, wmx = w_hi16.to(cuda_device), wbundle.to(cuda_device)
    out = torch.zeros((Mp, N), dtype=torch.float16, device=cuda_device)
    out_mx = torch.zeros(2 * mx_bundle_bytes(Mp, N), dtype=torch.uint8, device=cuda_device)
    op = AvlSegOp()
    op.kind, op.dtype = OP_GEMM, _lib.AVL_F16
    op.in_, op.out, op.weight, op.bias = a_d.data_ptr(), out.data_ptr(), wd.data_ptr(), bd.data_ptr()
    op.in_h, op.in_w, op.in_c, op.in_ld, op.in_rows = 1, M, K, K, Mp
    op.out_h, op.out_w, op.out_c, op.out_ld, op.out_rows = 1, M, N, N, Mp
    op.relu, op.w_rows, op.ksize, op.stride, op.dil, op.groups = 1, N, 1, 1, 1, 1
    op.w_split, op.w_mx, op.in_mx, op.out_mx = 2, wmx.data_ptr(), in_mx.data_ptr(), out_mx.data_ptr()
    op.in2, op.in2_ld, op.in2_mx = r_d.data_ptr(), N, r_mx.data_ptr()
    op.mx_flags = (AVL_MX_IN_LO if nmx == 2 else 0) | AVL_MX_RES_LO | AVL_MX_OUT_LO
    if K2:
        op.in3, op.in3_mx, op.in3_c, op.in3_ld = b_d.data_ptr(), in3_mx.data_ptr(), K2, K2
    plan = C.c_void_p()
    _lib.check(_lib.lib().avl_seg_plan_create((AvlSegOp * 1)(op), 1, C.byref(plan)), "avl_seg_plan_create")
    try:
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(_lib.lib().avl_seg_plan_run(plan, s), "avl_seg_plan_run")
        torch.cuda.synchronize()
        ref, ref_mx = out.clone(), out_mx.clone()
        # the first 4096 rows and the last tile against float64 (the whole product is checked at smaller sizes in test_mx_gemm)
        deq = lambda t: mx_dequant_fp4(*mx_quant_fp4(t.double()))            # noqa: E731
        w_hi = w64.to(torch.float16)
        w_lo = (w64 - w_hi.double()).to(torch.float16)
        for lo_, hi_ in ((0, 4096), (Mp - 256, M)):
            hi_p = Mp if hi_ == M else hi_
            xa, xl = a_hi[lo_:hi_p], a_lo[lo_:hi_p]
            want = xa.double() @ w_hi[:, :K].double().t() + deq(xa) @ deq(w_lo[:, :K]).t() + bd.cpu().double()
            if nmx == 2:
                want = want + deq(xl) @ deq(w_hi[:, :K]).t()
            if K2:
                xb, xbl = b_hi[lo_:hi_p], b_lo[lo_:hi_p]
                want = want + xb.double() @ w_hi[:, K:].double().t() + deq(xb) @ deq(w_lo[:, K:]).t() + deq(xbl) @ deq(w_hi[:, K:]).t()
            rr = r_hi[lo_:lo_ + want.shape[0]].double() + deq(r_lo[lo_:lo_ + want.shape[0]])
            want = torch.relu(want + rr)[:hi_ - lo_]
            got = ref[lo_:hi_].cpu().double()
            assert float((got - want).abs().max() / want.abs().max()) <= 2 ** -11 * 1.5
        outs = [(torch.zeros_like(out), torch.zeros_like(out_mx)) for _ in range(4)]
        for i in range(60):
            _lib.lib().avl_seg_plan_run(plan, s)
            outs[i % 4][0].copy_(out)
            outs[i % 4][1].copy_(out_mx)
            if i % 4 == 3:
                torch.cuda.synchronize()
                for o, om in outs:
                    assert torch.equal(o, ref) and torch.equal(om, ref_mx), "launch ~%d differs from the first one" % i
    finally:
        _lib.lib().avl_seg_plan_destroy(plan)


@pytest.mark.parametrize("case", [(23, 45, 256, 1, 1), (30, 41, 512, 1, 2), (19, 67, 1024, 1, 4)])
def test_grouped_conv_writes_the_mx_bundle(case, cuda_device):
    import torch
    from test_gpu_ops import _nhwc_rows, _spatial_op
    from vision_semantic_segmentation_amd import _lib
    from vision_semantic_segmentation_amd.network import OP_GCONV, mx_bundle_bytes, mx_quant_fp4, pack_gconv_windows
    H, W, width, s, d = case
    G = 32
    cg = width // G
    g = torch.Generator().manual_seed(H * 77 + W + width)
    x = torch.randn((1, width, H, W), generator=g).to(torch.float16)
    w64 = torch.randn((width, cg, 3, 3), generator=g, dtype=torch.float64) * (2.0 / (cg * 9)) ** 0.5
    b = torch.randn(width, generator=g) * 0.1
    w_hi, w_lo = _split(w64)
    src = _nhwc_rows(x).to(cuda_device)
    rows = (H * W + 255) // 256 * 256
    dst = torch.zeros((2, rows, width), dtype=torch.float16, device=cuda_device)
    mx = torch.full((2 * mx_bundle_bytes(rows, width),), 0xEE, dtype=torch.uint8, device=cuda_device)
    nwin = width // 32
    wd = torch.cat([pack_gconv_windows(w_hi.double(), G).reshape(nwin, 2, 9, 16, 32),
                    pack_gconv_windows(w_lo.double(), G).reshape(nwin, 2, 9, 16, 32)], dim=2).reshape(-1).to(torch.float16).to(cuda_device)
    bd = b.to(cuda_device)
    _run_plan([_spatial_op(OP_GCONV, _lib.AVL_F16, src, (H, W), width, dst[0], (H, W), width, weight=wd.data_ptr(), bias=bd.data_ptr(),
                           ksize=3, stride=s, pad=d, dil=d, groups=G, relu=1, w_layout=1, w_split=1, out_lo=dst[1].data_ptr(),
                           out_mx=mx.data_ptr())])
    M = H * W
    for half in (0, 1):
        q_dev, s_dev, v_dev = _unbundle(mx.cpu(), rows, width, half)
        q_ref, s_ref = mx_quant_fp4(dst[half, :M].cpu().double())
        from vision_semantic_segmentation_amd.network import mx_dequant_fp4
        assert torch.equal(s_dev[:, :M], s_ref) and torch.equal(v_dev[:M], mx_dequant_fp4(q_ref, s_ref)), "plane %d" % half


@pytest.mark.parametrize("ks", [(9000, 512, 256, 1024), (17000, 512, 1024, 1024), (12000, 1024, 512, 256), (9000, 256, 1024, 512)])
def test_mx_gemm_with_a_second_input_along_k(ks, cuda_device):
    """conv3 + stride-1 downsample as ONE MX GEMM: out = relu(W3 . t2 + Wd . x + b3 + bd), both inputs with FP4-only lo parts
    (in3 / in3_mx of include/avl_hip.h).  Reference: float64 on the operands the kernel is given.  K1 + K2 < 1024 runs
    k_gemm_ring_mx, the others the software-pipelined k_gemm_mx_pipe (256-row tiles; third case: 128-row tiles)."""
    import torch
    from vision_semantic_segmentation_amd import _lib
    from vision_semantic_segmentation_amd.network import (AVL_MX_IN_LO, AVL_MX_OUT_LO, OP_GEMM, AvlSegOp, mx_bundle_bytes, mx_dequant_fp4,
                                                          mx_quant_fp4, pack_mx_weights)
    M, K1, K2, N = ks
    g = torch.Generator().manual_seed(7)
    Mp = (M + 255) // 256 * 256
    deq = lambda t: mx_dequant_fp4(*mx_quant_fp4(t.double()))            # noqa: E731
    ins = []
    for K in (K1, K2):
        a64 = torch.randn((M, K), generator=g, dtype=torch.float64)
        hi, lo = _split(a64)
        ins.append((hi, lo))
    w64 = torch.randn((N, K1 + K2), generator=g, dtype=torch.float64) / (K1 + K2) ** 0.5
    w_hi16, wbundle = pack_mx_weights(w64)
    w_hi, w_lo = _split(w64)
    b = torch.randn(N, generator=g)
    ref = b.double().unsqueeze(0).repeat(M, 1)
    k0 = 0
    for (hi, lo), K in zip(ins, (K1, K2)):
        wh, wl = w_hi[:, k0:k0 + K], w_lo[:, k0:k0 + K]
        ref = ref + hi.double() @ wh.double().t() + deq(_pad_rows(hi, Mp))[:M] @ deq(wl).t() + deq(_pad_rows(lo, Mp))[:M] @ deq(wh).t()
        k0 += K
    ref = torch.relu(ref)
    dev = [(_pad_rows(hi, Mp).to(cuda_device), _bundle(hi, lo, Mp).to(cuda_device)) for hi, lo in ins]
    wd, wmx, bd = w_hi16.to(cuda_device), wbundle.to(cuda_device), b.to(cuda_device)
    out = torch.full((Mp, N), 7.0, dtype=torch.float16, device=cuda_device)
    out_mx = torch.zeros(2 * mx_bundle_bytes(Mp, N), dtype=torch.uint8, device=cuda_device)
    op = AvlSegOp()
    op.kind, op.dtype = OP_GEMM, _lib.AVL_F16
    op.in_, op.out, op.weight, op.bias = dev[0][0].data_ptr(), out.data_ptr(), wd.data_ptr(), bd.data_ptr()
    op.in_h, op.in_w, op.in_c, op.in_ld, op.in_rows = 1, M, K1, K1, Mp
    op.out_h, op.out_w, op.out_c, op.out_ld, op.out_rows = 1, M, N, N, Mp
    op.relu, op.w_rows, op.ksize, op.stride, op.dil, op.groups = 1, N, 1, 1, 1, 1
    op.w_split, op.w_mx, op.in_mx, op.out_mx = 2, wmx.data_ptr(), dev[0][1].data_ptr(), out_mx.data_ptr()
    op.in3, op.in3_mx, op.in3_c, op.in3_ld = dev[1][0].data_ptr(), dev[1][1].data_ptr(), K2, K2
    op.mx_flags = AVL_MX_IN_LO | AVL_MX_OUT_LO
    _run_plan([op])
    v_lo = _unbundle(out_mx.cpu(), Mp, N, 1)[2]
    got = out[:M].cpu().double() + v_lo[:M]
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err <= 2 ** -11 * 0.3, "mx gemm with a second input: %.3e" % err
    assert torch.all(out[M:] == 7.0)


@pytest.mark.parametrize("case", [(37, 53, 256, 1, 0), (20, 31, 512, 1, 0)])
def test_split_depthwise_writes_the_mx_bundle(case, cuda_device):
    """The mixed decoder's depthwise conv feeding an MX GEMM: f16 hi plane + FP4 copies of hi and of the lo part (which is kept as
    FP4 only, AVL_MX_OUT_LO)."""
    import torch
    import torch.nn.functional as F
    from test_gpu_ops import _from_rows, _nhwc_rows, _spatial_op
    from vision_semantic_segmentation_amd import _lib
    from vision_semantic_segmentation_amd.network import AVL_MX_OUT_LO, OP_DWCONV, mx_bundle_bytes, mx_dequant_fp4, mx_quant_fp4
    H, W, Cc, d, pad = case
    g = torch.Generator().manual_seed(H * 1000 + W + d)
    x_hi, x_lo = _split(torch.randn((1, Cc, H, W), generator=g, dtype=torch.float64))
    w = torch.randn((Cc, 1, 3, 3), generator=g) * 0.3
    b = torch.randn(Cc, generator=g) * 0.1
    OH, OW = H + 2 * pad - 2 * d, W + 2 * pad - 2 * d
    ref = F.relu(F.conv2d(x_hi.double() + x_lo.double(), w.double(), b.double(), padding=pad, dilation=d, groups=Cc))
    src = torch.stack([_nhwc_rows(x_hi), _nhwc_rows(x_lo)]).to(cuda_device)
    rows = (OH * OW + 255) // 256 * 256
    dst = torch.full((2, rows, Cc), 7.0, dtype=torch.float16, device=cuda_device)
    mx = torch.full((2 * mx_bundle_bytes(rows, Cc),), 0xEE, dtype=torch.uint8, device=cuda_device)
    wd, bd = w.reshape(Cc, 9).t().contiguous().reshape(-1).to(cuda_device), b.to(cuda_device)
    zero = torch.zeros(64, dtype=torch.uint8, device=cuda_device)
    _run_plan([_spatial_op(OP_DWCONV, _lib.AVL_F16, src[0], (H, W), Cc, dst[0], (OH, OW), Cc, weight=wd.data_ptr(), bias=bd.data_ptr(),
                           in2=zero.data_ptr(), ksize=3, stride=1, pad=pad, dil=d, groups=Cc, relu=1, in_lo=src[1].data_ptr(),
                           out_mx=mx.data_ptr(), mx_flags=AVL_MX_OUT_LO)])
    M = OH * OW
    hi = dst[0, :M].cpu().double()
    assert torch.all(dst[1] == 7.0)                                           # no f16 lo plane
    q_dev, s_dev, v_hi = _unbundle(mx.cpu(), rows, Cc, 0)
    q_ref, s_ref = mx_quant_fp4(hi)
    assert torch.equal(s_dev[:, :M], s_ref) and torch.equal(v_hi[:M], mx_dequant_fp4(q_ref, s_ref))
    v_lo = _unbundle(mx.cpu(), rows, Cc, 1)[2][:M]
    want = _from_rows_inv(ref, M, Cc)
    assert float((hi + v_lo - want).abs().max() / want.abs().max()) <= 2 ** -11 * 0.3


def _from_rows_inv(x, m, c):
    """[1,C,H,W] -> [H*W][C]"""
    return x[0].permute(1, 2, 0).reshape(m, c)


@pytest.mark.parametrize("case", [(23, 45, 256, 1, 1), (30, 41, 512, 1, 2), (19, 67, 1024, 1, 4), (26, 40, 256, 2, 1)])
@pytest.mark.parametrize("in_lo", [False, True])
def test_mx_grouped_conv(case, in_lo, cuda_device):
    """k_gconv_mx: f16 hi product + FP4 corrections Q4(W lo) x Q4(x hi) [+ Q4(W hi) x Q4(x lo)], the input given as an f16 plane plus
    its MX bundle.  Reference: float64 evaluation of exactly that sum (the FP4 operands dequantised on the host)."""
    import torch
    import torch.nn.functional as F
    from test_gpu_ops import _from_rows, _nhwc_rows, _spatial_op
    from vision_semantic_segmentation_amd import _lib
    from vision_semantic_segmentation_amd.network import (AVL_MX_IN_LO, AVL_MX_OUT_LO, OP_GCONV, fp4_quant_blocks, gconv_dense_windows,
                                                          mx_bundle_bytes, mx_dequant_fp4, mx_quant_fp4, pack_gconv_mx)
    H, W, width, s, d = case
    G = 32
    cg = width // G
    g = torch.Generator().manual_seed(H * 77 + W + width)
    x64 = torch.randn((1, width, H, W), generator=g, dtype=torch.float64) * torch.exp2(torch.randint(-2, 3, (1, width, 1, 1), generator=g).double())
    x_hi, x_lo = _split(x64)
    w64 = torch.randn((width, cg, 3, 3), generator=g, dtype=torch.float64) * (2.0 / (cg * 9)) ** 0.5
    b = torch.randn(width, generator=g) * 0.1
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    M_in = H * W
    rows_in = (M_in + 255) // 256 * 256
    # what the kernel multiplies: FP4 images of the activations per (pixel, 32-channel window) and of the weights per (row, tap, window)
    def deq_act(t):                                     # [1,C,H,W] -> same, through the MX bundle layout
        r = _pad_rows(t[0].permute(1, 2, 0).reshape(M_in, width).double(), rows_in)
        dq = mx_dequant_fp4(*mx_quant_fp4(r))[:M_in]
        return dq.reshape(H, W, width).permute(2, 0, 1).unsqueeze(0)
    w_hi, w_lo = _split(w64)

    def deq_w(part):                                    # weights quantised per (output channel, tap, window of 32 input channels)
        dense = gconv_dense_windows(part.double(), G)                                  # [win][32 co][9][32 ci]
        codes, sb = fp4_quant_blocks(dense)
        from vision_semantic_segmentation_amd.network import _FP4_GRID
        c = codes.to(torch.int64)
        code = torch.stack([c & 15, c >> 4], dim=-1).reshape(*dense.shape)
        val = _FP4_GRID[code & 7] * torch.where((code & 8) != 0, -1.0, 1.0) * torch.exp2(sb.double() - 127).unsqueeze(-1)
        out = torch.zeros_like(part, dtype=torch.float64).reshape(width, cg, 9)
        co = torch.arange(width)
        win, col = co // 32, co % 32
        gbase = (col // cg) * cg
        for ci in range(cg):
            out[co, ci, :] = val[win, col, :, gbase + ci]
        return out.reshape(width, cg, 3, 3)
    conv = lambda xx, ww: F.conv2d(xx, ww, None, stride=s, padding=d, dilation=d, groups=G)      # noqa: E731
    ref = conv(x_hi.double(), w_hi.double()) + conv(deq_act(x_hi), deq_w(w_lo)) + b.double().view(1, -1, 1, 1)
    if in_lo:
        ref = ref + conv(deq_act(x_lo), deq_w(w_hi))
    ref = F.relu(ref)
    src = _nhwc_rows(x_hi).to(cuda_device)
    in_mx = _bundle(_pad_rows(x_hi[0].permute(1, 2, 0).reshape(M_in, width), M_in), _pad_rows(x_lo[0].permute(1, 2, 0).reshape(M_in, width), M_in),
                    rows_in).to(cuda_device)
    rows = (OH * OW + 255) // 256 * 256
    dst = torch.full((rows, width), 7.0, dtype=torch.float16, device=cuda_device)
    out_mx = torch.zeros(2 * mx_bundle_bytes(rows, width), dtype=torch.uint8, device=cuda_device)
    frag, bundle = pack_gconv_mx(w64, G)
    wd, wb, bd = frag.reshape(-1).to(cuda_device), bundle.to(cuda_device), b.to(cuda_device)
    _run_plan([_spatial_op(OP_GCONV, _lib.AVL_F16, src, (H, W), width, dst, (OH, OW), width, weight=wd.data_ptr(), bias=bd.data_ptr(),
                           ksize=3, stride=s, pad=d, dil=d, groups=G, relu=1, w_layout=1, w_split=2, w_mx=wb.data_ptr(), in_mx=in_mx.data_ptr(),
                           out_mx=out_mx.data_ptr(), mx_flags=(AVL_MX_IN_LO if in_lo else 0) | AVL_MX_OUT_LO)])
    M = OH * OW
    v_lo = _unbundle(out_mx.cpu(), rows, width, 1)[2][:M]
    got = dst[:M].cpu().double() + v_lo
    want = ref[0].permute(1, 2, 0).reshape(M, width)
    err = float((got - want).abs().max() / want.abs().max())
    assert err <= 2 ** -11 * 0.3, "mx grouped conv %s in_lo=%s: %.3e" % (case, in_lo, err)
    assert torch.all(dst[M:] == 7.0)
