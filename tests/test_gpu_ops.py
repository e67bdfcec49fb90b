"""Single-op parity: every spatial kernel of the segmentation plan run ALONE through avl_seg_plan_* on
awkward shapes (sizes that are no multiple of any tile, dilations larger than the image, one-pixel
borders) and compared with the torch-CPU fp32 operator it replaces (F.conv2d / max_pool2d /
interpolate -- the calls oracle/network_oracle.py is made of).

Tolerances: fp32 ops 2e-6 of max|ref| (FMA-order only); 16-bit ops are compared with a reference
fed the SAME rounded inputs and weights, so only the fp32 accumulation order and the final rounding
to the 16-bit type differ: 2^-8 (bf16) / 2^-11 (f16) of max|ref| plus slack."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_plan(ops):
    import torch
    from vision_semantic_segmentation_amd import _lib
    from vision_semantic_segmentation_amd.network import AvlSegOp
    plan = C.c_void_p()
    arr = (AvlSegOp * len(ops))(*ops)
    _lib.check(_lib.lib().avl_seg_plan_create(arr, len(ops), C.byref(plan)), "avl_seg_plan_create")
    try:
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(_lib.lib().avl_seg_plan_run(plan, s), "avl_seg_plan_run")
        torch.cuda.synchronize()
    finally:
        _lib.lib().avl_seg_plan_destroy(plan)


def _dt(precision):
    import torch
    from vision_semantic_segmentation_amd import _lib
    return {"f32": (torch.float32, _lib.AVL_F32, 2e-6), "bf16": (torch.bfloat16, _lib.AVL_BF16, 2 ** -8 * 1.5),
            "f16": (torch.float16, _lib.AVL_F16, 2 ** -11 * 1.5)}[precision]


def _nhwc_rows(x, rows_pad=256):
    """[1,C,H,W] float -> [rows padded][C] NHWC matrix"""
    import torch
    _, c, h, w = x.shape
    m = x[0].permute(1, 2, 0).reshape(h * w, c)
    out = torch.zeros(((h * w + rows_pad - 1) // rows_pad * rows_pad, c), dtype=x.dtype)
    out[:h * w] = m
    return out


def _from_rows(buf, h, w, c):
    return buf[:h * w, :c].reshape(h, w, c).permute(2, 0, 1).unsqueeze(0)


def _spatial_op(kind, dtype_id, src, in_hw, cin, dst, out_hw, cout, **f):
    from vision_semantic_segmentation_amd.network import AvlSegOp
    op = AvlSegOp()
    op.kind, op.dtype = kind, dtype_id
    op.in_, op.out = src.data_ptr(), dst.data_ptr()
    op.in_h, op.in_w, op.in_c, op.in_ld, op.in_rows = in_hw[0], in_hw[1], cin, src.shape[1], src.shape[0]
    op.out_h, op.out_w, op.out_c, op.out_ld, op.out_rows = out_hw[0], out_hw[1], cout, dst.shape[1], dst.shape[0]
    for k, v in f.items():
        setattr(op, k, v)
    return op


@pytest.mark.parametrize("precision", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", [  # (H, W, C, dilation, pad)
    (37, 53, 64, 1, 0),            # decoder refine: padding 0, output shrinks by 2
    (37, 53, 128, 12, 12),         # ASPP: comb grid 4 x 5, ragged last comb row/column
    (20, 31, 2048, 24, 24),        # dilation larger than the image height: almost every tap is padding
    (135, 17, 64, 36, 36),
    (9, 9, 64, 2, 2), (3, 3, 64, 1, 1), (1, 70, 64, 6, 6),
])
def test_depthwise_conv(case, precision, cuda_device):
    import torch
    import torch.nn.functional as F
    from vision_semantic_segmentation_amd.network import OP_DWCONV
    H, W, Cc, d, pad = case
    tdt, did, tol = _dt(precision)
    g = torch.Generator().manual_seed(H * 1000 + W + d)
    x = torch.randn((1, Cc, H, W), generator=g).to(tdt)
    w = torch.randn((Cc, 1, 3, 3), generator=g) * 0.3
    b = torch.randn(Cc, generator=g) * 0.1
    OH, OW = H + 2 * pad - 2 * d, W + 2 * pad - 2 * d
    if OH <= 0 or OW <= 0:
        pytest.skip("empty output")
    wq = w.to(tdt).float() if precision != "f32" else w        # the 16-bit kernels round the folded weights to the activation type
    ref = F.relu(F.conv2d(x.float(), wq, b, padding=pad, dilation=d, groups=Cc))
    src = _nhwc_rows(x).to(cuda_device)
    dst = torch.full(((OH * OW + 255) // 256 * 256, Cc), 7.0, dtype=tdt, device=cuda_device)
    wd = w.reshape(Cc, 9).t().contiguous().reshape(-1).to(cuda_device)          # [tap][C] fp32
    bd = b.to(cuda_device)
    zero = torch.zeros(64, dtype=torch.uint8, device=cuda_device)
    _run_plan([_spatial_op(OP_DWCONV, did, src, (H, W), Cc, dst, (OH, OW), Cc, weight=wd.data_ptr(), bias=bd.data_ptr(),
                           in2=zero.data_ptr(), ksize=3, stride=1, pad=pad, dil=d, groups=Cc, relu=1)])
    got = _from_rows(dst.cpu().float(), OH, OW, Cc)
    err = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-6))
    assert err <= tol, "depthwise %s %s: %.3e" % (case, precision, err)
    assert torch.all(dst[OH * OW:] == 7.0)              # nothing written past the last pixel


@pytest.mark.parametrize("precision", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", [  # (H, W, width, stride, dilation)   groups = 32
    (23, 45, 128, 1, 1), (23, 45, 256, 2, 1), (30, 41, 512, 1, 2), (19, 67, 1024, 1, 4), (8, 8, 128, 1, 1), (5, 33, 256, 1, 4),
])
def test_grouped_conv(case, precision, cuda_device):
    import torch
    import torch.nn.functional as F
    from vision_semantic_segmentation_amd.network import OP_GCONV, pack_gconv_windows
    H, W, width, s, d = case
    G = 32
    cg = width // G
    tdt, did, tol = _dt(precision)
    g = torch.Generator().manual_seed(H * 77 + W + width)
    x = torch.randn((1, width, H, W), generator=g).to(tdt)
    w = torch.randn((width, cg, 3, 3), generator=g) * (2.0 / (cg * 9)) ** 0.5
    b = torch.randn(width, generator=g) * 0.1
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    wq = w.to(tdt).float() if precision != "f32" else w
    ref = F.relu(F.conv2d(x.float(), wq, b, stride=s, padding=d, dilation=d, groups=G))
    src = _nhwc_rows(x).to(cuda_device)
    dst = torch.full(((OH * OW + 255) // 256 * 256, width), 7.0, dtype=tdt, device=cuda_device)
    if precision == "f32":
        wd = w.reshape(G, cg, cg, 3, 3).permute(0, 3, 4, 2, 1).reshape(-1).contiguous().to(cuda_device)      # [g][ky][kx][ci][co]
        layout = 0
    else:
        wd = pack_gconv_windows(w.double(), G).to(tdt).to(cuda_device)
        layout = 1
    bd = b.to(cuda_device)
    _run_plan([_spatial_op(OP_GCONV, did, src, (H, W), width, dst, (OH, OW), width, weight=wd.data_ptr(), bias=bd.data_ptr(),
                           ksize=3, stride=s, pad=d, dil=d, groups=G, relu=1, w_layout=layout)])
    got = _from_rows(dst.cpu().float(), OH, OW, width)
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err <= tol * 2, "grouped conv %s %s: %.3e" % (case, precision, err)
    assert torch.all(dst[OH * OW:] == 7.0)


@pytest.mark.parametrize("precision", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("hw", [(33, 47), (64, 96), (7, 250), (224, 9)])
def test_stem_and_maxpool(hw, precision, cuda_device):
    import torch
    import torch.nn.functional as F
    from vision_semantic_segmentation_amd.network import OP_MAXPOOL, OP_STEM, pack_stem_mfma
    H, W = hw
    tdt, did, tol = _dt(precision)
    g = torch.Generator().manual_seed(H + W)
    img = torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8)
    w = torch.randn((64, 3, 7, 7), generator=g) * (2.0 / 147) ** 0.5
    b = torch.randn(64, generator=g) * 0.1
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    xn = (img.permute(2, 0, 1).unsqueeze(0).float() / 255 - mean) / std
    if precision != "f32":
        xn, wq = xn.to(tdt).float(), w.to(tdt).float()
    else:
        wq = w
    stem_ref = F.relu(F.conv2d(xn, wq, b, stride=2, padding=3))
    h2, w2 = stem_ref.shape[2:]
    pool_ref = F.max_pool2d(stem_ref.to(tdt).float(), 3, 2, 1)
    h4, w4 = pool_ref.shape[2:]
    imd = img.to(cuda_device)
    stem = torch.zeros(((h2 * w2 + 255) // 256 * 256, 64), dtype=tdt, device=cuda_device)
    pool = torch.zeros(((h4 * w4 + 255) // 256 * 256, 64), dtype=tdt, device=cuda_device)
    if precision == "f32":
        wd, layout = w.permute(2, 3, 1, 0).reshape(-1).contiguous().to(cuda_device), 0
    else:
        wd, layout = pack_stem_mfma(w.double()).to(tdt).to(cuda_device), 1
    bd = b.to(cuda_device)
    from vision_semantic_segmentation_amd.network import AvlSegOp
    op = AvlSegOp()
    op.kind, op.dtype = OP_STEM, did
    op.in_, op.out, op.weight, op.bias = imd.data_ptr(), stem.data_ptr(), wd.data_ptr(), bd.data_ptr()
    op.in_h, op.in_w, op.in_c, op.in_ld, op.in_rows = H, W, 3, 3, H * W
    op.out_h, op.out_w, op.out_c, op.out_ld, op.out_rows = h2, w2, 64, 64, stem.shape[0]
    op.ksize, op.stride, op.pad, op.dil, op.groups, op.relu, op.w_layout = 7, 2, 3, 1, 1, 1, layout
    mp = _spatial_op(OP_MAXPOOL, did, stem, (h2, w2), 64, pool, (h4, w4), 64, ksize=3, stride=2, pad=1, dil=1)
    _run_plan([op, mp])
    got = _from_rows(stem.cpu().float(), h2, w2, 64)
    err = float((got - stem_ref).abs().max() / stem_ref.abs().max())
    assert err <= tol * 2, "stem %s %s: %.3e" % (hw, precision, err)
    # the pool is exact given the stem output it was fed
    assert torch.equal(_from_rows(pool.cpu().float(), h4, w4, 64), F.max_pool2d(got, 3, 2, 1))


@pytest.mark.parametrize("precision", ["f32", "bf16"])
@pytest.mark.parametrize("case", [((17, 30), (68, 120)), ((5, 9), (33, 61)), ((1, 1), (4, 7)), ((135, 240), (270, 480))])
def test_bilinear_align_corners(case, precision, cuda_device):
    import torch
    import torch.nn.functional as F
    from vision_semantic_segmentation_amd.network import OP_BILINEAR
    (h, w), (oh, ow) = case
    Cc = 64
    tdt, did, tol = _dt(precision)
    x = torch.randn((1, Cc, h, w), generator=torch.Generator().manual_seed(h + ow)).to(tdt)
    ref = F.interpolate(x.float(), size=(oh, ow), mode="bilinear", align_corners=True)
    src = _nhwc_rows(x).to(cuda_device)
    dst = torch.zeros(((oh * ow + 255) // 256 * 256, Cc + 8), dtype=tdt, device=cuda_device)      # a column slice of a wider buffer
    _run_plan([_spatial_op(OP_BILINEAR, did, src, (h, w), Cc, dst, (oh, ow), Cc)])
    got = _from_rows(dst.cpu().float(), oh, ow, Cc)
    err = float((got - ref).abs().max() / ref.abs().max())
    # fp32: the weight is fract(scale * i); the rounding of scale * i (|i| up to 270) is amplified by that, so two
    # correct fp32 implementations differ by ~1e-5 of the value range here
    assert err <= max(tol, 3e-5), "bilinear %s %s: %.3e" % (case, precision, err)
    assert float(dst[:, Cc:].abs().max()) == 0.0                     # the neighbouring columns are untouched


def _experiments_build():
    """True when AVL_HIP_LIB points at libavl_hip_exp.so (`make experiments`): only that build holds the experiment kernels."""
    import os
    return os.path.basename(os.environ.get("AVL_HIP_LIB", "")).startswith("libavl_hip_exp")


# layout 5 = k_gemm_w4, the one-wave-per-SIMD experiment (16-bit, N % 256 == 0): compiled into the experiments build only
@pytest.mark.parametrize("layout", [0, 5] if _experiments_build() else [0])
@pytest.mark.parametrize("precision", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", [  # (M, K, N, residual, relu)
    (1000, 64, 128, False, True), (777, 256, 64, False, True), (2600, 128, 256, True, True), (50000, 512, 256, False, False),
    (300, 2048, 1024, True, True), (4097, 1024, 512, False, True), (65, 64, 19, False, False),
])
def test_pointwise_gemm(case, precision, layout, cuda_device):
    import torch
    if layout == 5 and (precision == "f32" or case[2] % 256):
        pytest.skip("k_gemm_w4 takes 16-bit operands and N % 256 == 0")
    from vision_semantic_segmentation_amd.network import OP_GEMM, AvlSegOp
    M, K, N, res, relu = case
    tdt, did, tol = _dt(precision)
    g = torch.Generator().manual_seed(M + K + N)
    Mp, Np = (M + 255) // 256 * 256, (N + 255) // 256 * 256
    a = torch.zeros((Mp, K), dtype=tdt)
    a[:M] = torch.randn((M, K), generator=g).to(tdt)
    w = torch.zeros((Np, K), dtype=tdt)
    w[:N] = (torch.randn((N, K), generator=g) / K ** 0.5).to(tdt)
    b = torch.zeros(Np)
    b[:N] = torch.randn(N, generator=g)
    r = torch.randn((Mp, N), generator=g).to(tdt) if res else None
    out_f32 = N == 19                                   # the classifier writes fp32 logits
    ref = a[:M].float() @ w[:N].float().t() + b[:N]
    if res:
        ref = ref + r[:M].float()
    if relu:
        ref = torch.relu(ref)
    ad, wd, bd = a.to(cuda_device), w.to(cuda_device), b.to(cuda_device)
    out = torch.full((Mp, N), 7.0, dtype=torch.float32 if out_f32 else tdt, device=cuda_device)
    op = AvlSegOp()
    op.kind, op.dtype = OP_GEMM, did
    op.in_, op.out, op.weight, op.bias = ad.data_ptr(), out.data_ptr(), wd.data_ptr(), bd.data_ptr()
    op.in_h, op.in_w, op.in_c, op.in_ld, op.in_rows = 1, M, K, K, Mp
    op.out_h, op.out_w, op.out_c, op.out_ld, op.out_rows = 1, M, N, N, Mp
    op.relu, op.out_f32, op.w_rows, op.ksize, op.stride, op.dil, op.groups = int(relu), int(out_f32), Np, 1, 1, 1, 1
    op.w_layout = layout
    if res:
        rd = r.to(cuda_device)
        op.in2, op.in2_ld = rd.data_ptr(), N
    _run_plan([op])
    got = out[:M].cpu().float()
    err = float((got - ref).abs().max() / ref.abs().max())
    bar = 2e-6 * max(1, K // 64) if precision == "f32" else (tol if not out_f32 else 1e-5)
    assert err <= bar, "gemm %s %s: %.3e" % (case, precision, err)
    assert torch.all(out[M:] == 7.0)                    # rows past M stay untouched


@pytest.mark.parametrize("precision", ["bf16", "f16"])
@pytest.mark.parametrize("case", [  # (H, W, K, N, dilation)
    (37, 53, 128, 256, 12), (20, 31, 2048, 256, 24), (135, 17, 64, 256, 36), (9, 9, 64, 48, 2), (16, 16, 256, 512, 1),
    (37, 53, 512, 256, 1, 0), (20, 31, 256, 256, 1, 0), (21, 40, 64, 256, 3, 1),       # padding != dilation: the decoder's pad-0 blocks
])
def test_fused_depthwise_pointwise(case, precision, cuda_device):
    """AVL_OP_DWPW == AVL_OP_DWCONV followed by AVL_OP_GEMM, bit for bit (same tap pairing, same K order), and both
    within the 16-bit tolerance of the torch fp32 operators."""
    import torch
    import torch.nn.functional as F
    from vision_semantic_segmentation_amd.network import OP_DWCONV, OP_DWPW, OP_GEMM, AvlSegOp, dwpw_tile_order, pack_dw_pairs
    H, W, K, N, d = case[:5]
    pad = case[5] if len(case) > 5 else d
    OH, OW = H + 2 * pad - 2 * d, W + 2 * pad - 2 * d
    tdt, did, tol = _dt(precision)
    g = torch.Generator().manual_seed(H * 31 + W + K + d)
    x = torch.randn((1, K, H, W), generator=g).to(tdt)
    w1 = torch.randn((K, 1, 3, 3), generator=g) * 0.3
    b1 = torch.randn(K, generator=g) * 0.1
    w2 = torch.randn((N, K), generator=g) / K ** 0.5
    b2 = torch.randn(N, generator=g) * 0.1
    M = OH * OW
    Mp, Np = (M + 255) // 256 * 256, (N + 255) // 256 * 256
    src = _nhwc_rows(x).to(cuda_device)
    zero = torch.zeros(64, dtype=torch.uint8, device=cuda_device)
    w1d = w1.reshape(K, 9).t().contiguous().reshape(-1).to(cuda_device)
    b1d = b1.to(cuda_device)
    w2p = torch.zeros((Np, K), dtype=tdt)
    w2p[:N] = w2.to(tdt)
    b2p = torch.zeros(Np)
    b2p[:N] = b2
    w2d, b2d = w2p.to(cuda_device), b2p.to(cuda_device)
    mid = torch.zeros((Mp, K), dtype=tdt, device=cuda_device)
    out_a = torch.full((Mp, N + 16), 7.0, dtype=tdt, device=cuda_device)      # written as a column slice of a wider buffer
    out_b = torch.full((Mp, N + 16), 7.0, dtype=tdt, device=cuda_device)
    dw = _spatial_op(OP_DWCONV, did, src, (H, W), K, mid, (OH, OW), K, weight=w1d.data_ptr(), bias=b1d.data_ptr(), in2=zero.data_ptr(),
                     ksize=3, stride=1, pad=pad, dil=d, groups=K, relu=1)
    pw = AvlSegOp()
    pw.kind, pw.dtype = OP_GEMM, did
    pw.in_, pw.out, pw.weight, pw.bias = mid.data_ptr(), out_a.data_ptr() + 16 * out_a.element_size(), w2d.data_ptr(), b2d.data_ptr()
    pw.in_h, pw.in_w, pw.in_c, pw.in_ld, pw.in_rows = OH, OW, K, K, Mp
    pw.out_h, pw.out_w, pw.out_c, pw.out_ld, pw.out_rows = OH, OW, N, N + 16, Mp
    pw.relu, pw.w_rows, pw.ksize, pw.stride, pw.dil, pw.groups = 1, Np, 1, 1, 1, 1
    _run_plan([dw, pw])
    params = torch.cat([pack_dw_pairs(w1.double(), b1.double(), tdt), dwpw_tile_order(OH, OW, d)]).to(cuda_device)
    fused = AvlSegOp()
    fused.kind, fused.dtype = OP_DWPW, did
    fused.in_, fused.in2, fused.out = src.data_ptr(), params.data_ptr(), out_b.data_ptr() + 16 * out_b.element_size()
    fused.weight, fused.bias = w2d.data_ptr(), b2d.data_ptr()
    fused.in_h, fused.in_w, fused.in_c, fused.in_ld, fused.in_rows = H, W, K, K, src.shape[0]
    fused.out_h, fused.out_w, fused.out_c, fused.out_ld, fused.out_rows = OH, OW, N, N + 16, Mp
    fused.relu, fused.w_rows, fused.ksize, fused.stride, fused.pad, fused.dil, fused.groups = 1, Np, 3, 1, pad, d, K
    _run_plan([fused])
    assert torch.equal(out_a, out_b), "fused and unfused differ: max %g" % float((out_a.float() - out_b.float()).abs().max())
    assert torch.all(out_b[:, :16] == 7.0) and torch.all(out_b[M:] == 7.0)       # neighbours and rows past M untouched
    a = F.relu(F.conv2d(x.float(), w1.to(tdt).float(), b1, padding=pad, dilation=d, groups=K)).to(tdt).float()
    ref = F.relu(F.conv2d(a, w2.to(tdt).float().view(N, K, 1, 1), b2))
    got = _from_rows(out_b[:, 16:].cpu().float(), OH, OW, N)
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err <= tol * 2, "dwpw %s %s: %.3e" % (case, precision, err)


@pytest.mark.parametrize("case", [(180, 240, 2048, 12, 12), (270, 480, 512, 1, 0)])
def test_fused_depthwise_pointwise_repeats_under_load(case, cuda_device):
    """Race screen for AVL_OP_DWPW: the same launch, back to back with copies in between (a busy queue, like a graph
    replay), must give the same bytes every time.  An earlier version that waited for inline-asm loads by hand passed
    every parity test and was wrong a few times in a thousand launches at sizes with more tiles than CUs."""
    import torch
    from vision_semantic_segmentation_amd import _lib
    from vision_semantic_segmentation_amd.network import OP_DWPW, AvlSegOp, dwpw_tile_order, pack_dw_pairs
    H, W, K, d, pad = case
    N = 256
    g = torch.Generator().manual_seed(1)
    OH, OW = H + 2 * pad - 2 * d, W + 2 * pad - 2 * d
    M, Mi = OH * OW, H * W
    Mp, Mip = (M + 255) // 256 * 256, (Mi + 255) // 256 * 256
    x = torch.zeros((Mip, K), dtype=torch.bfloat16)
    x[:Mi] = torch.randn((Mi, K), generator=g).to(torch.bfloat16)
    w1, b1 = torch.randn((K, 1, 3, 3), generator=g).double() * 0.3, torch.randn(K, generator=g).double() * 0.1
    xd = x.to(cuda_device)
    w2d = (torch.randn((N, K), generator=g) / K ** 0.5).to(torch.bfloat16).to(cuda_device)
    b2d = torch.randn(N, generator=g).to(cuda_device)
    params = torch.cat([pack_dw_pairs(w1, b1, torch.bfloat16), dwpw_tile_order(OH, OW, d)]).to(cuda_device)
    out = torch.zeros((Mp, N), dtype=torch.bfloat16, device=cuda_device)
    op = AvlSegOp()
    op.kind, op.dtype = OP_DWPW, _lib.AVL_BF16
    op.in_, op.in2, op.out, op.weight, op.bias = xd.data_ptr(), params.data_ptr(), out.data_ptr(), w2d.data_ptr(), b2d.data_ptr()
    op.in_h, op.in_w, op.in_c, op.in_ld, op.in_rows = H, W, K, K, Mip
    op.out_h, op.out_w, op.out_c, op.out_ld, op.out_rows = OH, OW, N, N, Mp
    op.relu, op.w_rows, op.ksize, op.stride, op.pad, op.dil, op.groups = 1, N, 3, 1, pad, d, K
    plan = C.c_void_p()
    _lib.check(_lib.lib().avl_seg_plan_create((AvlSegOp * 1)(op), 1, C.byref(plan)), "avl_seg_plan_create")
    try:
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(_lib.lib().avl_seg_plan_run(plan, s), "avl_seg_plan_run")
        torch.cuda.synchronize()
        ref = out.clone()
        outs = [torch.zeros_like(out) for _ in range(4)]
        for i in range(120):
            _lib.lib().avl_seg_plan_run(plan, s)
            outs[i % 4].copy_(out)
            if i % 4 == 3:
                torch.cuda.synchronize()
                for o in outs:
                    assert torch.equal(o, ref), "launch ~%d differs from the first one" % i
    finally:
        _lib.lib().avl_seg_plan_destroy(plan)


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_classifier_gemm_writes_the_argmax(precision, cuda_device):
    """round 5: torch.argmax (semantic_segmentation.py:56) in the classifier GEMM's epilogue (out_f32 with out_mx = uint8 labels): the
    labels are the arg-max of the very logits stored, the FIRST maximal index wins on ties -- inside one lane's 16 channels and across
    the two lanes of a row (classes 0-15 | 16-18) --, rows past M stay untouched"""
    import torch
    from vision_semantic_segmentation_amd import _lib
    from vision_semantic_segmentation_amd.network import OP_GEMM, AvlSegOp
    tdt, did, _ = _dt(precision)
    M, K, N = 1000, 256, 19
    Mp = 1024
    g = torch.Generator().manual_seed(5)
    a = torch.zeros((Mp, K), dtype=tdt)
    a[:M] = torch.randn((M, K), generator=g).to(tdt)
    w = torch.zeros((64, K), dtype=tdt)
    w[:N] = (torch.randn((N, K), generator=g) * 0.05).to(tdt)
    w[7] = w[3]                     # ties inside the first lane's block ...
    w[17] = w[16]                   # ... inside the second ...
    w[18] = w[5]                    # ... and across the two
    b = torch.zeros(64)
    b[:N] = torch.randn(N, generator=g) * 0.01
    b[7], b[17], b[18] = b[3], b[16], b[5]
    ad, wd, bd = a.to(cuda_device), w.to(cuda_device), b.to(cuda_device)
    logits = torch.full((Mp, N), -7.0, dtype=torch.float32, device=cuda_device)
    labels = torch.full((Mp,), 99, dtype=torch.uint8, device=cuda_device)
    op = AvlSegOp()
    op.kind, op.dtype = OP_GEMM, did
    op.in_, op.out, op.weight, op.bias, op.out_mx = ad.data_ptr(), logits.data_ptr(), wd.data_ptr(), bd.data_ptr(), labels.data_ptr()
    op.in_h, op.in_w, op.in_c, op.in_ld, op.in_rows = 1, M, K, K, Mp
    op.out_h, op.out_w, op.out_c, op.out_ld, op.out_rows = 1, M, N, N, Mp
    op.relu, op.out_f32, op.w_rows, op.ksize, op.stride, op.dil, op.groups = 0, 1, 64, 1, 1, 1, 1
    _run_plan([op])
    lg = logits[:M].cpu()
    want = torch.argmax(lg, dim=1)
    # torch.argmax returns the first maximal index on CPU
    assert torch.equal(labels[:M].cpu().long(), want)
    assert torch.all(labels[M:] == 99) and torch.all(logits[M:] == -7.0)
    assert bool(((lg[:, 3] == lg[:, 7]) & (lg[:, 16] == lg[:, 17]) & (lg[:, 5] == lg[:, 18])).all())       # the ties are real
    assert set(want.tolist()) & {3, 16, 5} and not (set(want.tolist()) & {7, 17, 18})
