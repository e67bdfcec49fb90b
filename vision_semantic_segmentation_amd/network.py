"""DeepLabV3+ / ResNeXt-50 (output stride 8) as a flat program for libavl_hip.so.

The reference assembles the model from torch modules:
  backbone  torchvision ResNet(Bottleneck,[3,4,6,3], groups=32, width_per_group=4,
            replace_stride_with_dilation=(False,True,True))      backbone/resnet.py:8-43, build.py:14-20
  ASPP      1x1 | 3 x (depthwise 3x3 d=12/24/36 + 1x1) | image pooling ; concat ; 1x1       aspp.py:16-95
  decoder   1x1 on layer1 ; bilinear x2 ; concat ; 2 x (depthwise 3x3 pad 0 + 1x1) ; 1x1    decoder.py:10-51
This module keeps the reference's checkpoint format (``{'model': state_dict}`` with ``module.``
prefixed keys, core/utils/checkpoint.py:52) and turns a state dict into what the HIP kernels eat:
BatchNorm folded into the preceding convolution (eval mode, eps 1e-5), NHWC activations, 1x1
weights as [Cout][Cin] rows padded to the GEMM tile, grouped / depthwise / stem weights re-laid for
their kernels, and a list of ``avl_seg_op`` records.

Two graph-level rewrites, both exact in real arithmetic:
  * torch.cat is free: producers write into channel slices of one wide buffer (out_ld);
  * the image-pooling branch is constant over the image after its bilinear "upsample" of a 1x1 map
    (aspp.py:86-89), so its share of the 1280->256 projection is a per-channel constant:
    it becomes a bias vector computed per frame by two small GEMVs.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib

BN_EPS = 1e-5

OP_STEM, OP_MAXPOOL, OP_GEMM, OP_GCONV, OP_DWCONV, OP_BILINEAR, OP_GAP, OP_GEMV, OP_ARGMAX, OP_SUBSAMPLE, OP_DWPW, OP_BOTTLENECK = range(1, 13)
OP_NAMES = {1: "stem", 2: "maxpool", 3: "gemm", 4: "gconv", 5: "dwconv", 6: "bilinear", 7: "gap", 8: "gemv", 9: "argmax", 10: "subsample", 11: "dwpw",
            12: "bottleneck"}


class AvlSegOp(C.Structure):
    """struct avl_seg_op of include/avl_hip.h"""
    _fields_ = [
        ("kind", C.c_int32), ("dtype", C.c_int32),
        ("in_", C.c_void_p), ("in2", C.c_void_p), ("out", C.c_void_p), ("weight", C.c_void_p), ("bias", C.c_void_p),
        ("in_h", C.c_int32), ("in_w", C.c_int32), ("in_c", C.c_int32), ("in_ld", C.c_int32), ("in_rows", C.c_int32),
        ("out_h", C.c_int32), ("out_w", C.c_int32), ("out_c", C.c_int32), ("out_ld", C.c_int32), ("out_rows", C.c_int32),
        ("in2_ld", C.c_int32),
        ("ksize", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("dil", C.c_int32), ("groups", C.c_int32),
        ("relu", C.c_int32), ("out_f32", C.c_int32), ("w_rows", C.c_int32), ("w_layout", C.c_int32), ("w_split", C.c_int32), ("mx_flags", C.c_int32),
        ("in_lo", C.c_void_p), ("in2_lo", C.c_void_p), ("out_lo", C.c_void_p),
        ("w_mx", C.c_void_p), ("in_mx", C.c_void_p), ("out_mx", C.c_void_p), ("in2_mx", C.c_void_p),
        ("in3", C.c_void_p), ("in3_mx", C.c_void_p), ("in3_c", C.c_int32), ("in3_ld", C.c_int32),
    ]


AVL_MX_IN_LO, AVL_MX_RES_LO, AVL_MX_OUT_LO = 1, 2, 4


_vp, _i = C.c_void_p, C.c_int
_lib._register_seg({
    "avl_seg_plan_create": (_i, [C.POINTER(AvlSegOp), _i, C.POINTER(C.c_void_p)]),
    "avl_seg_plan_destroy": (None, [_vp]),
    "avl_seg_plan_run": (_i, [_vp, _vp]),
    "avl_seg_plan_capture": (_i, [_vp, _vp]),
    "avl_seg_plan_profile": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "avl_seg_plan_num_ops": (_i, [_vp]),
    "avl_seg_plan_nonfinite": (_i, [_vp, _vp, _vp]),
})

# ----------------------------------------------------------------------------------------------
# state dict: names and shapes of the reference checkpoint
# ----------------------------------------------------------------------------------------------

LAYERS = (3, 4, 6, 3)
PLANES = (64, 128, 256, 512)
GROUPS, WIDTH_PER_GROUP, EXPANSION = 32, 4, 4


def _bn_keys(prefix, c):
    return [(prefix + ".weight", (c,)), (prefix + ".bias", (c,)), (prefix + ".running_mean", (c,)),
            (prefix + ".running_var", (c,))]


def state_spec(num_classes=19, in_channels=3, aspp_out=256, atrous_channels=(256, 256, 256, 256), low_level_out=256,
               refine_channels=(256, 256)):
    """[(key, shape)] of DeepLabV3Plus.state_dict() for MODEL.BACKBONE = resnext50_32x4d (keys as
    saved by the reference, without the DataParallel 'module.' prefix)."""
    spec = [("backbone.conv1.weight", (64, in_channels, 7, 7))] + _bn_keys("backbone.bn1", 64)
    inplanes = 64
    for li, (planes, nblocks) in enumerate(zip(PLANES, LAYERS), start=1):
        width = int(planes * (WIDTH_PER_GROUP / 64.0)) * GROUPS
        for b in range(nblocks):
            p = "backbone.layer%d.%d" % (li, b)
            spec += [(p + ".conv1.weight", (width, inplanes, 1, 1))] + _bn_keys(p + ".bn1", width)
            spec += [(p + ".conv2.weight", (width, width // GROUPS, 3, 3))] + _bn_keys(p + ".bn2", width)
            spec += [(p + ".conv3.weight", (planes * EXPANSION, width, 1, 1))] + _bn_keys(p + ".bn3", planes * EXPANSION)
            if b == 0:
                spec += [(p + ".downsample.0.weight", (planes * EXPANSION, inplanes, 1, 1))] + _bn_keys(p + ".downsample.1", planes * EXPANSION)
            inplanes = planes * EXPANSION
    feat, low = inplanes, PLANES[0] * EXPANSION
    spec += [("aspp.module_pyramid.0.conv.weight", (atrous_channels[0], feat, 1, 1))] + _bn_keys("aspp.module_pyramid.0.bn", atrous_channels[0])
    for i in range(1, len(atrous_channels)):
        p = "aspp.module_pyramid.%d" % i
        spec += [(p + ".depthwise_cnn.conv.weight", (feat, 1, 3, 3))] + _bn_keys(p + ".depthwise_cnn.bn", feat)
        spec += [(p + ".pointwise_cnn.conv.weight", (atrous_channels[i], feat, 1, 1))] + _bn_keys(p + ".pointwise_cnn.bn", atrous_channels[i])
    spec += [("aspp.global_avg_pool.1.conv.weight", (256, feat, 1, 1))] + _bn_keys("aspp.global_avg_pool.1.bn", 256)
    spec += [("aspp.conv.conv.weight", (aspp_out, sum(atrous_channels) + 256, 1, 1))] + _bn_keys("aspp.conv.bn", aspp_out)
    spec += [("decoder.low_level_conv.conv.weight", (low_level_out, low, 1, 1))] + _bn_keys("decoder.low_level_conv.bn", low_level_out)
    cin = low_level_out + aspp_out
    for i, rc in enumerate(refine_channels):
        p = "decoder.refine_layers.%d" % i
        spec += [(p + ".depthwise_cnn.conv.weight", (cin, 1, 3, 3))] + _bn_keys(p + ".depthwise_cnn.bn", cin)
        spec += [(p + ".pointwise_cnn.conv.weight", (rc, cin, 1, 1))] + _bn_keys(p + ".pointwise_cnn.bn", rc)
        cin = rc
    p = "decoder.refine_layers.%d" % len(refine_channels)
    spec += [(p + ".conv.weight", (num_classes, cin, 1, 1)), (p + ".conv.bias", (num_classes,))]
    return spec


def random_state_dict(seed=0, **kw):
    """Seeded random weights of the right shapes (there are no trained weights offline): Kaiming-normal
    convolutions, and NON-trivial BatchNorm statistics so that folding is exercised.  The gains are
    chosen so activations keep O(1) scale through all 50+ layers."""
    g = torch.Generator().manual_seed(seed)
    st = {}
    for key, shape in state_spec(**kw):
        if key.endswith("running_var"):
            t = torch.rand(shape, generator=g) * 0.5 + 0.75
        elif key.endswith("running_mean"):
            t = torch.randn(shape, generator=g) * 0.1
        elif key.endswith(".weight") and len(shape) == 1:
            t = torch.rand(shape, generator=g) * 0.4 + 0.8
            if ".bn3." in key or "downsample.1" in key:
                t = t * 0.6          # keep the residual sum from growing block after block
        elif key.endswith(".bias"):
            t = torch.randn(shape, generator=g) * 0.05
        else:
            fan_in = shape[1] * shape[2] * shape[3]
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        st[key] = t.to(torch.float32)
    return st


def load_checkpoint(path):
    """semantic_segmentation.py:31-32: torch.load(...).pop('model'); keys carry 'module.' (DataParallel).
    Loaded with weights_only=True -- nothing in the file is executed."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    state = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt else ckpt
    return {(k[7:] if k.startswith("module.") else k): v.to(torch.float32) for k, v in state.items()
            if not k.endswith("num_batches_tracked")}


def check_state_dict(state, **kw):
    missing = [k for k, s in state_spec(**kw) if k not in state or tuple(state[k].shape) != tuple(s)]
    if missing:
        raise KeyError("state dict lacks / mis-shapes %d tensors, e.g. %s" % (len(missing), missing[:3]))
    # the kernels' ReLU is a max with 0, which turns a NaN into 0 (torch's keeps it): a NaN weight would vanish silently
    bad = [k for k, _ in state_spec(**kw) if not bool(torch.isfinite(state[k]).all())]
    if bad:
        raise ValueError("state dict holds Inf / NaN in %d tensors, e.g. %s" % (len(bad), bad[:3]))


# ----------------------------------------------------------------------------------------------
# folding + packing
# ----------------------------------------------------------------------------------------------

def fold_bn(state, conv_key, bn_prefix):
    """conv -> BN(eval) == conv with w*s and bias b - mean*s, s = gamma/sqrt(var+eps).  float64 on the host."""
    w = state[conv_key].to(torch.float64)
    if bn_prefix is None:
        b = state.get(conv_key[:-6] + "bias")
        return w, (b.to(torch.float64) if b is not None else torch.zeros(w.shape[0], dtype=torch.float64))
    s = state[bn_prefix + ".weight"].to(torch.float64) / torch.sqrt(state[bn_prefix + ".running_var"].to(torch.float64) + BN_EPS)
    b = state[bn_prefix + ".bias"].to(torch.float64) - state[bn_prefix + ".running_mean"].to(torch.float64) * s
    return w * s.view(-1, 1, 1, 1), b


def pack_stem_mfma(w):
    """Stem weights [64][3][7][7] (BN folded) -> [nj 4][step 6][i 16][k 32] for the MFMA stem kernel: MFMA row i of
    n-tile nj is output channel (i>>2)*16 + nj*4 + (i&3); K position (step, k) is chunk q = step*4 + k//8 of 8 taps:
    kernel row ky = q//3, tap t = (q%3)*8 + k%8 of that row's 24 slots (kx = t//3, ci = t%3; slots 21..23 and q >= 21 are zero)."""
    out = torch.zeros((4, 6, 16, 32), dtype=torch.float64)
    for nj in range(4):
        for i in range(16):
            n = (i >> 2) * 16 + nj * 4 + (i & 3)
            for st in range(6):
                for k in range(32):
                    q, j = st * 4 + k // 8, k % 8
                    t = (q % 3) * 8 + j
                    if q < 21 and t < 21:
                        out[nj, st, i, k] = w[n, t % 3, q // 3, t // 3]
    return out.reshape(-1)


def pack_gconv_windows(w, groups):
    """Grouped 3x3 weights [C][C/groups][3][3] (BN folded) -> dense block-diagonal 32-channel windows for the
    MFMA kernel: [window][nj 2][tap 9][i 16][ci 32] where MFMA row i of n-tile nj is output channel
    (i>>2)*8 + nj*4 + (i&3) of the window and ci is the input channel inside the window (zero when the two
    channels belong to different groups)."""
    C_ = w.shape[0]
    cg = C_ // groups
    nwin = C_ // 32
    dense = torch.zeros((nwin, 32, 9, 32), dtype=torch.float64)              # [window][co_local][tap][ci_local]
    wt = w.reshape(C_, cg, 9)
    co = torch.arange(C_)
    win, col = co // 32, co % 32
    gbase = (col // cg) * cg                                                   # first window-local channel of co's group
    for ci in range(cg):
        dense[win, col, :, gbase + ci] = wt[co, ci, :]
    i = torch.arange(16)
    out = torch.empty((nwin, 2, 9, 16, 32), dtype=torch.float64)
    for nj in range(2):
        rows = (i >> 2) * 8 + nj * 4 + (i & 3)
        out[:, nj] = dense[:, rows].permute(0, 2, 1, 3)                        # [win][tap][i][ci]
    return out.reshape(-1)


def gconv_dense_windows(w, groups):
    """Grouped 3x3 weights [C][C/groups][3][3] -> dense block-diagonal windows float64 [window][co_local 32][tap 9][ci_local 32]."""
    C_ = w.shape[0]
    cg = C_ // groups
    nwin = C_ // 32
    dense = torch.zeros((nwin, 32, 9, 32), dtype=torch.float64)
    wt = w.reshape(C_, cg, 9).to(torch.float64)
    co = torch.arange(C_)
    win, col = co // 32, co % 32
    gbase = (col // cg) * cg
    for ci in range(cg):
        dense[win, col, :, gbase + ci] = wt[co, ci, :]
    return dense


def pack_gconv_mx(w, groups):
    """The MX grouped conv's weights (k_gconv_mx): float64 [C][C/groups][3][3] (BN folded) ->
      hi fragments  float16 [window][nj 2][tap 9][i 16][ci 32]                  (as pack_gconv_windows, of the f16 hi part)
      FP4 bundle    uint8: [window][nj 2][pair 2][g 3][lane 64][16 B] then [window][lane 64][12 B] scale bytes
    pair 0 = Q4(W lo) (multiplies Q4(x hi)), pair 1 = Q4(W hi) (multiplies Q4(x lo)); scaled MFMA g covers taps 4g .. 4g+3 (K block
    kb = tap 4g + kb, its 32 values the window's input channels; taps >= 9 are zero); lane = kb * 16 + i, MFMA row i of n-tile nj
    being output channel (i>>2)*8 + nj*4 + (i&3) of the window; scale byte index nj * 6 + pair * 3 + g."""
    whi, wlo = split_f16(w)
    nwin = w.shape[0] // 32
    frag_hi = pack_gconv_windows(whi.to(torch.float64), groups).to(torch.float16)
    i = torch.arange(16)
    w4 = torch.zeros((nwin, 2, 2, 3, 64, 16), dtype=torch.uint8)
    w4s = torch.zeros((nwin, 64, 12), dtype=torch.uint8)
    for pair, part in ((0, wlo), (1, whi)):
        dense = gconv_dense_windows(part.to(torch.float64), groups)                        # [win][32][9][32]
        for nj in range(2):
            rows = (i >> 2) * 8 + nj * 4 + (i & 3)
            d = torch.zeros((nwin, 16, 12, 32), dtype=torch.float64)
            d[:, :, :9] = dense[:, rows]
            codes, sbyte = fp4_quant_blocks(d.reshape(nwin, 16, 3, 4, 32))                 # [win][i][g][kb][16], [win][i][g][kb]
            w4[:, nj, pair] = codes.permute(0, 2, 3, 1, 4).reshape(nwin, 3, 64, 16)         # lane = kb * 16 + i
            sb = sbyte.permute(0, 3, 1, 2).reshape(nwin, 64, 3).to(torch.uint8)             # [win][lane][g]
            w4s[:, :, nj * 6 + pair * 3:nj * 6 + pair * 3 + 3] = sb
    return frag_hi, torch.cat([w4.reshape(-1), w4s.reshape(-1)])


def pack_dw_f32(w, b):
    """Depthwise parameters for the exact depthwise stage (w_split = 3 of AVL_OP_DWPW: k_dwpw_x, k_dwpw_xs; fp32 depthwise weights): w float64 [C][1][3][3]
    and b [C] (BN folded) -> the int32 bit patterns of float32 [C/64][chunk 8][row 10][8]: rows 0 .. 8 = tap t of the chunk's eight
    channels, row 9 = the bias."""
    c = w.shape[0]
    assert c % 64 == 0
    out = torch.empty((c // 64, 8, 10, 8), dtype=torch.float32)
    out[:, :, 0:9, :] = w.reshape(c // 64, 8, 8, 9).permute(0, 1, 3, 2).to(torch.float32)
    out[:, :, 9, :] = b.to(torch.float32).reshape(c // 64, 8, 8)
    return out.view(torch.int32).reshape(-1)


def pack_dw_pairs(w, b, act_dtype):
    """Depthwise parameters for the fused depthwise+pointwise kernel (AVL_OP_DWPW): w float64 [C][1][3][3] and b [C]
    (BN folded) -> int32 [C/64][chunk 8][6][8]: five tap pairs per channel (tap 2p in the low half, tap 2p+1 in the
    high half, rounded fp32 -> activation type exactly as k_dwconv does in-kernel; the ninth tap pairs with zero) and
    the fp32 bias bits."""
    c = w.shape[0]
    assert c % 64 == 0
    w9 = w.reshape(c, 9).to(torch.float32)                      # the unfused kernel receives fp32 and rounds from there
    w16 = torch.cat([w9, torch.zeros((c, 1), dtype=torch.float32)], dim=1).to(act_dtype)          # [C][10]
    bits = w16.view(torch.int16).to(torch.int32) & 0xFFFF                                          # raw 16-bit patterns
    pairs = bits[:, 0::2] | (bits[:, 1::2] << 16)                                                   # [C][5]
    bias_bits = b.to(torch.float32).view(torch.int32)                                              # [C]
    out = torch.empty((c // 64, 8, 6, 8), dtype=torch.int32)
    pc = pairs.reshape(c // 64, 8, 8, 5)                                                            # [step][chunk][ch][pair]
    out[:, :, :5, :] = pc.permute(0, 1, 3, 2)
    out[:, :, 5, :] = bias_bits.reshape(c // 64, 8, 8)
    return out.reshape(-1)


def dwpw_tile_order(h, w, dilation, tile=128):
    """Visiting order of the 128-pixel tiles for AVL_OP_DWPW: sorted by the tile centre's position inside a period of
    `dilation` image rows, so that tiles whose rows differ by a multiple of the dilation (they read the same input rows)
    are neighbours and end up in flight on the same XCD."""
    n = (h * w + tile - 1) // tile
    period = dilation * w
    keys = sorted(range(n), key=lambda t: ((tile * t + tile // 2) % period, t))
    return torch.tensor(keys, dtype=torch.int32)


def dwpw_block_order(h, w, band=4):
    """Visiting order of the 8 x 16-pixel tiles of AVL_OP_DWPW with w_layout = 1 (stride-1 taps: the decoder's refine blocks): bands
    of `band` block rows, walked column by column, so that the ~32 tiles an XCD has in flight form a 4 x 8 patch whose
    neighbours share their one-pixel halo in that XCD's L2 in both directions."""
    ty, tx = (h + 7) // 8, (w + 15) // 16
    order = [r * tx + c for b in range(0, ty, band) for c in range(tx) for r in range(b, min(b + band, ty))]
    return torch.tensor(order, dtype=torch.int32)


def _round_up(x, m):
    return (x + m - 1) // m * m


def split_f16(w):
    """float64 tensor -> (hi, lo) float16 with hi = f16(w), lo = f16(w - hi): hi + lo keeps ~22 significant bits."""
    hi = w.to(torch.float16)
    lo = (w - hi.to(torch.float64)).to(torch.float16)
    return hi, lo


def pack_split_rows(w, nsub):
    """1x1 weights float64 [rows][K] -> float16 [rows][K * nsub] in the order the "mixed" GEMM walks a 64-wide K block
    (seg_gemm.hip, GemmArgs::nsub): nsub 2 = [hi | lo] (the activation block is used twice), nsub 3 = [hi | lo | hi]
    (activation planes hi, hi, lo)."""
    rows, k = w.shape
    assert k % 64 == 0 and nsub in (2, 3)
    hi, lo = split_f16(w)
    parts = [hi.reshape(rows, k // 64, 1, 64), lo.reshape(rows, k // 64, 1, 64)]
    if nsub == 3:
        parts.append(parts[0])
    return torch.cat(parts, dim=2).reshape(rows, k * nsub).contiguous()


def _mfma_a_fragments(m):
    """float16 [rows % 16 == 0][K % 32 == 0] -> [rows/16][K/32][lane 64][8]: the A operand of v_mfma_f32_16x16x32_f16 as one wave
    loads it (lane l holds row l & 15, K values 8 (l >> 4) .. + 7 of the 32-wide step): 1 KB contiguous per fragment."""
    rows, k = m.shape
    return m.reshape(rows // 16, 16, k // 32, 4, 8).permute(0, 2, 3, 1, 4).reshape(rows // 16, k // 32, 64, 8)


def pack_bottleneck(w1, w2, w3, wd, groups):
    """Weights of AVL_OP_BOTTLENECK (seg_bottleneck.hip), every one as an f16 pair hi + lo in MFMA fragment order:
      conv1  float64 [width][cin]            -> [n width/16][ks cin/32][hi, lo][lane][8]
      conv2  float64 [width][width/groups][3][3] -> block-diagonal 16-channel windows, K = 32 = two taps x 16 input channels:
             [window width/16][ks 5][hi, lo][lane][8]; K value 8 kq + j of step ks is tap 2 ks + (kq >> 1) (the tenth is zero),
             input channel (kq & 1) * 8 + j of the window
      conv3  float64 [cout][width] (+ downsample [cout][cin] or None as extra K steps) -> [wave cout/32][ks][nj 2][hi, lo][lane][8];
             MFMA row i of n-tile nj is output channel 32 wave + (i >> 2) * 8 + nj * 4 + (i & 3) (a lane then owns 8 consecutive channels)
    -> three flat float16 tensors."""
    width, cin = w1.shape
    cout = w3.shape[0]
    cg = width // groups
    assert width % 16 == 0 and cin % 32 == 0 and cout % 32 == 0 and 16 % cg == 0

    def pair(m):          # [.., lane, 8] hi and lo fragments interleaved as [..][2][lane][8]
        hi, lo = split_f16(m)
        return torch.stack([_mfma_a_fragments(hi), _mfma_a_fragments(lo)], dim=2)

    p1 = pair(w1.to(torch.float64))                                                   # [n][ks][2][64][8]
    # conv2: dense [window][out 16][tap 10][in 16]
    nwin = width // 16
    dense = torch.zeros((nwin, 16, 10, 16), dtype=torch.float64)
    wt = w2.reshape(width, cg, 9).to(torch.float64)
    co = torch.arange(width)
    win, col = co // 16, co % 16
    gbase = (col // cg) * cg
    for ci in range(cg):
        dense[win, col, :9, gbase + ci] = wt[co, ci, :]
    p2 = pair(dense.reshape(nwin * 16, 160))                                          # K = tap * 16 + in: step ks = taps 2 ks, 2 ks + 1
    # conv3 (+ downsample): rows permuted per 32-channel wave block
    w3k = w3.to(torch.float64) if wd is None else torch.cat([w3.to(torch.float64), wd.to(torch.float64)], dim=1)
    i = torch.arange(16)
    rows = torch.cat([32 * wv + (i >> 2) * 8 + nj * 4 + (i & 3) for wv in range(cout // 32) for nj in range(2)])
    p3 = pair(w3k[rows])                                                              # [wave * 2 + nj][ks][2][64][8]
    ks3 = w3k.shape[1] // 32
    p3 = p3.reshape(cout // 32, 2, ks3, 2, 64, 8).permute(0, 2, 1, 3, 4, 5)            # [wave][ks][nj][2][64][8]
    return p1.reshape(-1).contiguous(), p2.reshape(-1).contiguous(), p3.reshape(-1).contiguous()


_FP4_GRID = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0], dtype=torch.float64)


def fp4_quant_blocks(b):
    """float64 [..., 32] -> (uint8 [..., 16], int64 [...]): one OCP MX-FP4 block per trailing 32 values: scale byte = biased
    exponent of the block maximum - 2 (the maximum lands in [4, 8)), + 1 when it would land above 6.5; e2m1 elements rounded to nearest even,
    element 2i in the LOW nibble of byte i -- the convention of the GPU's v_cvt_scalef32_pk_fp4_f32 (tools/micro/fp4_cvt_probe.hip)."""
    b = b.to(torch.float64)
    amax = b.abs().amax(dim=-1)
    e = (torch.floor(torch.log2(amax.clamp_min(2.0 ** -200))).to(torch.int64) + 127).clamp_min(0)   # biased fp32 exponent of amax (0: zero / subnormal, as the GPU reads it)
    sbyte = torch.where(e >= 3, e - 2, torch.ones_like(e))
    # a block maximum above 6.5 (in units of that scale) takes the next scale instead of saturating at 6 (seg_types.h,
    # mx_fp4_scale_byte: the test is on the float32 mantissa of the maximum, > 1.625)
    mant = amax.to(torch.float32).view(torch.int32).to(torch.int64) & 0x7FFFFF
    sbyte = (sbyte + (mant > 0x500000).to(torch.int64)).clamp(1, 254)
    scale = torch.exp2((sbyte - 127).to(torch.float64)).unsqueeze(-1)
    mag = (b.abs() / scale).clamp_max(6.0)
    mid = (_FP4_GRID[1:] + _FP4_GRID[:-1]) / 2
    code = torch.bucketize(mag, mid, right=False)                  # ties go to the lower code ...
    tie = mag == mid[code.clamp(0, 6)]                              # (bucketize(right=False): mag == mid[i] -> i)
    code = torch.where(tie & (code % 2 == 1), code + 1, code)       # ... unless that one is odd: round half to even mantissa
    code = code.clamp(0, 7) | ((b < 0).to(torch.int64) << 3)
    code = code.reshape(*code.shape[:-1], 16, 2)
    return (code[..., 0] | (code[..., 1] << 4)).to(torch.uint8), sbyte


def mx_quant_fp4(w):
    """float64 [rows][K] (K % 256 == 0) -> (uint8 [rows][K/2], uint8 [K/256][rows][8]): OCP MX-FP4 along K as libavl_hip's
    MX GEMM reads it (include/avl_hip.h, w_split = 2): blocks of 32 (fp4_quant_blocks), scales laid out per 256-wide K block."""
    rows, k = w.shape
    assert k % 256 == 0
    packed, sbyte = fp4_quant_blocks(w.reshape(rows, k // 32, 32))
    scales = sbyte.reshape(rows, k // 256, 8).permute(1, 0, 2).contiguous().to(torch.uint8)
    return packed.reshape(rows, k // 2), scales


def mx_dequant_fp4(packed, scales):
    """inverse of mx_quant_fp4 (also decodes what the GPU kernels write): -> float64 [rows][K]"""
    rows, k2 = packed.shape
    p = packed.to(torch.int64)
    code = torch.stack([p & 15, p >> 4], dim=2).reshape(rows, k2 * 2)
    val = _FP4_GRID[code & 7] * torch.where((code & 8) != 0, -1.0, 1.0)
    sc = scales.permute(1, 0, 2).reshape(rows, -1).to(torch.float64)                          # [rows][K/32]
    return (val.reshape(rows, -1, 32) * torch.exp2(sc - 127).unsqueeze(2)).reshape(rows, k2 * 2)


def mx_bundle_bytes(rows, c):
    """bytes of one half (plane + scales) of an MX bundle of a [rows][c] tensor"""
    return rows * (c // 2) + (c // 256) * rows * 8


def permute_w_scales(scales):
    """WEIGHT scales of an MX GEMM, uint8 [K/256][rows][8] (rows % 16 == 0) -> the same bytes re-ordered inside every 16-row block
    as [row & 3][kq][nj][kk] (row = 16 blk + 4 nj + (row & 3), K block of 32 = 4 kk + kq): the eight scale bytes one lane of
    k_gemm_mx_pipe needs in a sub-step (n-tiles nj = 0..3 x K halves kk = 0, 1 for its row & 3 and its kq) are then 8 contiguous
    bytes -- one LDS read, the byte picked by the MFMA's op_sel (seg_gemm.hip).  Activation scales keep the plain layout."""
    nb, rows, _ = scales.shape
    assert rows % 16 == 0
    s = scales.reshape(nb, rows // 16, 4, 4, 2, 4)             # [mb][blk][nj][c = row & 3][kk][kq]
    return s.permute(0, 1, 3, 5, 2, 4).contiguous().reshape(nb, rows, 8)


def pack_mx_weights(w):
    """float64 [w_rows][K] -> (f16 hi plain rows, uint8 bundle [Q4(W lo) | scales | Q4(W hi) | scales]); the scales in the
    per-lane order of permute_w_scales"""
    hi, lo = split_f16(w)
    ql, sl = mx_quant_fp4(lo.to(torch.float64))
    qh, sh = mx_quant_fp4(hi.to(torch.float64))
    return hi, torch.cat([ql.reshape(-1), permute_w_scales(sl).reshape(-1), qh.reshape(-1), permute_w_scales(sh).reshape(-1)])


# Packed weights are a pure function of (the state dict's content, the op, the packing variant): building the plan for another image size,
# or the five plans of the load-time self-check, repeats the float64 folding and -- far slower -- the host-side FP4 quantisation of the MX
# bundles.  Cached on the host, keyed by a fingerprint of the state dict (not its id(): that can be reused after a free), bounded in bytes.
_PACK_CACHE = {}
_PACK_CACHE_BYTES = [0]
PACK_CACHE_LIMIT = 3 << 30


def state_fingerprint(state):
    """(sum, sum of squares, number of elements) over all tensors, float64: two different weight sets do not share it"""
    s1 = s2 = 0.0
    n = 0
    for k in sorted(state):
        t = state[k]
        if torch.is_tensor(t) and t.is_floating_point():
            t64 = t.detach().to(torch.float64)
            s1 += float(t64.sum())
            s2 += float((t64 * t64).sum()) * (1.0 + 1e-3 * (hash(k) % 97))
            n += t.numel()
    return (s1, s2, n)


def _cached_pack(key, fn):
    hit = _PACK_CACHE.get(key)
    if hit is not None:
        return hit
    out = fn()
    ts = out if isinstance(out, (tuple, list)) else (out,)
    nbytes = sum(t.numel() * t.element_size() for t in ts if torch.is_tensor(t))
    if _PACK_CACHE_BYTES[0] + nbytes > PACK_CACHE_LIMIT:
        _PACK_CACHE.clear()
        _PACK_CACHE_BYTES[0] = 0
    _PACK_CACHE[key] = out
    _PACK_CACHE_BYTES[0] += nbytes
    return out


class Act(object):
    """An activation [rows][ch]: one plane of the activation type, or ("mixed" precision) two float16 planes hi + lo of the
    same shape."""
    __slots__ = ("hi", "lo", "pool_key", "mx", "mx_valid", "lo_fp4")

    def __init__(self, hi, lo=None, pool_key=None, mx=None):
        self.hi, self.lo, self.pool_key = hi, lo, pool_key
        self.mx = mx               # uint8 MX bundle (FP4 copies + scales of hi [and lo]) for the next MX GEMM, or None
        self.mx_valid = False      # set by the op that fills it
        self.lo_fp4 = False        # the lo part exists ONLY as the FP4 half of the bundle (no f16 lo plane)

    @property
    def shape(self):
        return self.hi.shape


class SegNet(object):
    """The compiled network for one input size and precision: device buffers, packed weights and the
    native plan.  ``forward(image_u8_cuda)`` runs it; ``labels`` / ``logits`` are views of its outputs."""

    ROW_PAD = 256        # GEMM tiles read whole 128/256-row tiles
    MIXED_OPTS = ("conv1_split", "conv2_split", "mx", "trunk_fp4", "fuse_ds", "gconv_mx", "dw_exact", "layer1_lo", "fuse_block", "full_split", "fuse_decoder", "fuse_classifier")    # keyword switches of the "mixed" mode

    def __init__(self, state, height, width, precision="bf16", device=None, num_classes=19, output_stride=8, fuse_dwpw=True, raw_frame=None,
                 part=None, **mixed_opts):
        """raw_frame = (src_h, src_w): the plan's input is the RAW BGR camera frame and the node's pre-processing
        (vision_semantic_segmentation_node.py:83-98: BGR->RGB, undistort, INTER_AREA by src_w // width) runs inside the stem's loader
        (16-bit precisions); ``set_camera`` chooses the camera model, ``forward`` takes the raw frame."""
        assert output_stride in (8, 16), "deeplab_v3_plus.py:30-36 / backbone/build.py:11-16 know output strides 8 (the reference configuration, base_cfg.py:106) and 16"
        self.output_stride = int(output_stride)
        assert precision in ("bf16", "f16", "f32", "mixed")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.H, self.W = int(height), int(width)
        self.raw_frame = None if raw_frame is None else (int(raw_frame[0]), int(raw_frame[1]))
        if self.raw_frame is not None:
            f = self.raw_frame[1] // self.W
            if precision == "f32" or f < 1 or (self.raw_frame[0] // f, self.raw_frame[1] // f) != (self.H, self.W):
                raise ValueError("raw_frame %r does not scale to %dx%d by an integer factor (or precision is f32)" % (self.raw_frame, self.H, self.W))
        self.precision = precision
        # "mixed": f16 MFMA with split operands where the error analysis (tools/precision_study.py)
        # says a single f16 rounding is too coarse: every weight is an f16 pair hi + lo, the residual trunk, the ASPP
        # outputs and the whole decoder are stored as two f16 planes, and a GEMM runs 2 or 3 MFMA passes per K block.
        self.mixed = precision == "mixed"
        unknown = sorted(set(mixed_opts) - set(self.MIXED_OPTS))
        if unknown:        # a typo (gconvmx=1) must not silently benchmark the default configuration
            raise TypeError("SegNet: unknown mixed-mode option(s) %s (known: %s)" % (unknown, ", ".join(self.MIXED_OPTS)))
        self.mixed_conv1_split = self.mixed and mixed_opts.get("conv1_split", True)     # conv1 / downsample read trunk hi + lo
        self.mixed_conv2_split = self.mixed and mixed_opts.get("conv2_split", True)     # conv2 writes hi + lo, conv3 reads both
        # correction products on the block-scaled matrix cores (MX-FP4, 4x the f16 rate) wherever shapes allow (K, N % 256)
        self.mixed_mx = self.mixed and mixed_opts.get("mx", True)
        self.mixed_trunk_fp4 = mixed_opts.get("trunk_fp4", True)
        self.mixed_fuse_ds = mixed_opts.get("fuse_ds", True)     # stride-1 downsample folded into conv3 (second input along K)
        self.mixed_fuse_classifier = mixed_opts.get("fuse_classifier", True)   # the classifier + arg-max in the last refine block's epilogue (k_dwpw_xs<CLS>)
        self.mixed_fuse_decoder = mixed_opts.get("fuse_decoder", True)   # the decoder's refine blocks as one k_dwpw_xs launch each (split input)
        self.mixed_dw_exact = mixed_opts.get("dw_exact", True)    # fused depthwise+pointwise (ASPP) with split depthwise weights and a split depthwise result (k_dwpw_x)
        # layer1_lo = True (default since the end of round 5): every block of layer1 keeps the lo plane of its output.  False: the first two
        # blocks write a SINGLE f16 trunk plane (the last one keeps hi + lo: the decoder's low-level branch and layer2 read it) -- round 3's
        # trade when layer1's GEMMs were separate HBM-bound launches (0.4 GB of traffic = 2.5 % of the frame for -10...-30 % logits error);
        # with the blocks fused the planes cost 0.03 ms (0.6 %) and the worst 1080p draw measures 7.3e-4 instead of 8.8e-4
        self.mixed_layer1_lo = mixed_opts.get("layer1_lo", True)
        # fuse_block (default): every Bottleneck of layer1 is ONE kernel (AVL_OP_BOTTLENECK: conv1 -> grouped 3x3 -> conv3 + residual with
        # both intermediates in LDS); False = the three-launch form of rounds 1-4
        self.mixed_fuse_block = self.mixed and mixed_opts.get("fuse_block", True)
        # full_split: the COMPLETE hi + lo pipeline -- every tensor two f16 planes (stem output, max-pool, conv1 outputs and the
        # ASPP depthwise stage included), every product three f16 passes, no FP4 anywhere: the plan a calibrated (trained) checkpoint needs
        # for 1e-3, about 3x the fp32 plan's speed.  It overrides the options it contradicts.
        self.full_split = self.mixed and bool(mixed_opts.get("full_split", False))
        if self.full_split:
            self.mixed_mx = False
            self.mixed_trunk_fp4 = False
            self.mixed_layer1_lo = True
            self.mixed_fuse_block = False        # (the fused layer1 block has no LDS for a t1 lo plane at 256 input channels)
        self.mixed_gconv_mx = mixed_opts.get("gconv_mx", True) and not self.full_split   # grouped conv with FP4 corrections for its weights AND for conv1's output (-10..-30 % logits error, -5 % frames/s)
        self.act_dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "mixed": torch.float16}[precision]
        self.avl_dtype = {"bf16": _lib.AVL_BF16, "f16": _lib.AVL_F16, "f32": _lib.AVL_F32, "mixed": _lib.AVL_F16}[precision]
        self.half = precision != "f32"              # 16-bit activations: MFMA stem / grouped-conv kernels
        self.fuse_dwpw = bool(fuse_dwpw)            # ASPP branches / decoder refine blocks: depthwise + pointwise as one kernel (16-bit types)
        self.num_classes = num_classes
        self._keep = []            # every tensor the plan points at
        self._free = {}            # numel -> [tensor] pool of released activation buffers
        self.ops = []
        self.op_names = []
        self._plan = C.c_void_p()
        self._fp = state_fingerprint(state)
        # part = None: the whole network.  ("aspp", C): height x width is the FEATURE map, `state` an ASPP module's state dict ("aspp." keys);
        # ("decoder", C_feature, C_low): height x width the feature map, the low-level map twice that -- sub-plans for the tests that compare
        # the HIP kernels with the reference MODULES' outputs (tests/golden/net_aspp256.pt, net_decoder256.pt)
        self.part = part
        if part is None:
            self._build(state)
        else:
            self._build_part(state)
        arr = (AvlSegOp * len(self.ops))(*self.ops)
        _lib.check(_lib.lib().avl_seg_plan_create(arr, len(self.ops), C.byref(self._plan)), "avl_seg_plan_create")

    def __del__(self):
        try:
            if self._plan:
                _lib.lib().avl_seg_plan_destroy(self._plan)
        except Exception:
            pass

    # -------------------------------------------------------------------------------- buffers
    def _act(self, rows, ch, split=False, mx=False, lo_fp4=False):
        """activation buffer [rows padded][ch] (split: two planes; mx: plus the MX-FP4 bundle the next MX GEMM reads;
        lo_fp4: one f16 plane, the lo part only as FP4 in the bundle)"""
        prow = _round_up(rows, self.ROW_PAD)
        mx = bool(mx and self.mixed_mx and ch % 256 == 0)
        lo_fp4 = bool(lo_fp4 and mx)
        split = bool(split and not lo_fp4)
        key = (prow, ch, bool(split), mx, lo_fp4)
        if self._free.get(key):
            a = self._free[key].pop()
            a.mx_valid = False
            return a
        if split:
            t = torch.zeros((2, prow, ch), dtype=self.act_dtype, device=self.device)
            a = Act(t[0], t[1], key)
        else:
            t = torch.zeros((prow, ch), dtype=self.act_dtype, device=self.device)
            a = Act(t, None, key)
        self._keep.append(t)
        if mx:
            a.mx = torch.zeros(2 * mx_bundle_bytes(prow, ch), dtype=torch.uint8, device=self.device)
            self._keep.append(a.mx)
        a.lo_fp4 = lo_fp4
        return a

    def _release(self, a):
        self._free.setdefault(a.pool_key, []).append(a)

    def _dev(self, t, dtype):
        t = t.to(dtype).contiguous().to(self.device)
        self._keep.append(t)
        return t

    # -------------------------------------------------------------------------------- op emitters
    def _op(self, name, kind, **f):
        op = AvlSegOp()
        op.kind = kind
        op.dtype = self.avl_dtype
        for k, v in f.items():
            setattr(op, k, v)
        self.ops.append(op)
        self.op_names.append(name)

    @staticmethod
    def _view(t, col=0):
        """(pointer to column `col` of a [rows][ld] buffer, ld, rows); t: Act (its high plane) or a tensor"""
        t = t.hi if isinstance(t, Act) else t
        return t.data_ptr() + col * t.element_size(), t.shape[1], t.shape[0]

    @staticmethod
    def _lo(a, col=0):
        """pointer to column `col` of an Act's low plane (0 = the activation is a single plane)"""
        return 0 if (not isinstance(a, Act) or a.lo is None) else a.lo.data_ptr() + col * a.lo.element_size()

    def _gemm(self, name, src, hw, cin, w, b, dst, dst_col=0, relu=True, res=None, out_f32=False, src_col=0, bias_dev=None,
              read_lo=True, src2=None, w2=None, b2=None, labels=None):
        """1x1 conv.  w float64 [cout][cin] (BN folded), b float64 [cout].  "mixed": weights become f16 pairs; the low
        plane of `src` is read if it has one (unless read_lo = False), `res` and `dst` are used with all the planes they have."""
        h, wd = hw
        cout = w.shape[0]
        w_rows = _round_up(cout, 256)
        wp = torch.zeros((w_rows, cin), dtype=torch.float64)
        wp[:cout] = w.reshape(cout, cin)
        if src2 is not None:          # a second input appended along K (MX GEMM only): conv3 + downsample in one product
            cin2 = w2.shape[1]
            wp2 = torch.zeros((w_rows, cin2), dtype=torch.float64)
            wp2[:cout] = w2.reshape(cout, cin2)
            wp = torch.cat([wp, wp2], dim=1)
            b = b + b2
        in_lo = self._lo(src, src_col) if (self.mixed and read_lo) else 0
        use_mx = (self.mixed_mx and isinstance(src, Act) and src.mx is not None and src.mx_valid and src_col == 0 and not out_f32
                  and cin % 256 == 0 and cout % 256 == 0 and src.hi.shape[1] == cin)
        assert src2 is None or use_mx, "%s: a second input needs the MX GEMM" % name
        w_mx = None
        if use_mx:
            whi, bundle = _cached_pack((self._fp, name, "mx", tuple(wp.shape)), lambda: pack_mx_weights(wp))
            wdev = self._dev(whi, torch.float16)
            w_mx = bundle.to(self.device)
            self._keep.append(w_mx)
        elif self.mixed:
            nsub = 3 if in_lo else 2
            wdev = self._dev(_cached_pack((self._fp, name, "split", nsub, tuple(wp.shape)), lambda: pack_split_rows(wp, nsub)), torch.float16)
        else:
            wdev = self._dev(wp, self.act_dtype)
        if bias_dev is None:
            bp = torch.zeros(w_rows, dtype=torch.float64)
            bp[:cout] = b
            bias_dev = self._dev(bp, torch.float32)
        ip, ild, irows = self._view(src, src_col)
        op_, old, orows = self._view(dst, dst_col)
        f = dict(in_=ip, out=op_, weight=wdev.data_ptr(), bias=bias_dev.data_ptr(), in_h=h, in_w=wd, in_c=cin, in_ld=ild,
                 in_rows=irows, out_h=h, out_w=wd, out_c=cout, out_ld=old, out_rows=orows, relu=int(relu), out_f32=int(out_f32),
                 w_rows=w_rows, ksize=1, stride=1, dil=1, groups=1)
        if res is not None:
            rp, rld, _ = self._view(res)
            f.update(in2=rp, in2_ld=rld)
        if self.mixed:
            f.update(w_split=1, in_lo=in_lo, out_lo=self._lo(dst, dst_col), in2_lo=self._lo(res) if res is not None else 0)
        flags = 0
        if use_mx:
            f.update(w_split=2, w_mx=w_mx.data_ptr(), in_mx=src.mx.data_ptr())
            if src.lo_fp4 and read_lo:
                flags |= AVL_MX_IN_LO
            if src2 is not None:
                assert src2.mx_valid and src2.lo_fp4 == src.lo_fp4 and src2.lo is None and src.lo is None
                f.update(in3=src2.hi.data_ptr(), in3_mx=src2.mx.data_ptr(), in3_c=src2.hi.shape[1], in3_ld=src2.hi.shape[1])
            if isinstance(res, Act) and res.lo_fp4:
                f.update(in2_mx=res.mx.data_ptr())
                flags |= AVL_MX_RES_LO
            if isinstance(dst, Act) and dst.mx is not None and dst_col == 0 and dst.hi.shape[1] == cout:
                f.update(out_mx=dst.mx.data_ptr())
                dst.mx_valid = True
                if dst.lo_fp4:
                    flags |= AVL_MX_OUT_LO
        else:
            for t_, what in ((src if read_lo else None, "input"), (res, "residual"), (dst, "output")):
                if isinstance(t_, Act) and t_.lo_fp4:
                    raise RuntimeError("%s: the %s keeps its lo part as FP4 only, which needs the MX GEMM" % (name, what))
        f["mx_flags"] = flags
        if labels is not None:
            assert out_f32 and not use_mx and res is None and not relu
            f["out_mx"] = labels.data_ptr()
        self._op(name, OP_GEMM, **f)

    def _bottleneck(self, p, st, x, hw, cin, width, cout, y):
        """One Bottleneck (stride 1, dilation 1) as AVL_OP_BOTTLENECK: BN folded, weights as f16 pairs in fragment order."""
        w1, b1 = fold_bn(st, p + ".conv1.weight", p + ".bn1")
        w2, b2 = fold_bn(st, p + ".conv2.weight", p + ".bn2")
        w3, b3 = fold_bn(st, p + ".conv3.weight", p + ".bn3")
        wd = None
        if (p + ".downsample.0.weight") in st:
            wd, bd = fold_bn(st, p + ".downsample.0.weight", p + ".downsample.1")
            wd, b3 = wd.reshape(cout, cin), b3 + bd
        p1, p2, p3 = (self._dev(t, torch.float16) for t in _cached_pack(
            (self._fp, p, "bottleneck"), lambda: pack_bottleneck(w1.reshape(width, cin), w2, w3.reshape(cout, width), wd, GROUPS)))
        bias = self._dev(torch.cat([b1, b2, b3]), torch.float32)
        ip, ild, irows = self._view(x)
        op_, old, orows = self._view(y)
        self._op(p, OP_BOTTLENECK, in_=ip, in_lo=self._lo(x), out=op_, out_lo=self._lo(y), weight=p1.data_ptr(), in2=p2.data_ptr(), in3=p3.data_ptr(),
                 in3_c=width, bias=bias.data_ptr(), in_h=hw[0], in_w=hw[1], in_c=cin, in_ld=ild, in_rows=irows, out_h=hw[0], out_w=hw[1],
                 out_c=cout, out_ld=old, out_rows=orows, ksize=3, stride=1, pad=1, dil=1, groups=GROUPS, relu=1, w_layout=int(wd is not None),
                 w_split=int(wd is not None))

    def _dwpw(self, name, src, hw, cin, w_dw, b_dw, w_pw, b_pw, dst, dst_col, dilation, padding=None, classifier=None):
        """DepthwiseSeparableConv2d (3x3 depthwise dil d pad p + BN + ReLU, 1x1 + BN + ReLU) as one op.  classifier = (w [K][cout] float64, b [K],
        logits fp32 [rows][K], labels uint8 [rows]): the network's last 1x1 conv and the arg-max run in the kernel's epilogue (split input, cout = 256,
        K <= 32); dst is then not written (None)."""
        h, wd = hw
        padding = dilation if padding is None else padding
        oh, ow = h + 2 * padding - 2 * dilation, wd + 2 * padding - 2 * dilation
        cout = w_pw.shape[0]
        w_rows = _round_up(cout, 256)
        wp = torch.zeros((w_rows, cin), dtype=torch.float64)
        wp[:cout] = w_pw.reshape(cout, cin)
        bp = torch.zeros(w_rows, dtype=torch.float64)
        bp[:cout] = b_pw
        wdev = self._dev(pack_split_rows(wp, 2), torch.float16) if self.mixed else self._dev(wp, self.act_dtype)
        bdev = self._dev(bp, torch.float32)
        exact = self.mixed and self.mixed_dw_exact
        # exact: fp32 depthwise weights (w_split 3), the depthwise result as a split tile; a split input (hi + lo planes) exists in this form
        # only (k_dwpw_xs), its stride-1 case (the decoder) walks 8 x 16-pixel blocks
        in_lo = self._lo(src) if exact else 0
        dwp = pack_dw_f32(w_dw, b_dw) if exact else pack_dw_pairs(w_dw, b_dw, self.act_dtype)
        blocks = bool(in_lo) and dilation == 1
        params = torch.cat([dwp, dwpw_block_order(oh, ow) if blocks else dwpw_tile_order(oh, ow, dilation)]).to(self.device)
        self._keep.append(params)
        ip, ild, irows = self._view(src)
        if classifier is not None:
            wc, bc, logits, labels = classifier
            ncls = wc.shape[0]
            assert in_lo and cout == 256 and ncls <= 32 and wc.shape[1] == cout
            wc32 = torch.zeros((32, cout), dtype=torch.float64)
            wc32[:ncls] = wc
            bc32 = torch.zeros(32, dtype=torch.float64)
            bc32[:ncls] = bc
            wcd = self._dev(torch.stack(split_f16(wc32)), torch.float16)           # [hi | lo][32][cout]
            bcd = self._dev(bc32, torch.float32)
            self._op(name, OP_DWPW, in_=ip, in_lo=in_lo, in2=params.data_ptr(), in2_lo=bcd.data_ptr(), in3=wcd.data_ptr(), in3_c=ncls, out=logits.data_ptr(),
                     out_mx=labels.data_ptr(), out_f32=1, weight=wdev.data_ptr(), bias=bdev.data_ptr(), in_h=h, in_w=wd, in_c=cin, in_ld=ild, in_rows=irows,
                     out_h=oh, out_w=ow, out_c=cout, out_ld=ncls, out_rows=logits.shape[0], relu=1, w_rows=w_rows, ksize=3, stride=1, pad=padding,
                     dil=dilation, groups=cin, w_split=3, w_layout=int(blocks))
            return
        op_, old, orows = self._view(dst, dst_col)
        self._op(name, OP_DWPW, in_=ip, in_lo=in_lo, in2=params.data_ptr(), out=op_, weight=wdev.data_ptr(), bias=bdev.data_ptr(), in_h=h, in_w=wd,
                 in_c=cin, in_ld=ild, in_rows=irows, out_h=oh, out_w=ow, out_c=cout, out_ld=old, out_rows=orows, relu=1, w_rows=w_rows,
                 ksize=3, stride=1, pad=padding, dil=dilation, groups=cin, w_split=(3 if exact else int(self.mixed)), w_layout=int(blocks),
                 out_lo=self._lo(dst, dst_col) if self.mixed else 0)

    def _spatial(self, name, kind, src, in_hw, cin, dst, out_hw, cout, weight=None, bias=None, dst_col=0, **extra):
        ip, ild, irows = self._view(src)
        op_, old, orows = self._view(dst, dst_col)
        f = dict(in_=ip, out=op_, in_h=in_hw[0], in_w=in_hw[1], in_c=cin, in_ld=ild, in_rows=irows, out_h=out_hw[0],
                 out_w=out_hw[1], out_c=cout, out_ld=old, out_rows=orows)
        if weight is not None:
            f["weight"] = weight.data_ptr()
        if bias is not None:
            f["bias"] = bias.data_ptr()
        if self.mixed and kind in (OP_BILINEAR, OP_DWCONV, OP_GCONV):
            f["out_lo"] = self._lo(dst, dst_col)
            if kind != OP_GCONV:
                f["in_lo"] = self._lo(src)
            if (isinstance(dst, Act) and dst.mx is not None and dst_col == 0
                    and ((kind == OP_GCONV and extra.get("w_layout") == 1) or (kind == OP_DWCONV and f.get("in_lo")))):
                f["out_mx"] = dst.mx.data_ptr()
                dst.mx_valid = True
                if dst.lo_fp4:
                    f["mx_flags"] = AVL_MX_OUT_LO
        f["mx_flags"] = f.get("mx_flags", 0) | extra.pop("mx_flags", 0)
        f.update(extra)
        self._op(name, kind, **f)

    # -------------------------------------------------------------------------------- the network
    def _build(self, st):
        H, W = self.H, self.W
        dev = self.device
        # the plan's input: the RGB network input, or the raw BGR camera frame when the stem pre-processes
        self.image = torch.zeros((H, W, 3) if self.raw_frame is None else self.raw_frame + (3,), dtype=torch.uint8, device=dev)
        self.zero_page = torch.zeros(64, dtype=torch.uint8, device=dev)          # what a depthwise tap outside the image reads
        self.camera_block = torch.zeros(64, dtype=torch.uint8, device=dev)       # AVL_STEM_CAMERA_BYTES: zeros = no undistortion
        self._keep += [self.image, self.zero_page, self.camera_block]

        # ---- stem: conv1 7x7 s2 + bn1 + relu (resnet.py:25-27), maxpool (:28)
        h2, w2 = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        w, b = fold_bn(st, "backbone.conv1.weight", "backbone.bn1")
        if self.full_split:         # f16 pairs: the hi parts' fragments, then the lo parts'
            whi, wlo = split_f16(w)
            w_stem, stem_layout = self._dev(torch.cat([pack_stem_mfma(whi.to(torch.float64)), pack_stem_mfma(wlo.to(torch.float64))]), torch.float16), 1
        elif self.half:
            w_stem, stem_layout = self._dev(pack_stem_mfma(w), self.act_dtype), 1
        else:
            w_stem, stem_layout = self._dev(w.permute(2, 3, 1, 0).reshape(-1), torch.float32), 0   # [ky][kx][ci][co]
        b_stem = self._dev(b, torch.float32)
        stem = self._act(h2 * w2, 64, split=self.full_split)
        raw = {} if self.raw_frame is None else dict(in2=self.camera_block.data_ptr(), in2_ld=self.raw_frame[1])
        self._op("backbone.conv1", OP_STEM, in_=self.image.data_ptr(), out=stem.hi.data_ptr(), weight=w_stem.data_ptr(),
                 bias=b_stem.data_ptr(), in_h=H, in_w=W, in_c=3, in_ld=3, in_rows=self.image.shape[0] * self.image.shape[1], out_h=h2, out_w=w2,
                 out_c=64, out_ld=64, out_rows=stem.shape[0], ksize=7, stride=2, pad=3, dil=1, groups=1, relu=1, w_layout=stem_layout,
                 w_split=int(self.full_split), out_lo=self._lo(stem), **raw)
        h4, w4 = (h2 + 2 - 3) // 2 + 1, (w2 + 2 - 3) // 2 + 1
        x = self._act(h4 * w4, 64, split=self.full_split)
        self._spatial("backbone.maxpool", OP_MAXPOOL, stem, (h2, w2), 64, x, (h4, w4), 64, ksize=3, stride=2, pad=1, dil=1,
                      **(dict(in_lo=self._lo(stem), out_lo=self._lo(x)) if self.full_split else {}))
        self._release(stem)

        # ---- layer1..4 (torchvision _make_layer, replace_stride_with_dilation = (False, True, True))
        hw, cin = (h4, w4), 64
        dilation = 1
        low = None
        # replace_stride_with_dilation (backbone/build.py:11-16): OS8 (False, True, True), OS16 (False, False, True)
        dilated = (False, False, True, True) if self.output_stride == 8 else (False, False, False, True)
        for li, (planes, nblocks, stride0, dilate) in enumerate(zip(PLANES, LAYERS, (1, 2, 2, 2), dilated), start=1):
            width = int(planes * (WIDTH_PER_GROUP / 64.0)) * GROUPS
            cout = planes * EXPANSION
            previous_dilation = dilation
            stride = stride0
            if dilate:
                dilation *= stride
                stride = 1
            for bi in range(nblocks):
                p = "backbone.layer%d.%d" % (li, bi)
                trunk_lo = self.mixed_layer1_lo or li != 1 or bi == nblocks - 1      # does this block's output keep its lo plane
                s = stride if bi == 0 else 1
                d = previous_dilation if bi == 0 else dilation
                ohw = ((hw[0] + 2 * d - 2 * d - 1) // s + 1, (hw[1] + 2 * d - 2 * d - 1) // s + 1)
                if (self.mixed_fuse_block and s == 1 and d == 1 and width == 128 and cout == 256 and cin in (64, 256)
                        and ((p + ".downsample.0.weight") in st) == (cin == 64) and x.hi.shape[1] == cin and (cin == 256 or x.lo is None)):
                    y = self._act(ohw[0] * ohw[1], cout, split=trunk_lo)
                    self._bottleneck(p, st, x, hw, cin, width, cout, y)
                    if x is not low:
                        self._release(x)
                    x, hw, cin = y, ohw, cout
                    continue
                # conv1 1x1 + bn1 + relu
                w, b = fold_bn(st, p + ".conv1.weight", p + ".bn1")
                # where conv1 runs as an MX GEMM its output keeps FP4 copies of both parts (the lo part only so): the grouped conv
                # then corrects for the rounding of conv1's output too -- the largest single error term otherwise
                conv1_mx = (self.mixed_mx and self.mixed_gconv_mx and x.mx is not None and x.mx_valid and x.hi.shape[1] == cin
                            and cin % 256 == 0 and width % 256 == 0 and 32 % (width // GROUPS) == 0)
                t1 = self._act(hw[0] * hw[1], width, split=self.full_split, mx=conv1_mx, lo_fp4=conv1_mx)
                self._gemm(p + ".conv1", x, hw, cin, w, b, t1, read_lo=self.mixed_conv1_split)
                # conv2 3x3 grouped + bn2 + relu
                w, b = fold_bn(st, p + ".conv2.weight", p + ".bn2")
                cg = width // GROUPS
                gconv_mx = self.mixed and t1.mx is not None and t1.mx_valid and t1.lo_fp4
                wsplit = int(self.mixed)
                extra_g = {}
                if gconv_mx:
                    frag_hi, bundle = _cached_pack((self._fp, p + ".conv2", "gconv_mx"), lambda: pack_gconv_mx(w, GROUPS))
                    wg_d, layout, wsplit = self._dev(frag_hi.reshape(-1), torch.float16), 1, 2
                    wb = bundle.to(self.device)
                    self._keep.append(wb)
                    extra_g = dict(w_mx=wb.data_ptr(), in_mx=t1.mx.data_ptr(), mx_flags=AVL_MX_IN_LO)
                elif self.half and width % 64 == 0 and 32 % cg == 0:
                    if self.mixed:       # f16 pairs: 18 "taps" = 9 hi + 9 lo per window and n-tile
                        nwin = width // 32
                        whi, wlo = split_f16(w)
                        packed = torch.cat([pack_gconv_windows(whi.to(torch.float64), GROUPS).reshape(nwin, 2, 9, 16, 32),
                                            pack_gconv_windows(wlo.to(torch.float64), GROUPS).reshape(nwin, 2, 9, 16, 32)], dim=2)
                        wg_d, layout = self._dev(packed.reshape(-1), torch.float16), 1
                    else:
                        wg_d, layout = self._dev(pack_gconv_windows(w, GROUPS), self.act_dtype), 1
                else:
                    wg = w.reshape(GROUPS, cg, cg, 3, 3).permute(0, 3, 4, 2, 1).reshape(-1)   # [g][ky][kx][ci][co]
                    wg_d, layout = self._dev(wg, torch.float32), 0
                bg_d = self._dev(b, torch.float32)
                # (conv3 as an MX GEMM reads the 3x3 output's lo part only through its FP4 copy: no f16 lo plane then)
                t2_fp4 = self.mixed_conv2_split and self.mixed_trunk_fp4 and width % 256 == 0 and cout % 256 == 0 and layout == 1
                t2 = self._act(ohw[0] * ohw[1], width, split=self.mixed_conv2_split, mx=True, lo_fp4=t2_fp4)
                if self.full_split:
                    assert layout == 1 and wsplit == 1 and t1.lo is not None, "full_split needs the MFMA grouped conv with split weights"
                    extra_g = dict(in_lo=self._lo(t1))
                if self.full_split and s != 1:
                    # two tile buffers of a stride-2 tile do not fit the LDS with a lo plane beside the hi plane: the one strided 3x3 of the
                    # network (layer2.0) runs at stride 1 and its result is sub-sampled (pad 1: out(y, x) = out1(2 y, 2 x))
                    t2f = self._act(hw[0] * hw[1], width, split=True)
                    self._spatial(p + ".conv2", OP_GCONV, t1, hw, width, t2f, hw, width, wg_d, bg_d, ksize=3, stride=1, pad=d, dil=d,
                                  groups=GROUPS, relu=1, w_layout=layout, w_split=wsplit, **extra_g)
                    self._spatial(p + ".conv2.sub", OP_SUBSAMPLE, t2f, hw, width, t2, ohw, width, stride=s)
                    self._spatial(p + ".conv2.sub[lo]", OP_SUBSAMPLE, t2f.lo, hw, width, t2.lo, ohw, width, stride=s)
                    self._release(t2f)
                else:
                    self._spatial(p + ".conv2", OP_GCONV, t1, hw, width, t2, ohw, width, wg_d, bg_d, ksize=3, stride=s, pad=d, dil=d,
                                  groups=GROUPS, relu=1, w_layout=layout, w_split=wsplit, **extra_g)
                self._release(t1)
                # identity / downsample.  Stride-1 downsamples (layer3.0, layer4.0) whose input and the 3x3 output both carry MX
                # bundles are folded into conv3 as a second input along K: the identity tensor never exists
                fuse_ds = ((p + ".downsample.0.weight") in st and s == 1 and self.mixed_mx and self.mixed_fuse_ds and x.mx is not None
                           and x.mx_valid and x.hi.shape[1] == cin and cin % 256 == 0 and t2.mx is not None and t2.mx_valid
                           and width % 256 == 0 and cout % 256 == 0 and t2.lo_fp4 and x.lo_fp4)
                if fuse_ds:
                    idn = None
                elif (p + ".downsample.0.weight") in st:
                    w, b = fold_bn(st, p + ".downsample.0.weight", p + ".downsample.1")
                    src = x
                    if s != 1:
                        keep_lo = self.mixed_conv1_split and x.lo is not None
                        sub = self._act(ohw[0] * ohw[1], cin, split=keep_lo)
                        self._spatial(p + ".downsample.sub", OP_SUBSAMPLE, x, hw, cin, sub, ohw, cin, stride=s)
                        if keep_lo:
                            self._spatial(p + ".downsample.sub[lo]", OP_SUBSAMPLE, x.lo, hw, cin, sub.lo, ohw, cin, stride=s)
                        src = sub
                    idn = self._act(ohw[0] * ohw[1], cout, split=self.mixed and trunk_lo)
                    self._gemm(p + ".downsample", src, ohw, cin, w, b, idn, relu=False, read_lo=self.mixed_conv1_split)
                    if s != 1:
                        self._release(sub)
                else:
                    idn = x
                # conv3 1x1 + bn3 + residual + relu
                w, b = fold_bn(st, p + ".conv3.weight", p + ".bn3")
                # where conv3 runs as an MX GEMM the trunk keeps its lo part only as FP4 (3 instead of 5 bytes per element move
                # through conv3's epilogue, which is what bounds it: the 10 % error of FP4 hits a term that is 2^-11 of the sum)
                trunk_fp4 = (self.mixed_mx and self.mixed_trunk_fp4 and t2.mx is not None and t2.mx_valid and width % 256 == 0
                             and cout % 256 == 0 and t2.hi.shape[1] == width)
                y = self._act(ohw[0] * ohw[1], cout, split=self.mixed and trunk_lo, mx=True, lo_fp4=trunk_fp4)
                if fuse_ds:
                    wd_, bd_ = fold_bn(st, p + ".downsample.0.weight", p + ".downsample.1")
                    self._gemm(p + ".conv3+downsample", t2, ohw, width, w, b, y, relu=True, src2=x, w2=wd_.reshape(cout, cin), b2=bd_)
                else:
                    self._gemm(p + ".conv3", t2, ohw, width, w, b, y, relu=True, res=idn)
                self._release(t2)
                if idn is not x and idn is not None:
                    self._release(idn)
                if x is not low:              # layer1's output stays alive for the decoder
                    self._release(x)
                x, hw, cin = y, ohw, cout
            if li == 1:
                low, low_hw, low_c = x, hw, cin           # low_features = layer1 output (resnet.py:33-34)

        aspp, aspp_out = self._emit_aspp(st, x, hw, cin)
        self._emit_decoder(st, aspp, hw, aspp_out, low, low_hw, low_c)

    def _part_input(self, rows, ch):
        """a sub-plan's input activation (hi [+ lo] planes of the activation type) -> (Act, setter(float tensor [ch, h, w]))"""
        a = self._act(rows, ch, split=self.mixed)

        def put(x):
            m = x.permute(1, 2, 0).reshape(rows, ch).to(torch.float64)
            hi = m.to(self.act_dtype)
            a.hi[:rows].copy_(hi.to(self.device))
            if a.lo is not None:
                a.lo[:rows].copy_((m - hi.to(torch.float64)).to(self.act_dtype).to(self.device))
        return a, put

    def _build_part(self, st):
        H, W = self.H, self.W
        self.zero_page = torch.zeros(64, dtype=torch.uint8, device=self.device)
        self._keep.append(self.zero_page)
        if self.part[0] == "aspp":
            feat, self.set_feature = self._part_input(H * W, self.part[1])
            self.part_out, self.part_out_c = self._emit_aspp(st, feat, (H, W), self.part[1])
            self.out_h, self.out_w = H, W
        else:
            feat, self.set_feature = self._part_input(H * W, self.part[1])
            low, self.set_low = self._part_input(4 * H * W, self.part[2])
            self._emit_decoder(st, feat, (H, W), self.part[1], low, (2 * H, 2 * W), self.part[2])

    def part_output(self):
        """float32 [C, h, w] of a ("aspp", ...) sub-plan's output (hi + lo planes)"""
        a, rows = self.part_out, self.out_h * self.out_w
        y = a.hi[:rows, :self.part_out_c].float()
        if a.lo is not None:
            y = y + a.lo[:rows, :self.part_out_c].float()
        return y.reshape(self.out_h, self.out_w, self.part_out_c).permute(2, 0, 1)

    def _emit_aspp(self, st, feat, fhw, fc):
        """ASPP (aspp.py:79-95), dilations forced to 1,12,24,36 for OS8 (deeplab_v3_plus.py:33-34) -> (output Act, its channels)"""
        dev = self.device
        M = fhw[0] * fhw[1]
        dil = (1, 12, 24, 36) if getattr(self, "output_stride", 8) == 8 else (1, 6, 12, 18)
        branches = []
        i = 0
        while ("aspp.module_pyramid.%d.conv.weight" % i) in st or ("aspp.module_pyramid.%d.depthwise_cnn.conv.weight" % i) in st:
            branches.append(i)
            i += 1
        bch = [st["aspp.module_pyramid.0.conv.weight"].shape[0]] + [st["aspp.module_pyramid.%d.pointwise_cnn.conv.weight" % k].shape[0]
                                                                    for k in branches[1:]]
        ncat = sum(bch)
        cat = self._act(M, ncat, split=self.mixed)
        w, b = fold_bn(st, "aspp.module_pyramid.0.conv.weight", "aspp.module_pyramid.0.bn")
        self._gemm("aspp.module_pyramid.0", feat, fhw, fc, w, b, cat, dst_col=0)
        col = bch[0]
        for k in branches[1:]:
            p = "aspp.module_pyramid.%d" % k
            w, b = fold_bn(st, p + ".depthwise_cnn.conv.weight", p + ".depthwise_cnn.bn")
            w2, b2 = fold_bn(st, p + ".pointwise_cnn.conv.weight", p + ".pointwise_cnn.bn")
            if self.half and self.fuse_dwpw and fc % 64 == 0 and fc <= 2048:
                # depthwise + pointwise in one kernel: the 132 MB intermediate never goes to HBM (AVL_OP_DWPW)
                self._dwpw(p, feat, fhw, fc, w, b, w2, b2, cat, col, dil[k])
            else:
                wd_, bd_ = self._dev(w.reshape(fc, 9).t().reshape(-1), torch.float32), self._dev(b, torch.float32)    # [tap][C]
                t = self._act(M, fc, split=self.mixed)
                self._spatial(p + ".depthwise_cnn", OP_DWCONV, feat, fhw, fc, t, fhw, fc, wd_, bd_, ksize=3, stride=1, pad=dil[k],
                              dil=dil[k], groups=fc, relu=1, in2=self.zero_page.data_ptr())
                self._gemm(p + ".pointwise_cnn", t, fhw, fc, w2, b2, cat, dst_col=col)
                self._release(t)
            col += bch[k]
        # image pooling branch -> per-frame bias of the projection
        wg_, bg_ = fold_bn(st, "aspp.global_avg_pool.1.conv.weight", "aspp.global_avg_pool.1.bn")
        wp_, bp_ = fold_bn(st, "aspp.conv.conv.weight", "aspp.conv.bn")
        npool = wg_.shape[0]
        aspp_out = wp_.shape[0]
        wp_ = wp_.reshape(aspp_out, ncat + npool)
        gap_partial = torch.zeros((256, fc), dtype=torch.float32, device=dev)
        gap_vec = torch.zeros(fc, dtype=torch.float32, device=dev)
        pool_vec = torch.zeros(npool, dtype=torch.float32, device=dev)
        proj_bias = torch.zeros(_round_up(aspp_out, 256), dtype=torch.float32, device=dev)
        self._keep += [gap_partial, gap_vec, pool_vec, proj_bias]
        fp, fld, frows = self._view(feat)
        self._op("aspp.global_avg_pool.0", OP_GAP, in_=fp, in2=gap_partial.data_ptr(), out=gap_vec.data_ptr(), in_h=fhw[0], in_w=fhw[1],
                 in_c=fc, in_ld=fld, in_rows=frows, out_h=1, out_w=1, out_c=fc, out_ld=fc, out_rows=1)
        wgd, bgd = self._dev(wg_.reshape(npool, fc), torch.float32), self._dev(bg_, torch.float32)
        self._op("aspp.global_avg_pool.1", OP_GEMV, dtype=_lib.AVL_F32, in_=gap_vec.data_ptr(), out=pool_vec.data_ptr(), weight=wgd.data_ptr(),
                 bias=bgd.data_ptr(), in_h=1, in_w=1, in_c=fc, in_ld=fc, in_rows=1, out_h=1, out_w=1, out_c=npool, out_ld=npool,
                 out_rows=1, relu=1)
        wpd, bpd = self._dev(wp_[:, ncat:], torch.float32), self._dev(bp_, torch.float32)
        self._op("aspp.conv[pool slice]", OP_GEMV, dtype=_lib.AVL_F32, in_=pool_vec.data_ptr(), out=proj_bias.data_ptr(), weight=wpd.data_ptr(),
                 bias=bpd.data_ptr(), in_h=1, in_w=1, in_c=npool, in_ld=npool, in_rows=1, out_h=1, out_w=1, out_c=aspp_out,
                 out_ld=aspp_out, out_rows=1, relu=0)
        aspp = self._act(M, aspp_out, split=self.mixed)
        self._gemm("aspp.conv", cat, fhw, ncat, wp_[:, :ncat], None, aspp, bias_dev=proj_bias)       # dropout = identity (eval)
        self._release(cat)
        self._release(feat)
        return aspp, aspp_out

    def _emit_decoder(self, st, aspp, fhw, aspp_out, low, low_hw, low_c):
        """decoder (decoder.py:45-51) -> logits_buf / labels_buf"""
        dev = self.device
        w, b = fold_bn(st, "decoder.low_level_conv.conv.weight", "decoder.low_level_conv.bn")
        # MODEL.DECODER.LOW_LEVEL_OUT_CHANNELS other than the reference's 256 (48 in the DeepLabV3+ paper): the low-level branch is padded with
        # zero channels to the kernels' granule (zero rows of this conv, zero depthwise taps and zero pointwise columns in the first refine
        # block): ReLU(0) = 0 contributes exactly nothing
        low_true = w.shape[0]
        low_out = _round_up(low_true, 128 if self.mixed else 64)
        if low_out != low_true:
            w = torch.cat([w, torch.zeros((low_out - low_true,) + tuple(w.shape[1:]), dtype=w.dtype)])
            b = torch.cat([b, torch.zeros(low_out - low_true, dtype=b.dtype)])
        Ml = low_hw[0] * low_hw[1]
        cat2 = self._act(Ml, aspp_out + low_out, split=self.mixed)
        self._gemm("decoder.low_level_conv", low, low_hw, low_c, w, b, cat2, dst_col=aspp_out)
        self._spatial("decoder.interpolate", OP_BILINEAR, aspp, fhw, aspp_out, cat2, low_hw, aspp_out)
        self._release(aspp)
        self._release(low)
        x, hw, cin = cat2, low_hw, aspp_out + low_out
        k = 0
        while ("decoder.refine_layers.%d.depthwise_cnn.conv.weight" % k) in st:
            p = "decoder.refine_layers.%d" % k
            ohw = (hw[0] - 2, hw[1] - 2)                                # padding 0 (decoder.py:33-36 default)
            w, b = fold_bn(st, p + ".depthwise_cnn.conv.weight", p + ".depthwise_cnn.bn")
            w2, b2 = fold_bn(st, p + ".pointwise_cnn.conv.weight", p + ".pointwise_cnn.bn")
            if k == 0 and w.shape[0] != cin:             # the padded low-level channels (see above)
                npad = cin - w.shape[0]
                w = torch.cat([w, torch.zeros((npad,) + tuple(w.shape[1:]), dtype=w.dtype)])
                b = torch.cat([b, torch.zeros(npad, dtype=b.dtype)])
                w2 = torch.cat([w2, torch.zeros((w2.shape[0], npad) + tuple(w2.shape[2:]), dtype=w2.dtype)], dim=1)
            last = ("decoder.refine_layers.%d.depthwise_cnn.conv.weight" % (k + 1)) not in st
            fused_mixed = self.mixed and self.mixed_fuse_decoder and self.mixed_dw_exact
            # the last refine block also carries the classifier (decoder.py:42-43) and the arg-max in its epilogue: its 256-channel result never goes to memory
            if (last and fused_mixed and self.mixed_fuse_classifier and self.half and self.fuse_dwpw and cin % 64 == 0 and w2.shape[0] == 256
                    and self.num_classes <= 32 and x.lo is not None):
                pc = "decoder.refine_layers.%d" % (k + 1)
                wc, bc = fold_bn(st, pc + ".conv.weight", None)
                self.out_h, self.out_w = ohw
                Mo = ohw[0] * ohw[1]
                self.logits_buf = torch.zeros((_round_up(Mo, self.ROW_PAD), self.num_classes), dtype=torch.float32, device=dev)
                self.labels_buf = torch.zeros(_round_up(Mo, self.ROW_PAD), dtype=torch.uint8, device=dev)
                self._keep += [self.logits_buf, self.labels_buf]
                self._dwpw(p + "+classifier", x, hw, cin, w, b, w2, b2, None, 0, 1, padding=0,
                           classifier=(wc.reshape(self.num_classes, w2.shape[0]), bc, self.logits_buf, self.labels_buf))
                self._release(x)
                return
            y = self._act(ohw[0] * ohw[1], w2.shape[0], split=self.mixed)
            # "mixed": the decoder keeps every activation as hi + lo (the logits are most sensitive to roundings here:
            # tools/precision_study.py).  fuse_decoder (default): one k_dwpw_xs launch per block -- split input, depthwise weights as
            # f16 pairs, the depthwise result as a split tile in LDS, three MFMA passes, split output; off: the split depthwise
            # kernel (fp32 weights) -> HBM -> an MX / three-pass GEMM
            if self.half and self.fuse_dwpw and cin % 64 == 0 and cin <= 2048 and (not self.mixed or (self.mixed_fuse_decoder and self.mixed_dw_exact)):
                self._dwpw(p, x, hw, cin, w, b, w2, b2, y, 0, 1, padding=0)
                self._release(x)
            else:
                wd_, bd_ = self._dev(w.reshape(cin, 9).t().reshape(-1), torch.float32), self._dev(b, torch.float32)
                # mixed: the depthwise output feeds an MX GEMM where shapes allow (f16 plane + FP4 copies, lo part as FP4 only)
                t_fp4 = self.mixed_mx and self.mixed_trunk_fp4 and cin % 256 == 0 and w2.shape[0] % 256 == 0 and x.lo is not None
                t = self._act(ohw[0] * ohw[1], cin, split=self.mixed, mx=t_fp4, lo_fp4=t_fp4)
                self._spatial(p + ".depthwise_cnn", OP_DWCONV, x, hw, cin, t, ohw, cin, wd_, bd_, ksize=3, stride=1, pad=0, dil=1, groups=cin, relu=1,
                              in2=self.zero_page.data_ptr())
                self._release(x)
                self._gemm(p + ".pointwise_cnn", t, ohw, cin, w2, b2, y)
                self._release(t)
            w = w2
            x, hw, cin = y, ohw, w.shape[0]
            k += 1
        p = "decoder.refine_layers.%d" % k
        w, b = fold_bn(st, p + ".conv.weight", None)
        self.out_h, self.out_w = hw
        Mo = hw[0] * hw[1]
        self.logits_buf = torch.zeros((_round_up(Mo, self.ROW_PAD), self.num_classes), dtype=torch.float32, device=dev)
        self.labels_buf = torch.zeros(_round_up(Mo, self.ROW_PAD), dtype=torch.uint8, device=dev)
        self._keep += [self.logits_buf, self.labels_buf]
        # the arg-max (semantic_segmentation.py:56) rides in the classifier's epilogue: the 19 logits of a pixel sit in two lanes' registers there
        self._gemm(p, x, hw, cin, w, b, Act(self.logits_buf), relu=False, out_f32=True, labels=self.labels_buf if self.num_classes <= 32 else None)
        if self.num_classes > 32:
            self._op("argmax", OP_ARGMAX, dtype=_lib.AVL_F32, in_=self.logits_buf.data_ptr(), out=self.labels_buf.data_ptr(), in_h=hw[0], in_w=hw[1],
                     in_c=self.num_classes, in_ld=self.num_classes, in_rows=self.logits_buf.shape[0], out_h=hw[0], out_w=hw[1], out_c=1,
                     out_ld=1, out_rows=self.labels_buf.shape[0])

    # -------------------------------------------------------------------------------- running
    @property
    def labels(self):
        """uint8 CUDA tensor [out_h, out_w] of the last forward (argmax over classes)."""
        return self.labels_buf[:self.out_h * self.out_w].view(self.out_h, self.out_w)

    @property
    def logits(self):
        """float32 CUDA tensor [out_h, out_w, K] of the last forward (NHWC)."""
        return self.logits_buf[:self.out_h * self.out_w].view(self.out_h, self.out_w, self.num_classes)

    def set_camera(self, K=None, dist=None, stream=None):
        """raw_frame plans: the camera model the stem undistorts with (3x3 K, k1 k2 p1 p2 k3); None = no undistortion.
        Stream-ordered, so it may change between two forwards of a captured plan."""
        assert self.raw_frame is not None, "set_camera needs a plan built with raw_frame"
        assert (K is None) == (dist is None)
        k = d = None
        if K is not None:
            k = (C.c_double * 9)(*np.asarray(K, dtype=np.float64).ravel().tolist())
            d = (C.c_double * 5)(*np.asarray(dist, dtype=np.float64).ravel()[:5].tolist())
        s = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        _lib.check(_lib.lib().avl_stem_camera_set(C.c_void_p(self.camera_block.data_ptr()), k, d, C.c_void_p(s)), "avl_stem_camera_set")

    def forward(self, image_u8=None, stream=None):
        """image_u8: CUDA/CPU uint8 [H,W,3] RGB -- or, for a raw_frame plan, the [src_h,src_w,3] BGR camera frame -- (copied
        into the plan's input buffer) or None to reuse it."""
        if image_u8 is not None:
            assert self.part is None, "a sub-plan takes its inputs through set_feature / set_low"
            if not isinstance(image_u8, torch.Tensor):
                image_u8 = torch.from_numpy(np.ascontiguousarray(image_u8))
            assert tuple(image_u8.shape) == tuple(self.image.shape) and image_u8.dtype == torch.uint8
            self.image.copy_(image_u8, non_blocking=True)
        s = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        _lib.check(_lib.lib().avl_seg_plan_run(self._plan, C.c_void_p(s)), "avl_seg_plan_run")
        return self.labels if hasattr(self, "labels_buf") else None          # (an ASPP sub-plan has no classifier: part_output())

    def capture_graph(self):
        """Record the plan into a hipGraph (one launch per forward afterwards).  Runs the plan once first, on a
        side stream (stream capture is not allowed on the legacy default stream)."""
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            _lib.check(_lib.lib().avl_seg_plan_run(self._plan, C.c_void_p(side.cuda_stream)), "avl_seg_plan_run")
            side.synchronize()
            _lib.check(_lib.lib().avl_seg_plan_capture(self._plan, C.c_void_p(side.cuda_stream)), "avl_seg_plan_capture")
        torch.cuda.current_stream(self.device).wait_stream(side)
        self.graphed = True

    def profile(self):
        """HIP-event time of every op (ms), plus its algorithmic flops and bytes -> list of dicts."""
        n = len(self.ops)
        ms, fl, by = (C.c_float * n)(), (C.c_double * n)(), (C.c_double * n)()
        s = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.lib().avl_seg_plan_profile(self._plan, C.c_void_p(s), ms, fl, by), "avl_seg_plan_profile")
        return [dict(name=self.op_names[i], kind=OP_NAMES[self.ops[i].kind], ms=ms[i], flops=fl[i], bytes=by[i]) for i in range(n)]

    def op_output(self, i):
        """float32 CPU tensor [out_h * out_w, out_c] of what op i wrote (hi + lo planes) -- valid right after a run of ops 0 .. i only (later ops
        recycle the buffers): diagnostics and tests (tools/layer_error_trace.py)."""
        op = self.ops[i]
        rows, cols = op.out_h * op.out_w, (op.in3_c if (op.kind == OP_DWPW and op.out_f32) else op.out_c)     # (the fused classifier writes in3_c logits per pixel)
        dt = torch.float32 if (op.out_f32 or op.dtype == _lib.AVL_F32) else self.act_dtype

        def plane(ptr):
            for t in self._keep + ([self.logits_buf] if hasattr(self, "logits_buf") else []):
                lo, hi = t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()
                if lo <= ptr < hi:
                    flat = t.reshape(-1).view(torch.uint8)[ptr - lo:].view(dt)
                    return torch.as_strided(flat, (rows, cols), (op.out_ld, 1)).float().cpu()
            raise KeyError("op %d: output pointer not in the plan's buffers" % i)
        y = plane(op.out)
        return y + plane(op.out_lo) if op.out_lo else y

    def run_prefix(self, n_ops, stream=None):
        """ops 0 .. n_ops - 1 only (a plan of their own): diagnostics and tests"""
        plan = C.c_void_p()
        arr = (AvlSegOp * n_ops)(*self.ops[:n_ops])
        _lib.check(_lib.lib().avl_seg_plan_create(arr, n_ops, C.byref(plan)), "avl_seg_plan_create")
        try:
            s = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
            _lib.check(_lib.lib().avl_seg_plan_run(plan, C.c_void_p(s)), "avl_seg_plan_run")
            torch.cuda.synchronize(self.device)
        finally:
            _lib.lib().avl_seg_plan_destroy(plan)

    def nonfinite_counts(self):
        """Runs the plan's ops one by one (never the captured graph) and counts Inf / NaN values in every op's output where it is
        produced -> {op name: count} of the ops that produced any.  An f16 overflow shows up here even where a later ReLU hides it."""
        n = len(self.ops)
        cnt = (C.c_ulonglong * n)()
        s = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.lib().avl_seg_plan_nonfinite(self._plan, C.c_void_p(s), cnt), "avl_seg_plan_nonfinite")
        return {self.op_names[i]: int(cnt[i]) for i in range(n) if cnt[i]}

    def total_flops(self):
        n = len(self.ops)
        return sum(r["flops"] for r in self.profile()) if n else 0.0
