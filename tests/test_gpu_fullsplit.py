"""The COMPLETE hi + lo pipeline (SegNet(full_split=True); the "split16" rung of MODEL.MIXED_SELF_CHECK's ladder): every
tensor two f16 planes, every product three f16 passes (Wh.xh + Wl.xh + Wh.xl), no FP4 and no single-plane tensor anywhere.  The kernels
that exist only for it -- grouped 3x3 with a split INPUT, split stem, split max-pool -- alone against float64 (3e-6 of max|ref|), then the
network: on the seeded weights far inside 1e-3, and on CALIBRATED heavy-tailed weights -- where every other 16-bit plan is 3e-2 .. 2e-1 --
within 1e-3 of the oracle (the reference loads a trained checkpoint: src/semantic_segmentation.py:28-32)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 3e-6


def _split(x64):
    import torch
    hi = x64.to(torch.float16)
    lo = (x64 - hi.double()).to(torch.float16)
    return hi, lo


@pytest.mark.parametrize("case", [(23, 45, 128, 1, 1), (30, 41, 512, 1, 2), (19, 67, 1024, 1, 4), (40, 70, 256, 1, 1), (9, 200, 64, 1, 1)])
def test_grouped_conv_with_split_input_weights_and_output(case, cuda_device):
    import torch
    import torch.nn.functional as F
    from test_gpu_ops import _from_rows, _nhwc_rows, _run_plan, _spatial_op
    from vision_semantic_segmentation_amd import _lib
    from vision_semantic_segmentation_amd.network import OP_GCONV, pack_gconv_windows
    H, W, width, s, d = case
    G = 32
    cg = width // G
    g = torch.Generator().manual_seed(H * 77 + W + width)
    x64 = torch.randn((1, width, H, W), generator=g, dtype=torch.float64)
    xh, xl = _split(x64)
    w64 = torch.randn((width, cg, 3, 3), generator=g, dtype=torch.float64) * (2.0 / (cg * 9)) ** 0.5
    b = torch.randn(width, generator=g) * 0.1
    OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
    w_hi, w_lo = _split(w64)
    ref = F.relu(F.conv2d(xh.double() + xl.double(), w_hi.double() + w_lo.double(), b.double(), stride=s, padding=d, dilation=d, groups=G))
    src = torch.stack([_nhwc_rows(xh), _nhwc_rows(xl)]).to(cuda_device)
    dst = torch.full((2, (OH * OW + 255) // 256 * 256, width), 7.0, dtype=torch.float16, device=cuda_device)
    nwin = width // 32
    wd = torch.cat([pack_gconv_windows(w_hi.double(), G).reshape(nwin, 2, 9, 16, 32),
                    pack_gconv_windows(w_lo.double(), G).reshape(nwin, 2, 9, 16, 32)], dim=2).reshape(-1).to(torch.float16).to(cuda_device)
    bd = b.to(cuda_device)
    _run_plan([_spatial_op(OP_GCONV, _lib.AVL_F16, src[0], (H, W), width, dst[0], (OH, OW), width, weight=wd.data_ptr(), bias=bd.data_ptr(),
                           ksize=3, stride=s, pad=d, dil=d, groups=G, relu=1, w_layout=1, w_split=1, in_lo=src[1].data_ptr(), out_lo=dst[1].data_ptr())])
    got = _from_rows(dst[0].cpu().double(), OH, OW, width) + _from_rows(dst[1].cpu().double(), OH, OW, width)
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err <= TOL, "grouped conv, split input %s: %.3e" % (case, err)
    assert torch.all(dst[:, OH * OW:] == 7.0)


def test_split_stem_and_maxpool(cuda_device):
    """the first two ops of a full_split plan against float64 on the exact normalised image: stem (normalise -> 7x7 s2 -> BN -> ReLU) keeps
    ~22 bits through input, weights and output; the max-pool picks the larger VALUE hi + lo and hands on that element's own pair"""
    import torch
    import torch.nn.functional as F
    import _full_size as fs
    from oracle import network_oracle as no
    from vision_semantic_segmentation_amd.network import SegNet, fold_bn
    st = fs.state_dict(0)
    h, w = 96, 160
    img = np.random.default_rng(4).integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    net = SegNet(st, h, w, precision="mixed", device=cuda_device, full_split=True)
    assert net.op_names[:2] == ["backbone.conv1", "backbone.maxpool"] and net.ops[0].out_lo and net.ops[1].in_lo and net.ops[1].out_lo
    net.image.copy_(torch.from_numpy(img).to(cuda_device))
    wf, bf = fold_bn(st, "backbone.conv1.weight", "backbone.bn1")
    whi, wlo = _split(wf)
    x = no.normalize_image(img).double()                 # (float32 divisions, as the kernel's table)
    xh, xl = _split(x)
    y = F.relu(F.conv2d(xh.double() + xl.double(), whi.double() + wlo.double(), bf, stride=2, padding=3))
    net.run_prefix(1)
    got = net.op_output(0).double()
    ref = y[0].permute(1, 2, 0).reshape(-1, 64)
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err <= TOL, "split stem: %.3e" % err
    net.run_prefix(2)
    got = net.op_output(1).double()
    ref = F.max_pool2d(y, 3, 2, 1)[0].permute(1, 2, 0).reshape(-1, 64)
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err <= TOL, "split max-pool: %.3e" % err


@pytest.mark.parametrize("hw", [(96, 128), (320, 416)])
def test_full_split_logits_on_the_seeded_weights(hw, cuda_device):
    import torch
    import _full_size as fs
    from vision_semantic_segmentation_amd.network import OP_DWPW, SegNet
    h, w = hw
    ref = fs.oracle_logits(0, 3, h, w)
    net = SegNet(fs.state_dict(0), h, w, precision="mixed", device=cuda_device, full_split=True)
    # no FP4 anywhere (w_split 2 of the fused depthwise op means "exact depthwise stage", out_mx of the fp32 classifier is the label map)
    assert not any(((op.w_split == 2 or op.out_mx) and not op.out_f32) or op.in_mx for op in net.ops)
    assert all(op.in_lo and (op.out_lo or op.out_f32) and op.w_split == 3 for op in net.ops if op.kind == OP_DWPW) and sum(op.kind == OP_DWPW for op in net.ops) == 5
    assert all(op.out_lo for n, op in zip(net.op_names, net.ops) if op.kind in (1, 2, 3, 4, 5, 6) and not op.out_f32), \
        [n for n, op in zip(net.op_names, net.ops) if op.kind in (1, 2, 3, 4, 5, 6) and not op.out_f32 and not op.out_lo]
    net.forward(torch.from_numpy(fs.image_for(3, h, w)).to(cuda_device))
    got = net.logits.permute(2, 0, 1).float().cpu()
    err = float((got - ref).abs().max() / ref.abs().max())
    print("full_split %dx%d, seeded weights: %.2e of max|logit|" % (h, w, err))
    assert err <= 5e-5
    assert net.nonfinite_counts() == {}


@pytest.mark.parametrize("wseed", [0, 1, 2, 3])
def test_full_split_holds_1e3_on_calibrated_heavy_tailed_weights(wseed, cuda_device):
    """every other 16-bit plan is 3e-2 .. 2e-1 here (tests/test_gpu_robust.py); the fp32-input HIP plan itself 3e-5 .. 8e-4"""
    import torch
    import _full_size as fs
    from oracle import network_oracle as no
    from vision_semantic_segmentation_amd.network import SegNet
    st = fs.heavy_tailed_state_dict(wseed)
    h, w = 320, 416
    img = np.random.default_rng(50 + wseed).integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    ref = no.forward_logits(st, img)[0]
    net = SegNet(st, h, w, precision="mixed", device=cuda_device, full_split=True)
    net.forward(torch.from_numpy(img).to(cuda_device))
    got = net.logits.permute(2, 0, 1).float().cpu()
    err = float((got - ref).abs().max() / ref.abs().max())
    agree = float((got.argmax(0) == ref.argmax(0)).float().mean())
    print("full_split, heavy-tailed calibrated weights %d at %dx%d: %.2e of max|logit|, arg-max agreement %.4f" % (wseed, h, w, err, agree))
    assert err <= 1e-3 and agree >= 0.998


def test_full_split_plan_with_the_preprocessing_stem(cuda_device):
    """the split stem with the node's pre-processing in its loader (k_stem_mfma<f16, PRE, SPLIT>: raw BGR frame in): same bits as
    avl_preprocess_image followed by the plain full_split plan"""
    import torch
    import _full_size as fs
    from vision_semantic_segmentation_amd.camera import camera_setup_1
    from vision_semantic_segmentation_amd.network import SegNet
    from vision_semantic_segmentation_amd.vision_semantic_segmentation_node import preprocess_device
    rng = np.random.default_rng(77)
    bgr = rng.integers(0, 256, size=(192, 256, 3), dtype=np.uint8)
    cam = camera_setup_1().scaled(256 / 1920.0, 192 / 1440.0)
    st = fs.state_dict(0)
    for factor in (1, 2):
        rgb = preprocess_device(bgr, cam, factor)
        h, w = int(rgb.shape[0]), int(rgb.shape[1])
        plain = SegNet(st, h, w, precision="mixed", device=cuda_device, full_split=True)
        plain.forward(rgb)
        want = plain.logits.clone()
        raw = SegNet(st, h, w, precision="mixed", device=cuda_device, full_split=True, raw_frame=(192, 256))
        raw.set_camera(cam.K, cam.dist)
        raw.forward(torch.from_numpy(bgr).to(cuda_device))
        assert torch.equal(raw.logits, want), factor
