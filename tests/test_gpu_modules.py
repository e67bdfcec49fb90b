"""The HIP kernels against the reference MODULES' own outputs: tests/golden/net_aspp256.pt and net_decoder256.pt were
produced by running the reference's AtrousSpatialPyramidPoolingModule (aspp.py:79-95) and Decoder (decoder.py:45-51) themselves
(oracle/gen_golden.py imports them) at widths the kernels take -- 256-channel branches -- so the comparison is direct, not through the
oracle: a sub-plan (SegNet(part=...)) is built from the module's state dict and its output compared with the module's `y`.
f32: fp32-input MFMA, 1e-5 of max|y|.  mixed: f16 + split planes, 1e-3.  The conv weights and inputs of the fixtures are float16 values,
so the 16-bit planes hold them exactly."""
import os

import pytest

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    import torch
    d = torch.load(os.path.join(golden_dir, name), map_location="cpu", weights_only=True)
    return {k: (v.float() if hasattr(v, "float") else {kk: vv.float() for kk, vv in v.items()}) for k, v in d.items()}


@pytest.mark.parametrize("precision,bar", [("f32", 1e-5), ("mixed", 1e-3), ("f16", 4e-3)])
def test_aspp_kernels_against_the_reference_module(precision, bar, golden_dir, cuda_device):
    import torch
    from vision_semantic_segmentation_amd.network import SegNet
    d = _load(golden_dir, "net_aspp256.pt")
    st = {"aspp." + k: v for k, v in d["state"].items() if not k.endswith("num_batches_tracked")}
    x, y = d["x"][0], d["y"][0]
    net = SegNet(st, x.shape[1], x.shape[2], precision=precision, device=cuda_device, part=("aspp", x.shape[0]))
    net.set_feature(x)
    net.forward()
    torch.cuda.synchronize()
    got = net.part_output().cpu()
    err = float((got - y).abs().max() / y.abs().max())
    print("ASPP 256 -> 256 at %dx%d, %s: %.2e of max|y| against the reference module" % (x.shape[1], x.shape[2], precision, err))
    assert got.shape == y.shape and err <= bar


@pytest.mark.parametrize("precision,bar", [("f32", 1e-5), ("mixed", 1e-3), ("f16", 4e-3)])
def test_decoder_kernels_against_the_reference_module(precision, bar, golden_dir, cuda_device):
    import torch
    from vision_semantic_segmentation_amd.network import SegNet
    d = _load(golden_dir, "net_decoder256.pt")
    st = {"decoder." + k: v for k, v in d["state"].items() if not k.endswith("num_batches_tracked")}
    f, low, y = d["feature"][0], d["low"][0], d["y"][0]
    net = SegNet(st, f.shape[1], f.shape[2], precision=precision, device=cuda_device, part=("decoder", f.shape[0], low.shape[0]))
    net.set_feature(f)
    net.set_low(low)
    net.forward()
    torch.cuda.synchronize()
    got = net.logits.permute(2, 0, 1).float().cpu()
    err = float((got - y).abs().max() / y.abs().max())
    print("Decoder 256 + 256 -> 19 at %dx%d / %dx%d, %s: %.2e of max|y| against the reference module" % (f.shape[1], f.shape[2], low.shape[1], low.shape[2], precision, err))
    assert got.shape == y.shape == (19, 2 * f.shape[1] - 4, 2 * f.shape[2] - 4) and err <= bar
