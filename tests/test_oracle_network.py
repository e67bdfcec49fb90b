"""The torch-CPU network oracle against outputs of the REFERENCE's own ASPP / Decoder / Conv2d /
DepthwiseSeparableConv2d modules (tests/golden/net_*.pt, made by oracle/gen_golden.py)."""
import os

import torch

from oracle import network_oracle as no

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return torch.load(os.path.join(GOLD, name), weights_only=True)


def test_aspp_matches_reference_module():
    g = _load("net_aspp.pt")
    st = {"aspp." + k: v for k, v in g["state"].items()}
    y = no.aspp_forward(st, g["x"])
    assert y.shape == g["y"].shape
    assert torch.allclose(y, g["y"], rtol=0, atol=1e-6)


def test_decoder_matches_reference_module():
    g = _load("net_decoder.pt")
    st = {"decoder." + k: v for k, v in g["state"].items()}
    y = no.decoder_forward(st, g["feature"], g["low"])
    assert tuple(y.shape) == (1, 19, 36, 46)          # (H/4 - 4) x (W/4 - 4): the two pad-0 depthwise convs
    assert torch.allclose(y, g["y"], rtol=0, atol=1e-6)


def test_conv_blocks_match_reference_modules():
    g = _load("net_blocks.pt")
    st = {"c." + k: v for k, v in g["conv_state"].items()}
    y = no.conv2d_block(g["x"], st, "c", stride=2, padding=2, dilation=2, groups=4)
    assert torch.allclose(y, g["y_conv"], rtol=0, atol=1e-6)
    st = {"d." + k: v for k, v in g["dw_state"].items()}
    y = no.depthwise_separable(g["x"], st, "d", padding=3, dilation=3, pointwise_relu=False)
    assert torch.allclose(y, g["y_dw"], rtol=0, atol=1e-6)


def test_oracle_matches_the_kernel_width_fixtures(golden_dir):
    """round 5: the same two reference modules at 256-channel widths (what tests/test_gpu_modules.py feeds the HIP kernels)"""
    a = torch.load(os.path.join(golden_dir, "net_aspp256.pt"), map_location="cpu", weights_only=True)
    st = {"aspp." + k: v.float() for k, v in a["state"].items()}
    y = no.aspp_forward(st, a["x"].float())
    assert float((y - a["y"]).abs().max() / a["y"].abs().max()) <= 1e-6
    d = torch.load(os.path.join(golden_dir, "net_decoder256.pt"), map_location="cpu", weights_only=True)
    st = {"decoder." + k: v.float() for k, v in d["state"].items()}
    y = no.decoder_forward(st, d["feature"].float(), d["low"].float())
    assert float((y - d["y"]).abs().max() / d["y"].abs().max()) <= 1e-6
