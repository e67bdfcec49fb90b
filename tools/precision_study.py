"""CPU simulation of the 16-bit pipelines: where does the logits error come from?

Mirrors the plan of vision_semantic_segmentation_amd/network.py (BatchNorm folded in float64, weights rounded
to the storage type, every op output rounded to the activation type, fp32 accumulation) on torch-CPU and
switches the rounding on and off per tensor class / per stage.  Test infrastructure (uses the oracle).

  python tools/precision_study.py [H W]
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import network_oracle as NO                      # noqa: E402
from vision_semantic_segmentation_amd import network as N    # noqa: E402


def rnd(x, kind):
    if kind == "f32":
        return x
    if kind == "f16":
        return x.to(torch.float16).to(torch.float32)
    if kind == "bf16":
        return x.to(torch.bfloat16).to(torch.float32)
    if kind == "f16x2":
        hi = x.to(torch.float16).to(torch.float32)
        lo = (x - hi).to(torch.float16).to(torch.float32)
        return hi + lo
    if kind == "bf16x2":
        hi = x.to(torch.bfloat16).to(torch.float32)
        lo = (x - hi).to(torch.bfloat16).to(torch.float32)
        return hi + lo
    if kind == "bf16x3":
        hi = x.to(torch.bfloat16).to(torch.float32)
        mid = (x - hi).to(torch.bfloat16).to(torch.float32)
        lo = (x - hi - mid).to(torch.bfloat16).to(torch.float32)
        return hi + mid + lo
    raise ValueError(kind)


class Policy(object):
    """what(stage, role) -> rounding kind.  role: 'w' weights, 'a' op output, 't' trunk (block output)."""

    def __init__(self, default="f16", **over):
        self.default = default
        self.over = over

    def __call__(self, stage, role):
        # most specific first: "<stage prefix>:<role>", then "<stage prefix>", then ":<role>"
        parts = stage.split(".")
        for n in range(len(parts), 0, -1):
            pre = ".".join(parts[:n])
            if pre + ":" + role in self.over:
                return self.over[pre + ":" + role]
            if pre in self.over:
                return self.over[pre]
        return self.over.get(":" + role, self.default)


def fold(st, conv_key, bn):
    w, b = N.fold_bn(st, conv_key, bn)
    return w, b


@torch.no_grad()
def forward(st, image_u8, pol):
    """roles: w weights, a op output, t trunk as stored (what the residual add reads), ta trunk as read by conv1/downsample."""
    st = NO.strip_module_prefix(st)
    x = NO.normalize_image(image_u8)

    def conv(x, stage, ck, bn, relu=True, res=None, out_role="a", **kw):
        w, b = fold(st, ck, bn)
        w = rnd(w.to(torch.float32), pol(stage, "w"))
        y = F.conv2d(x, w, b.to(torch.float32), **kw)
        if res is not None:
            y = y + res
        if relu:
            y = F.relu(y)
        return rnd(y, pol(stage, out_role))

    x = conv(x, "stem", "backbone.conv1.weight", "backbone.bn1", stride=2, padding=3)
    x = F.max_pool2d(x, 3, 2, 1)
    low = None
    for p, stride, dil in NO.layer_plan(st, 8):
        stage = p[len("backbone."):]
        if p.startswith("backbone.layer2.0") and low is None:
            low = x
        groups = st[p + ".conv2.weight"].shape[0] // st[p + ".conv2.weight"].shape[1]
        xa = rnd(x, pol(stage + ".conv1", "ta"))
        o = conv(xa, stage + ".conv1", p + ".conv1.weight", p + ".bn1")
        o = conv(o, stage + ".conv2", p + ".conv2.weight", p + ".bn2", stride=stride, padding=dil, dilation=dil, groups=groups)
        if (p + ".downsample.0.weight") in st:
            idn = conv(xa, stage + ".down", p + ".downsample.0.weight", p + ".downsample.1", relu=False, stride=stride, out_role="t")
        else:
            idn = x
        x = conv(o, stage + ".conv3", p + ".conv3.weight", p + ".bn3", res=idn, out_role="t")
    feat = x
    fa = rnd(feat, pol("aspp", "ta"))
    dils = (1, 12, 24, 36)
    outs = [conv(fa, "aspp.b0", "aspp.module_pyramid.0.conv.weight", "aspp.module_pyramid.0.bn")]
    for i in (1, 2, 3):
        pp = "aspp.module_pyramid.%d" % i
        c = feat.shape[1]
        t = conv(fa, "aspp.b%d.dw" % i, pp + ".depthwise_cnn.conv.weight", pp + ".depthwise_cnn.bn", padding=dils[i], dilation=dils[i], groups=c)
        outs.append(conv(t, "aspp.b%d.pw" % i, pp + ".pointwise_cnn.conv.weight", pp + ".pointwise_cnn.bn"))
    g = F.adaptive_avg_pool2d(feat, (1, 1))
    g = NO.conv2d_block(g, st, "aspp.global_avg_pool.1")
    outs.append(g.expand(-1, -1, feat.shape[2], feat.shape[3]))
    cat = torch.cat(outs, 1)
    y = conv(cat, "aspp.proj", "aspp.conv.conv.weight", "aspp.conv.bn")
    lowa = rnd(low, pol("dec.low", "ta"))
    lowc = conv(lowa, "dec.low", "decoder.low_level_conv.conv.weight", "decoder.low_level_conv.bn")
    up = rnd(F.interpolate(y, size=lowc.shape[2:], mode="bilinear", align_corners=True), pol("dec.up", "a"))
    x = torch.cat([up, lowc], 1)
    for i in (0, 1):
        pp = "decoder.refine_layers.%d" % i
        c = x.shape[1]
        t = conv(x, "dec.r%d.dw" % i, pp + ".depthwise_cnn.conv.weight", pp + ".depthwise_cnn.bn", groups=c)
        x = conv(t, "dec.r%d.pw" % i, pp + ".pointwise_cnn.conv.weight", pp + ".pointwise_cnn.bn")
    w, b = fold(st, "decoder.refine_layers.2.conv.weight", None)
    w = rnd(w.to(torch.float32), pol("dec.out", "w"))
    return F.conv2d(x, w, b.to(torch.float32))


def main():
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (96, 128)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    st = N.random_state_dict(seed=0)
    img = torch.randint(0, 256, (H, W, 3), dtype=torch.uint8).numpy()
    ref = NO.forward_logits(st, img)
    den = ref.abs().max().item()

    def report(name, pol):
        y = forward(st, img, pol)
        e = (y - ref).abs()
        agree = (y.argmax(1) == ref.argmax(1)).float().mean().item()
        print("%-46s max %.3e  rms %.3e  argmax %.4f" % (name, e.max().item() / den, e.pow(2).mean().sqrt().item() / den, agree), flush=True)

    X = "f16x2"
    sel = sys.argv[3] if len(sys.argv) > 3 else "s2"
    report("all f16", Policy("f16"))
    if sel == "s2":
        report("S2: w x2, a f16, ta f16, t exact", Policy("f16", **{":w": X, ":t": "f32"}))
        report("S2 + dec a exact", Policy("f16", **{":w": X, ":t": "f32", "dec:a": "f32"}))
        report("S2 + dec a exact + aspp a exact", Policy("f16", **{":w": X, ":t": "f32", "dec:a": "f32", "aspp:a": "f32"}))
        report("S2 + dec,aspp,l1,l2 exact", Policy("f16", **{":w": X, ":t": "f32", "dec": "f32", "aspp": "f32", "layer1": "f32", "layer2": "f32"}))
        report("S2 but w f16 in l1,l2,stem", Policy("f16", **{":w": X, ":t": "f32", "layer1:w": "f16", "layer2:w": "f16", "stem:w": "f16"}))
        report("S2, t f16 in l1,l2 ", Policy("f16", **{":w": X, ":t": "f32", "layer1:t": "f16", "layer2:t": "f16"}))
        report("S2 + dec a exact, t f16 everywhere", Policy("f16", **{":w": X, "dec:a": "f32"}))
        for st_ in ("dec.up", "dec.r0.dw", "dec.r0.pw", "dec.r1.dw", "dec.r1.pw", "dec.low"):
            report("S2 + %s a exact" % st_, Policy("f16", **{":w": X, ":t": "f32", st_ + ":a": "f32"}))
    print("den (max|logit|) = %.3f" % den)


if __name__ == "__main__":
    main()
