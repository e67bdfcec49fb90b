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
    if kind.startswith("mx"):            # outside the 1x1 convs the MX modes mean "weights split exactly"
        kind = "f16x2"
    if kind == "f16":
        return x.to(torch.float16).to(torch.float32)
    if kind == "bf16":
        return x.to(torch.bfloat16).to(torch.float32)
    if kind == "f16x2":
        hi = x.to(torch.float16).to(torch.float32)
        lo = (x - hi).to(torch.float16).to(torch.float32)
        return hi + lo
    if kind in ("f16q4", "f16q6", "f16q8"):      # f16 plane + the lo part as an MX-FP4 / FP6 / FP8 copy only (blocks of 32 channels): the trunk as stored
        hi = x.to(torch.float16).to(torch.float32)
        return hi + mx_quant(x - hi, 1, {"f16q4": "fp4", "f16q6": "fp6", "f16q8": "fp8"}[kind])
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


def mx_quant(t, dim, fmt="fp4"):
    """OCP MX block quantisation along `dim`: blocks of 32, shared power-of-two scale, e2m1 (fp4) or e2m3 (fp6) elements."""
    t = t.movedim(dim, -1)
    shp = t.shape
    k = shp[-1]
    pad = (-k) % 32
    if pad:
        t = F.pad(t, (0, pad))
    b = t.reshape(-1, 32)
    amax = b.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    scale = torch.exp2(torch.floor(torch.log2(amax)) - 2.0)          # max / scale in [4, 8)
    if os.environ.get("PS_SCALE", "bump") == "bump":                 # the rule the kernels use (seg_types.h mx_fp4_scale_byte); "floor": OCP's
        scale = torch.where(amax / scale > (float(os.environ.get("PS_SAT_T", "6.5")) if fmt == "fp4" else 7.5), scale * 2, scale)
    v = (b / scale)
    if fmt == "fp8":
        # OCP e4m3 (bias 7, max 448) under an E8M0 block scale that puts the block maximum into [256, 512): round to nearest by float arithmetic
        scale8 = torch.exp2(torch.floor(torch.log2(amax)) - 8.0)
        v8 = (b / scale8).clamp(-448.0, 448.0)
        e = torch.floor(torch.log2(v8.abs().clamp_min(2.0 ** -9))).clamp_min(-6.0)          # subnormals below 2^-6
        step = torch.exp2(e - 3.0)
        q = (torch.round(v8 / step) * step * scale8).reshape(*shp[:-1], k + pad)[..., :k]
        return q.movedim(-1, dim)
    if fmt == "fp4":
        grid = torch.tensor([0, 0.5, 1, 1.5, 2, 3, 4, 6.0])
    else:
        grid = torch.cat([torch.arange(0, 2, 0.125), torch.arange(2, 4, 0.25), torch.arange(4, 8, 0.5)])
    mag = v.abs().clamp_max(float(grid[-1]))
    idx = torch.bucketize(mag, (grid[1:] + grid[:-1]) / 2)
    q = grid[idx] * torch.sign(v) * scale
    q = q.reshape(*shp[:-1], k + pad)[..., :k]
    return q.movedim(-1, dim)


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
        wk = pol(stage, "w")
        if wk.startswith("mx") and w.shape[2] == 1 and not kw.get("groups"):
            # main pass on f16 hi parts, correction passes on MX-quantised operands: Q(W lo) Q(x hi) [+ Q(W hi) Q(x lo)]
            fmt = "fp6" if wk.endswith("6") else ("fp8" if wk.endswith("8") else "fp4")
            w32 = w.to(torch.float32)
            wh = rnd(w32, "f16")
            wl = (w.to(torch.float64) - wh.to(torch.float64)).to(torch.float32)
            xh = rnd(x, "f16")
            xl = x - xh
            y = F.conv2d(xh, wh, b.to(torch.float32), **kw)
            # PS_SKIP_WL = comma-separated stage prefixes whose Q(W lo) . Q(x hi) term is dropped (is the FP4 copy of a
            # trunk tensor's HI part worth its bytes?)
            skip_wl = any(stage.startswith(pre) for pre in os.environ.get("PS_SKIP_WL", "").split(",") if pre)
            if not skip_wl:
                y = y + F.conv2d(mx_quant(xh, 1, fmt), mx_quant(wl, 1, fmt), None, **kw)
            if float(xl.abs().max()) > 0:
                y = y + F.conv2d(mx_quant(xl, 1, fmt), mx_quant(wh, 1, fmt), None, **kw)
        else:
            w = rnd(w.to(torch.float32), "f16x2" if wk.startswith("mx") else wk)
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
    dils = (1, 12, 24, 36)
    outs = [conv(rnd(feat, pol("aspp.b0", "ta")), "aspp.b0", "aspp.module_pyramid.0.conv.weight", "aspp.module_pyramid.0.bn")]
    for i in (1, 2, 3):
        pp = "aspp.module_pyramid.%d" % i
        c = feat.shape[1]
        t = conv(rnd(feat, pol("aspp.b%d.dw" % i, "ta")), "aspp.b%d.dw" % i, pp + ".depthwise_cnn.conv.weight", pp + ".depthwise_cnn.bn", padding=dils[i], dilation=dils[i], groups=c)
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
    st = N.random_state_dict(seed=int(os.environ.get("PS_SEED", "0")))
    if os.environ.get("PS_HEAVY"):          # a checkpoint-like draw with calibrated BatchNorm statistics (tests/_full_size.heavy_tailed_state_dict)
        sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
        import _full_size as fs
        st = fs.heavy_tailed_state_dict(int(os.environ.get("PS_SEED", "0")))
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
    if sel == "v":
        base = {":w": X, ":t": "f32", "dec:a": "f32", "aspp:a": "f32"}
        report("V1: S2 + dec,aspp a exact", Policy("f16", **base))
        report("V2: V1 + l3/l4 ta exact", Policy("f16", **dict(base, **{"layer3:ta": "f32", "layer4:ta": "f32"})))
        report("V3: V1 + l1/l2 ta exact", Policy("f16", **dict(base, **{"layer1:ta": "f32", "layer2:ta": "f32"})))
        report("V4: V1 + all ta exact", Policy("f16", **dict(base, **{":ta": "f32"})))
        d5 = dict(base, **{":ta": "f32"})
        for li, nb in ((1, 3), (2, 4), (3, 6), (4, 3)):
            for b in range(nb):
                d5["layer%d.%d.conv2:a" % (li, b)] = "f32"
        report("V5: V4 + conv2 out exact", Policy("f16", **d5))
        d6 = dict(base, **{":ta": "f32"})
        for li, nb in ((1, 3), (2, 4), (3, 6), (4, 3)):
            for b in range(nb):
                d6["layer%d.%d.conv1:a" % (li, b)] = "f32"
        report("V6: V4 + conv1 out exact", Policy("f16", **d6))
        report("V7: V4 but aspp a f16", Policy("f16", **{":w": X, ":t": "f32", "dec:a": "f32", ":ta": "f32"}))
        report("V8: V4 but stem exact too", Policy("f16", **dict(base, **{":ta": "f32", "stem": "f32"})))
    if sel == "v9":
        d = {":w": X, ":t": "f32", ":ta": "f32", "stem": "f16", "dec": "f32", "dec:w": X,
             "aspp.b0:a": "f32", "aspp.proj:a": "f32"}
        for i in (1, 2, 3):
            d["aspp.b%d.dw" % i] = "f16"          # reads hi only, f16 depthwise weights, f16 A tile
            d["aspp.b%d.pw:a" % i] = "f32"        # split output
        report("V9: the scheme to build", Policy("f16", **d))
        report("V9 + stem w x2", Policy("f16", **dict(d, **{"stem:w": X})))
        d2 = dict(d)
        for i in (1, 2, 3):
            d2["aspp.b%d.dw:ta" % i] = "f32"
        report("V9 + aspp dw reads split", Policy("f16", **d2))
        report("V9 but dec dw weights f16", Policy("f16", **dict(d, **{"dec.r0.dw:w": "f16", "dec.r1.dw:w": "f16"})))
    if sel == "c":
        c1 = {":w": X, ":t": "f32", "stem": "f16", "dec": "f32", "dec:w": X, "dec.low:ta": "f32",
              "aspp.b0:a": "f32", "aspp.b0:ta": "f32", "aspp.proj:a": "f32"}
        for i in (1, 2, 3):
            c1["aspp.b%d.dw" % i] = "f16"
            c1["aspp.b%d.pw:a" % i] = "f32"
        report("C1: S2 + aspp/dec exact (conv1 reads hi)", Policy("f16", **c1))
        c2 = dict(c1, **{"layer3:ta": "f32", "layer4:ta": "f32"})
        report("C2: C1 + l3/l4 conv1,down read split", Policy("f16", **c2))
        c3 = dict(c1)
        c4 = dict(c1)
        for li, nb in ((3, 6), (4, 3)):
            for b in range(nb):
                c3["layer%d.%d.conv2:a" % (li, b)] = "f32"
                c4["layer%d.%d.conv1:a" % (li, b)] = "f32"
        report("C3: C1 + l3/l4 conv2 out split", Policy("f16", **c3))
        report("C4: C1 + l3/l4 conv1 out split", Policy("f16", **c4))
        c5 = dict(c3)
        for li, nb in ((1, 3), (2, 4)):
            for b in range(nb):
                c5["layer%d.%d.conv2:a" % (li, b)] = "f32"
        report("C5: C3 + l1/l2 conv2 out split", Policy("f16", **c5))
        report("C6: C2 + C3", Policy("f16", **dict(c3, **{"layer3:ta": "f32", "layer4:ta": "f32"})))
    if sel == "w":
        d = {":w": X, ":t": "f32", ":ta": "f32", "stem": "f16", "dec": "f32", "dec:w": X,
             "aspp.b0:a": "f32", "aspp.proj:a": "f32"}
        for i in (1, 2, 3):
            d["aspp.b%d.dw" % i] = "f16"
            d["aspp.b%d.pw:a" % i] = "f32"
        report("built (V9)", Policy("f16", **d))
        def blocks(layers, conv, dd):
            for li, nb in layers:
                for b in range(nb):
                    dd["layer%d.%d.%s:a" % (li, b, conv)] = "f32"
            return dd
        L34, L12 = ((3, 6), (4, 3)), ((1, 3), (2, 4))
        report("+ l3/4 conv1 out split", Policy("f16", **blocks(L34, "conv1", dict(d))))
        report("+ l1-4 conv1 out split", Policy("f16", **blocks(L34 + L12, "conv1", dict(d))))
        report("+ l1/2 conv1+conv2 out split", Policy("f16", **blocks(L12, "conv2", blocks(L12, "conv1", dict(d)))))
        report("+ l1-4 conv1 split, l1/2 conv2 split", Policy("f16", **blocks(L12, "conv2", blocks(L34 + L12, "conv1", dict(d)))))
        report("+ l1-4 conv1 split, l1/2 conv2 split, stem w x2", Policy("f16", **dict(blocks(L12, "conv2", blocks(L34 + L12, "conv1", dict(d))), **{"stem:w": X})))
        report("+ all conv1, conv2 split", Policy("f16", **blocks(L34 + L12, "conv2", blocks(L34 + L12, "conv1", dict(d)))))
        report("+ all conv1, conv2 split, stem exact, aspp dw exact", Policy("f16", **dict(blocks(L34 + L12, "conv2", blocks(L34 + L12, "conv1", dict(d))), **{"stem": "f32", "aspp.b1.dw": "f32", "aspp.b2.dw": "f32", "aspp.b3.dw": "f32"})))
    if sel == "mx":
        def blocks(layers, conv, dd):
            for li, nb in layers:
                for b in range(nb):
                    dd["layer%d.%d.%s:a" % (li, b, conv)] = "f32"
            return dd
        L34, L12 = ((3, 6), (4, 3)), ((1, 3), (2, 4))
        for fmt in ("mx4", "mx6"):
            d = {":w": fmt, ":t": "f32", ":ta": "f32", "stem": "f16", "dec": "f32", "dec:w": fmt,
                 "aspp.b0:a": "f32", "aspp.proj:a": "f32"}
            for i in (1, 2, 3):
                d["aspp.b%d.dw" % i] = "f16"
                d["aspp.b%d.pw:a" % i] = "f32"
            report("built config, corrections in %s" % fmt, Policy("f16", **d))
            report("  + all conv1, conv2 out split (%s)" % fmt, Policy("f16", **blocks(L34 + L12, "conv2", blocks(L34 + L12, "conv1", dict(d)))))
    if sel == "asp":
        def blocks(layers, conv, dd):
            for li, nb in layers:
                for b in range(nb):
                    dd["layer%d.%d.%s:a" % (li, b, conv)] = "f32"
            return dd
        L34, L12 = ((3, 6), (4, 3)), ((1, 3), (2, 4))
        d = {":w": "mx4", ":t": "f32", ":ta": "f32", "stem": "f16", "dec": "f32", "dec:w": "mx4",
             "aspp.b0:a": "f32", "aspp.proj:a": "f32"}
        for i in (1, 2, 3):
            d["aspp.b%d.dw" % i] = "f16"
            d["aspp.b%d.pw:a" % i] = "f32"
        d = blocks(L34 + L12, "conv2", d)
        report("built + conv2 split everywhere", Policy("f16", **d))
        report("  + stem weights split", Policy("f16", **dict(d, **{"stem:w": X})))
        report("  + stem weights split + stem out exact", Policy("f16", **dict(d, **{"stem:w": X, "stem:a": "f32"})))
        e = dict(d)
        for i in (1, 2, 3):
            e["aspp.b%d.dw:w" % i] = "f32"
        report("  + aspp dw weights fp32", Policy("f16", **e))
        e2 = dict(e)
        for i in (1, 2, 3):
            e2["aspp.b%d.dw:a" % i] = "f32"
        report("  + aspp dw weights fp32 + dw out split", Policy("f16", **e2))
        e3 = dict(e2)
        for i in (1, 2, 3):
            e3["aspp.b%d.dw:ta" % i] = "f32"
        report("  + aspp dw all exact", Policy("f16", **e3))
        report("  + aspp dw all exact + stem w split + stem out exact", Policy("f16", **dict(e3, **{"stem:w": X, "stem:a": "f32"})))
        report("  + aspp dw out split only (weights f16)", Policy("f16", **dict(d, **{"aspp.b1.dw:a": "f32", "aspp.b2.dw:a": "f32", "aspp.b3.dw:a": "f32"})))
    if sel == "wl":
        # the built configuration (gconv_mx, trunk f16 + FP4 lo) with the weights' lo correction dropped for the conv1s of a layer
        def blocks(which, conv, dd):
            for li, b in which:
                dd["layer%d.%d.%s:a" % (li, b, conv)] = "f32"
            return dd
        ALL = [(li, b) for li, nb in ((1, 3), (2, 4), (3, 6), (4, 3)) for b in range(nb)]
        MXB = [(li, b) for (li, b) in ALL if li >= 3 or (li == 2 and b >= 1)]
        d = {":w": "mx4", ":t": "f16q4", ":ta": "f32", "stem": "f16", "dec": "f32", "dec:w": "mx4", "aspp.b0:a": "f32", "aspp.proj:a": "f32"}
        for i in (1, 2, 3):
            d["aspp.b%d.dw:w" % i] = "f32"
            d["aspp.b%d.dw:a" % i] = "f32"
            d["aspp.b%d.pw:a" % i] = "f32"
        g = blocks(MXB, "conv1", blocks(ALL, "conv2", d))
        for skip in ("", "layer4.1.conv1,layer4.2.conv1", "layer4.0.conv1,layer4.1.conv1,layer4.2.conv1", "layer3", "layer4", "layer3,layer4"):
            os.environ["PS_SKIP_WL"] = skip
            report("built config; W lo term dropped in: %s" % (skip or "-"), Policy("f16", **g))
        os.environ["PS_SKIP_WL"] = ""
    if sel == "heavy":
        # round 5: which roundings does a CALIBRATED network (every BN subtracts the mean of what it normalises) not forgive?
        def blocks(which, conv, dd, kind="f32"):
            for li, b in which:
                dd["layer%d.%d.%s:a" % (li, b, conv)] = kind
            return dd
        ALL = [(li, b) for li, nb in ((1, 3), (2, 4), (3, 6), (4, 3)) for b in range(nb)]
        report("everything f16 x2 (weights, every tensor)", Policy(X))
        report("  ... but the stem's output one f16 plane", Policy(X, **{"stem:a": "f16"}))
        report("  ... but the stem all f16 (weights too)", Policy(X, **{"stem": "f16"}))
        report("  ... but conv1 outputs one f16 plane", Policy(X, **blocks(ALL, "conv1", {}, "f16")))
        report("  ... but conv2 outputs one f16 plane", Policy(X, **blocks(ALL, "conv2", {}, "f16")))
        report("  ... but the trunk f16 + FP4(lo)", Policy(X, **{":t": "f16q4"}))
        report("  ... but the trunk one f16 plane", Policy(X, **{":t": "f16"}))
        report("  ... but aspp depthwise stage f16", Policy(X, **{"aspp.b1.dw": "f16", "aspp.b2.dw": "f16", "aspp.b3.dw": "f16"}))
        report("  ... but the decoder's tensors one f16 plane", Policy(X, **{"dec:a": "f16"}))
        report("  ... but weights one f16 plane", Policy(X, **{":w": "f16"}))
        report("  ... stem f16 + conv1 outputs f16 (round 4's split16)", Policy(X, **blocks(ALL, "conv1", {"stem": "f16"}, "f16")))
        # what would a cheaper complete pipeline look like?  lo parts as block-scaled FP8 (e4m3: 4 significant bits on a 2^-11 term; the
        # scaled matrix cores run FP8 at twice the f16 rate: 1 + 1/2 + 1/2 = 2 passes instead of 3)
        report("every tensor f16 + FP8(lo), weights x2", Policy("f16q8", **{":w": X}))
        report("every tensor f16 x2, 1x1 weights f16 + FP8(lo) products", Policy(X, **{":w": "mx8"}))
        report("every tensor f16 + FP8(lo), 1x1 corrections in FP8", Policy("f16q8", **{":w": "mx8"}))
        report("every tensor f16 + FP6(lo), weights x2", Policy("f16q6", **{":w": X}))
    if sel == "r3s":
        def blocks(which, conv, dd):
            for li, b in which:
                dd["layer%d.%d.%s:a" % (li, b, conv)] = "f32"
            return dd
        ALL = [(li, b) for li, nb in ((1, 3), (2, 4), (3, 6), (4, 3)) for b in range(nb)]
        MXB = [(li, b) for (li, b) in ALL if li >= 3 or (li == 2 and b >= 1)]
        d = {":w": "mx4", ":t": "f32", ":ta": "f32", "stem": "f16", "dec": "f32", "dec:w": "mx4",
             "aspp.b0:a": "f32", "aspp.proj:a": "f32"}
        for i in (1, 2, 3):
            d["aspp.b%d.dw" % i] = "f16"
            d["aspp.b%d.pw:a" % i] = "f32"
        g = blocks(MXB, "conv1", blocks(ALL, "conv2", d))
        report("gconv_mx, PS_SCALE=%s" % os.environ.get("PS_SCALE", "floor"), Policy("f16", **g))
    if sel == "r3":
        # round 3: what is left once the grouped conv corrects for conv1's output too (MODEL.MIXED_GCONV_MX, layer2.1 onwards)?
        def blocks(which, conv, dd):
            for li, b in which:
                dd["layer%d.%d.%s:a" % (li, b, conv)] = "f32"
            return dd
        ALL = [(li, b) for li, nb in ((1, 3), (2, 4), (3, 6), (4, 3)) for b in range(nb)]
        MXB = [(li, b) for (li, b) in ALL if li >= 3 or (li == 2 and b >= 1)]            # blocks whose conv1 runs as an MX GEMM
        d = {":w": "mx4", ":t": "f32", ":ta": "f32", "stem": "f16", "dec": "f32", "dec:w": "mx4",
             "aspp.b0:a": "f32", "aspp.proj:a": "f32"}
        for i in (1, 2, 3):
            d["aspp.b%d.dw" % i] = "f16"
            d["aspp.b%d.pw:a" % i] = "f32"
        d = blocks(ALL, "conv2", d)
        report("default (conv2 split everywhere)", Policy("f16", **d))
        g = blocks(MXB, "conv1", dict(d))
        report("gconv_mx: + conv1 split in MX blocks", Policy("f16", **g))
        report("  + conv1 split in layer1, layer2.0 too", Policy("f16", **blocks(ALL, "conv1", dict(g))))
        report("  + stem weights split", Policy("f16", **dict(g, **{"stem:w": X})))
        report("  + stem weights split + stem out exact", Policy("f16", **dict(g, **{"stem:w": X, "stem:a": "f32"})))
        e = dict(g)
        for i in (1, 2, 3):
            e["aspp.b%d.dw:a" % i] = "f32"
        report("  + aspp dw out split", Policy("f16", **e))
        e2 = dict(e)
        for i in (1, 2, 3):
            e2["aspp.b%d.dw:w" % i] = "f32"
            e2["aspp.b%d.dw:ta" % i] = "f32"
        report("  + aspp dw all exact", Policy("f16", **e2))
        report("  + everything above", Policy("f16", **dict(blocks(ALL, "conv1", dict(e2)), **{"stem:w": X, "stem:a": "f32"})))
        report("gconv_mx, trunk stored as f16 + FP4(lo)", Policy("f16", **dict(g, **{":t": "f16q4"})))
        report("gconv_mx, trunk stored as f16 + FP6(lo)", Policy("f16", **dict(g, **{":t": "f16q6"})))
        report("gconv_mx, trunk f16 + FP4(lo), + aspp dw all exact", Policy("f16", **dict(e2, **{":t": "f16q4"})))
        g6 = dict(g, **{":w": "mx6", "dec:w": "mx6"})
        report("gconv_mx with FP6 corrections", Policy("f16", **g6))
    print("den (max|logit|) = %.3f" % den)


if __name__ == "__main__":
    main()
