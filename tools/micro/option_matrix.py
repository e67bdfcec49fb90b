"""Builder options and decoder widths off the beaten path: does every combination still build, run and match the torch-CPU oracle?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from oracle import network_oracle as no
from vision_semantic_segmentation_amd.network import SegNet, random_state_dict
dev = torch.device("cuda:0")
h, w = 96, 128
img = np.random.default_rng(4).integers(0, 256, size=(h, w, 3), dtype=np.uint8)
CASES = [({}, "mixed", {}), ({}, "mixed", dict(fuse_dwpw=False)), ({}, "mixed", dict(fuse_decoder=False)), ({}, "mixed", dict(fuse_classifier=False)),
         ({}, "mixed", dict(dw_exact=False)), ({}, "mixed", dict(full_split=True)), ({}, "mixed", dict(full_split=True, fuse_dwpw=False)),
         ({}, "mixed", dict(fuse_block=False)), ({}, "mixed", dict(layer1_lo=False)), ({}, "mixed", dict(mx=False)), ({}, "mixed", dict(gconv_mx=False)),
         (dict(low_level_out=48), "mixed", {}), (dict(low_level_out=48), "f16", {}), (dict(low_level_out=48), "f32", {}),
         (dict(low_level_out=64), "mixed", {}), (dict(low_level_out=128), "mixed", dict(full_split=True)),
         ({}, "f16", {}), ({}, "bf16", {}), ({}, "f16", dict(fuse_dwpw=False))]
for spec, prec, opts in CASES:
    try:
        state = random_state_dict(0, **spec)
        ref = no.forward_logits(state, img)[0]
        net = SegNet(state, h, w, precision=prec, device=dev, **opts)
        net.forward(torch.from_numpy(img).to(dev))
        got = net.logits.permute(2, 0, 1).float().cpu()
        err = float((got - ref).abs().max() / ref.abs().max())
        ok = bool(np.array_equal(net.labels.cpu().numpy(), got.argmax(0).numpy()))
        print("%-22s %-6s %-40s ops %2d: logits %.2e, labels == argmax: %s" % (spec, prec, opts, len(net.ops), err, ok), flush=True)
    except Exception as e:
        print("%-22s %-6s %-40s FAILED %s: %s" % (spec, prec, opts, type(e).__name__, str(e)[:200]), flush=True)
