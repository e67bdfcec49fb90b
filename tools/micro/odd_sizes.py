"""Does the plan hold at image sizes that are not multiples of 4 / 8 / 16?  HIP logits against the torch-CPU oracle."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from oracle import network_oracle as no
from vision_semantic_segmentation_amd.network import SegNet, random_state_dict
dev = torch.device("cuda:0")
state = random_state_dict(0)
SIZES = ((97, 131), (250, 333), (375, 1242), (121, 160), (66, 70)) if len(sys.argv) < 2 else ((20, 20), (21, 37), (24, 64), (33, 35), (40, 2000), (2000, 40))     # any argument: tiny and extreme aspect ratios
for (h, w) in SIZES:
    img = np.random.default_rng(h).integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    ref = no.forward_logits(state, img)[0]
    for prec, opts in (("f32", {}), ("mixed", {}), ("mixed", dict(full_split=True)), ("bf16", {})):
        try:
            net = SegNet(state, h, w, precision=prec, device=dev, **opts)
            net.forward(torch.from_numpy(img).to(dev))
            got = net.logits.permute(2, 0, 1).float().cpu()
            assert got.shape == ref.shape, (got.shape, ref.shape)
            err = float((got - ref).abs().max() / ref.abs().max())
            lab = net.labels.cpu().numpy()
            print("%4d x %4d %-6s%-8s logits %s: %.2e of max|logit|, labels == argmax: %s" % (h, w, prec, "+split" if opts else "", tuple(got.shape), err,
                                                                                        bool(np.array_equal(lab, got.argmax(0).numpy()))), flush=True)
        except Exception as e:
            print("%4d x %4d %-6s%-8s FAILED: %s: %s" % (h, w, prec, "+split" if opts else "", type(e).__name__, str(e)[:300]), flush=True)
