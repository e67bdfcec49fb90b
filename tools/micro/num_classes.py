"""The classifier paths at other class counts than the reference's 19: <= 32 rides in the last refine block's epilogue (mixed) / in the classifier GEMM's
epilogue (other precisions); more than 32 classes: classifier GEMM + the stand-alone arg-max.  HIP logits / labels against the torch-CPU oracle."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from oracle import network_oracle as no
from vision_semantic_segmentation_amd.network import OP_ARGMAX, SegNet, random_state_dict
dev = torch.device("cuda:0")
h, w = 96, 128
img = np.random.default_rng(4).integers(0, 256, size=(h, w, 3), dtype=np.uint8)
for ncls in (5, 19, 32, 33, 40):
    state = random_state_dict(0, num_classes=ncls)
    ref = no.forward_logits(state, img)[0]
    for prec in ("f32", "mixed", "f16"):
        net = SegNet(state, h, w, precision=prec, device=dev, num_classes=ncls)
        net.forward(torch.from_numpy(img).to(dev))
        got = net.logits.permute(2, 0, 1).float().cpu()
        err = float((got - ref).abs().max() / ref.abs().max())
        lab = net.labels.cpu().numpy()
        print("%2d classes %-6s ops %2d (stand-alone arg-max: %s): logits %.2e of max|logit|, labels == argmax(logits): %s" % (
            ncls, prec, len(net.ops), any(op.kind == OP_ARGMAX for op in net.ops), err, bool(np.array_equal(lab, got.argmax(0).numpy()))), flush=True)
