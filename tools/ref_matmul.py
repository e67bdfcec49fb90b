#!/usr/bin/env python3
"""Reference point only: what the vendor BLAS (through torch.matmul, bf16) reaches on the layer shapes."""
import torch
shapes = [("layer4.conv1", 32400, 2048, 1024), ("layer4.conv3", 32400, 1024, 2048), ("layer3.conv1", 32400, 1024, 512),
          ("layer3.conv3", 32400, 512, 1024), ("aspp.pw", 32400, 2048, 256), ("layer1.conv3", 129600, 128, 256), ("square8k", 8192, 8192, 8192)]
for name, M, K, N in shapes:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16()
    for _ in range(3):
        c = a @ w.t()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        c = a @ w.t()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%-14s M=%6d K=%5d N=%5d %8.1f us %8.1f TF/s" % (name, M, K, N, ms * 1e3, 2.0 * M * K * N / ms / 1e9))
