"""debug: one small k_dwpw_xs launch, error pattern per 16 x 16 block of the output"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "tests")))
from test_gpu_ops import _from_rows, _nhwc_rows, _run_plan
from vision_semantic_segmentation_amd import _lib
from vision_semantic_segmentation_amd.network import OP_DWPW, AvlSegOp, dwpw_tile_order, pack_dw_f32, pack_dw_pairs_split, pack_split_rows, split_f16
WS = int(os.environ.get('WS', '3'))
dev = torch.device("cuda:0")
H, W, K, N, d, pad = [int(v) for v in sys.argv[1:7]] if len(sys.argv) > 6 else (16, 16, 64, 256, 1, 1)
OH, OW = H + 2 * pad - 2 * d, W + 2 * pad - 2 * d
g = torch.Generator().manual_seed(3)
x = torch.randn((1, K, H, W), generator=g, dtype=torch.float64)
xh = x.to(torch.float16); xl = (x - xh.double()).to(torch.float16)
w1 = (torch.randn((K, 1, 3, 3), generator=g) * 0.3).double()
b1 = (torch.randn(K, generator=g) * 0.1).double()
w2 = torch.randn((N, K), generator=g, dtype=torch.float64) / K ** 0.5
b2 = torch.randn(N, generator=g) * 0.1
M = OH * OW
Mp, Np = (M + 255) // 256 * 256, (N + 255) // 256 * 256
src = torch.stack([_nhwc_rows(xh), _nhwc_rows(xl)]).to(dev)
w2p = torch.zeros((Np, K), dtype=torch.float64); w2p[:N] = w2
b2p = torch.zeros(Np); b2p[:N] = b2
w2d, b2d = pack_split_rows(w2p, 2).to(dev), b2p.to(dev)
out = torch.full((2, Mp, N), 7.0, dtype=torch.float16, device=dev)
params = torch.cat([pack_dw_f32(w1, b1) if WS == 3 else pack_dw_pairs_split(w1, b1), dwpw_tile_order(OH, OW, d)]).to(dev)
op = AvlSegOp()
op.kind, op.dtype = OP_DWPW, _lib.AVL_F16
op.in_, op.in_lo, op.in2, op.out, op.out_lo = src[0].data_ptr(), src[1].data_ptr(), params.data_ptr(), out[0].data_ptr(), out[1].data_ptr()
op.weight, op.bias, op.w_split = w2d.data_ptr(), b2d.data_ptr(), WS
op.in_h, op.in_w, op.in_c, op.in_ld, op.in_rows = H, W, K, K, src.shape[1]
op.out_h, op.out_w, op.out_c, op.out_ld, op.out_rows = OH, OW, N, N, Mp
op.relu, op.w_rows, op.ksize, op.stride, op.pad, op.dil, op.groups = 1, Np, 3, 1, pad, d, K
_run_plan([op])
w1h, w1l = split_f16(w1.reshape(K, 9))
w1s = (w1h.double() + w1l.double()).reshape(K, 1, 3, 3) if WS == 2 else w1.float().double()
a64 = F.relu(F.conv2d(xh.double() + xl.double(), w1s, b1.float().double(), padding=pad, dilation=d, groups=K))
ref = F.relu(F.conv2d(a64, w2.view(N, K, 1, 1), b2.double()))
got = _from_rows(out[0].cpu().double() + out[1].cpu().double(), OH, OW, N)
e = (got - ref).abs()[0].reshape(N, M).t()          # [M, N]
print("max rel err %.3e" % float(e.max() / ref.abs().max()))
blk = e[:M // 16 * 16, :N // 16 * 16].reshape(M // 16, 16, N // 16, 16).amax(dim=(1, 3)) / float(ref.abs().max())
torch.set_printoptions(linewidth=250, precision=1, sci_mode=False)
print((blk[:16] > 1e-3).int())
print("got[0,:8]", got[0, :8, 0, 0], "ref", ref[0, :8, 0, 0])
G = got[0].reshape(N, M).t(); R = ref[0].reshape(N, M).t()
for r in (47, 48, 49, 63, 64, 112):
    print("row", r, "got", G[r, :6].numpy().round(3), "ref", R[r, :6].numpy().round(3))
# which A rows would explain it: solve nothing, just ratio
print("ratio rows 48..52 col 0..3", (G[48:53, :4] / R[48:53, :4].clamp_min(1e-9)).numpy().round(3))
bad = (e / float(ref.abs().max()) > 1e-3).nonzero()
print("bad elements", bad.shape[0], "rows", sorted(set(bad[:, 0].tolist()))[:40], "cols", sorted(set(bad[:, 1].tolist()))[:40])
print("xy of bad rows", [(r // OW, r % OW) for r in sorted(set(bad[:, 0].tolist()))[:20]])
