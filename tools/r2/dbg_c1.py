import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from cogview_amd.vqvae.vqvae_zc import pack_conv_weight, _conv
from cogview_amd import _lib as L
torch.manual_seed(0)
for (B, H, W, Cin, Cout) in ((2, 8, 8, 32, 16), (2, 8, 8, 64, 16), (4, 16, 16, 32, 32), (2, 8, 8, 512, 256), (1, 4, 4, 16, 8)):
    x = torch.randn(B, Cin, H, W)
    w = torch.randn(Cout, Cin, 1, 1) * 0.1
    b = torch.randn(Cout)
    ref = F.conv2d(x, w, b)
    y = _conv(L.CONV_1X1, x.permute(0, 2, 3, 1).contiguous().cuda(), pack_conv_weight(w.cuda()), b.cuda(), Cout, False)
    o = y.permute(0, 3, 1, 2).cpu()
    err = (o - ref).abs()
    print((B, H, W, Cin, Cout), "rel", ((o - ref).norm() / ref.norm()).item(), "per-channel max err", [round(v, 3) for v in err.amax(dim=(0, 2, 3)).tolist()][:16])
