"""Conv kernel micro-benchmark (GPU box): the last transposed convolution of the VQ-VAE decoder (512 -> 512, 128^2 -> 256^2)
and the second encoder convolution, batch MB_BATCH (default 32).  COGV_CONV_EXP selects the timing probes of conv.hip."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cogview_amd import _lib as L
from cogview_amd.vqvae.vqvae_zc import _conv, pack_conv_weight, pack_convt_weight
from tools.microbench import timeit
b = int(os.environ.get("MB_BATCH", "32"))
x = torch.randn(b, 128, 128, 512, device="cuda")
wt = pack_convt_weight(torch.randn(512, 512, 4, 4, device="cuda") * 0.02)
wc = pack_conv_weight(torch.randn(512, 512, 4, 4, device="cuda") * 0.02)
bias = torch.zeros(512, device="cuda")
row = {"exp": os.environ.get("COGV_CONV_EXP", "0"), "batch": b}
t = timeit(lambda: _conv(L.CONVT_4X4_S2, x, wt, bias, 512, True), iters=3, warm=1)
row["convT_TF"] = round(2.0 * b * 256 * 256 * 512 * 4 * 512 / t / 1e12, 1)
t = timeit(lambda: _conv(L.CONV_4X4_S2, x, wc, bias, 512, True), iters=3, warm=1)
row["conv_TF"] = round(2.0 * b * 64 * 64 * 512 * 16 * 512 / t / 1e12, 1)
print(json.dumps(row), flush=True)
