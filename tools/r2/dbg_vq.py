import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F
from cogview_amd.vqvae.vqvae_zc import VQVAE, pack_conv_weight, pack_convt_weight, _conv
from cogview_amd import _lib as L
z = np.load("tests/golden/vqvae_small.npz")
g = {k: torch.from_numpy(z[k]) for k in z.files}
p = {k[6:]: v for k, v in g.items() if k.startswith("param.")}
from cogview_amd import ops
for trial in range(4):
    if trial == 3:
        a = torch.randn(4096, 2560, device="cuda", dtype=torch.bfloat16); wgt = torch.randn(2560, 2560, device="cuda", dtype=torch.bfloat16)
        for _ in range(3): ops.gemm(a, wgt)
        torch.cuda.synchronize(); print("ran GEMMs")
    if trial:
        junk = [torch.full((1 << 28,), float("nan"), device="cuda") for _ in range(2)]
        for sz in (64, 256, 1024, 4096, 16384, 65536, 200000):
            junk += [torch.full((sz,), 3.0e4 if trial == 1 else float("nan"), device="cuda") for _ in range(100)]
        del junk
    m = VQVAE(channel=32, n_res_block=0, n_res_channel=32, embed_dim=16, n_embed=64, stride=6)
    m.load_state_dict(p); m = m.cuda().eval()
    img = g["img"]
    x = img
    refs = []
    x = F.relu(F.conv2d(x, p["enc_b.blocks.0.weight"], p["enc_b.blocks.0.bias"], stride=2, padding=1)); refs.append(x)
    x = F.relu(F.conv2d(x, p["enc_b.blocks.2.weight"], p["enc_b.blocks.2.bias"], stride=2, padding=1)); refs.append(x)
    x = F.relu(F.conv2d(x, p["enc_b.blocks.4.weight"], p["enc_b.blocks.4.bias"], stride=2, padding=1)); refs.append(x)
    x = F.conv2d(x, p["enc_b.blocks.6.weight"], p["enc_b.blocks.6.bias"]); refs.append(x)
    import ctypes as C
    enc = m.enc_b
    (w1, b1), (w2, b2), (w3, b3), (w4, b4) = enc._weights([pack_conv_weight] * 4)
    xi = img.cuda().contiguous().float()
    b, c, h, w = xi.shape
    x4 = torch.empty((b, h, w, 4), dtype=torch.float32, device="cuda")
    L.check(L.lib().cogv_nchw3_to_nhwc4_f32(C.c_void_p(xi.data_ptr()), C.c_void_p(x4.data_ptr()), b, h, w, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "x")
    outs = []
    y = _conv(L.CONV_4X4_S2, x4, w1, b1, 32, True); outs.append(y)
    y = _conv(L.CONV_4X4_S2, y, w2, b2, 32, True); outs.append(y)
    y = _conv(L.CONV_4X4_S2, y, w3, b3, 32, True); outs.append(y)
    y = _conv(L.CONV_1X1, y, w4, b4, 16, False); outs.append(y)
    for i, (o, r) in enumerate(zip(outs, refs)):
        o2 = o.permute(0, 3, 1, 2).cpu()
        e = ((o2 - r).norm() / r.norm()).item()
        print(f"trial {trial} stage {i} shape {tuple(o.shape)} rel err {e:.3e} nan {bool(torch.isnan(o2).any())}")
    ids = m.quantize_t.nearest_code(outs[-1])
    print("trial", trial, "ids match", (ids.cpu().reshape(-1) == g["ids"].reshape(-1)).float().mean().item())
    dec = m.decode_code(g["ids"].cuda())
    print("trial", trial, "dec rel", ((dec.cpu() - g["dec"]).norm() / g["dec"].norm()).item())
