"""GPU parity of the VQ-VAE tokenizer path (img2code / code2img) against the reference's own outputs
(tests/golden/vqvae_small.npz) and the CPU oracle at the production size.

Ids are an argmin over fp32 distances; the HIP path keeps fp32 products and fp32 accumulation (exact-fp32 MFMA).
BASELINE.json's bar for the token ids is BIT-EXACT, and that is what is asserted (`torch.equal`), on the reference's
golden ids, at the production size and inside a batch of 256 (cfg 5).  The audit (_audit_ids) only words the failure:
should an id ever differ, it reports whether the two best codes were within rounding distance of each other (a different
summation order) or not (a bug).  Decoded images: relative L2 <= 1e-5 (fp32).
"""
import os

import numpy as np
import pytest
import torch

from oracle import cogview_oracle as O

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _audit_ids(ids_hip, ids_ref, dist, name):
    ids_hip, ids_ref = ids_hip.cpu().reshape(-1), ids_ref.reshape(-1)
    match = (ids_hip == ids_ref).float().mean().item()
    bad = (ids_hip != ids_ref).nonzero().reshape(-1)
    worst = 0.0
    for i in bad.tolist():
        d = dist[i]
        gap = (d[ids_hip[i]] - d.min()).abs().item() / max(d.min().abs().item(), 1e-12)
        worst = max(worst, gap)
    msg = f"[{name}] id exact-match {match * 100:.3f}% ({len(bad)} of {ids_ref.numel()} differ), worst top-2 gap {worst:.2e}"
    print(msg)
    assert torch.equal(ids_hip, ids_ref), msg + (" -- near-ties (summation order)" if worst < 1e-4 else " -- NOT near-ties")


def test_small_vs_reference_golden(golden_dir):
    from cogview_amd.vqvae.vqvae_zc import VQVAE
    z = np.load(os.path.join(golden_dir, "vqvae_small.npz"))
    g = {k: torch.from_numpy(z[k]) for k in z.files}
    m = VQVAE(channel=32, n_res_block=0, n_res_channel=32, embed_dim=16, n_embed=64, stride=6)
    m.load_state_dict({k[6:]: v for k, v in g.items() if k.startswith("param.")})
    m = m.cuda().eval()
    p = {k[6:]: v for k, v in g.items() if k.startswith("param.")}
    _, _, dist = O.vqvae_encode(g["img"], p)
    with torch.no_grad():
        quant, diff, ids = m.encode(g["img"].cuda())
        assert quant.shape == (2, 16, 8, 8) and diff.shape == (1,) and ids.shape == (2, 8, 8)
        _audit_ids(ids, g["ids"], dist, "small/golden")
        dec = m.decode_code(g["ids"].cuda())
    assert rel(dec, g["dec"]) < 1e-5
    from cogview_amd.vqvae import code2img
    assert rel(code2img(m, g["ids"].cuda()), g["dec_denorm"]) < 1e-5


def test_production_size_vs_oracle():
    """vqvae.new_model(): channel 512, embed 256, 8192 codes, 256x256 images -> 32x32 codes (BASELINE configs[4])."""
    from cogview_amd import vqvae
    torch.manual_seed(0)
    m = vqvae.new_model().eval()
    p = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(0)
    img = torch.randn(2, 3, 256, 256, generator=g)
    with torch.no_grad():
        ids_ref, _, dist = O.vqvae_encode(img, p)
        dec_ref = O.code2img_denorm(O.vqvae_decode(ids_ref, p))
    m = m.cuda()
    ids = vqvae.img2code(m, img.cuda())
    assert ids.shape == (2, 1024) and ids.dtype == torch.int64
    _audit_ids(ids, ids_ref, dist, "production")
    out = vqvae.code2img(m, ids_ref.cuda())
    assert out.shape == (2, 3, 256, 256)
    assert rel(out, dec_ref) < 1e-5
    # flat codes are accepted for batch 1 only, like the reference (vqvae/api.py:38-40)
    one = vqvae.code2img(m, ids_ref[:1].reshape(1, -1).cuda())
    assert torch.equal(one, out[:1])
    # determinism + batch independence at a larger batch
    big = torch.randn(8, 3, 256, 256, generator=g).cuda()
    a, b = vqvae.img2code(m, big), vqvae.img2code(m, big)
    assert torch.equal(a, b)
    assert torch.equal(vqvae.img2code(m, big[3:5]), a[3:5])


def test_batch_256_ids_and_images_vs_oracle():
    """BASELINE configs[4] at its own batch: img2code of 256 images of 256 x 256 (seed 0, as bench.py --config vqvae) --
    the ids of a 16-image subset must equal the CPU oracle's bit for bit and must not depend on the batch they were
    encoded in; code2img of all 256 id maps, a 4-image subset against the oracle's decoder."""
    from cogview_amd import vqvae
    torch.manual_seed(0)
    m = vqvae.new_model().eval()
    p = {k: v.detach().clone() for k, v in m.state_dict().items()}
    img = torch.randn(256, 3, 256, 256, generator=torch.Generator().manual_seed(0))
    sub = list(range(0, 256, 17))[:16]
    with torch.no_grad():
        ids_ref, _, dist = O.vqvae_encode(img[sub], p)
        dec_ref = O.code2img_denorm(O.vqvae_decode(ids_ref[:4], p))
    m = m.cuda()
    ids = vqvae.img2code(m, img.cuda())
    assert ids.shape == (256, 1024)
    _audit_ids(ids[sub], ids_ref, dist, "batch 256, 16-image subset")
    assert torch.equal(vqvae.img2code(m, img[sub].cuda()), ids[sub]), "ids depend on the batch"
    out = vqvae.code2img(m, ids.view(256, 32, 32))
    assert out.shape == (256, 3, 256, 256) and bool(torch.isfinite(out).all())
    assert rel(out[sub[:4]], dec_ref) < 1e-5


def test_images_to_compact_binary_pipeline(tmp_path):
    """The img2code stage of the preprocessing (preprocess/preprocess_text_image_data.py:28-64) writing the
    CompactBinaryDataset file: batched encoding, row layout, and what the dataset reader then hands the trainer."""
    from types import SimpleNamespace
    from cogview_amd import vqvae
    from cogview_amd.data_utils import get_dataset_by_type, images_to_compact_binary
    from cogview_amd.generation import IdSpace
    torch.manual_seed(5)
    model = vqvae.new_model().cuda().eval()
    imgs = torch.rand(3, 3, 256, 256, device="cuda") * 2 - 1
    texts = [[8192 + 5, 8192 + 77], [8192 + 9], list(range(9000, 9010))]
    path = str(tmp_path / "train.bin")
    assert images_to_compact_binary(model, imgs, texts, path, batch_size=2) == 3
    rows = np.fromfile(path, dtype=np.int32).reshape(-1, 64 + 1024)
    codes = vqvae.img2code(model, imgs).reshape(3, -1).cpu().numpy()
    assert rows.shape[0] == 3 and np.array_equal(rows[:, 64:], codes) and codes.min() >= 0 and codes.max() < 8192
    assert rows[0, :2].tolist() == texts[0] and (rows[0, 2:64] == -1).all()
    ids = IdSpace()
    s = get_dataset_by_type("CompactBinaryDataset", path, SimpleNamespace(max_position_embeddings=1089))[1]
    assert s["text"][:4].tolist() == [ids['[ROI1]'], 8192 + 9, ids['[BASE]'], ids['[BOI1]']] and s["loss_mask"].sum() == 1 + 1 + 2 + 1024 + 1


@pytest.mark.parametrize("kind,B,H,W,Cin,Cout", [
    ("1x1", 2, 8, 8, 32, 16), ("1x1", 2, 8, 8, 64, 16), ("1x1", 4, 16, 16, 32, 32), ("1x1", 1, 4, 4, 16, 8),
    ("1x1", 2, 8, 8, 96, 24), ("conv", 2, 16, 16, 4, 32), ("conv", 2, 16, 16, 8, 16), ("conv", 1, 8, 8, 32, 32),
    ("convT", 2, 8, 8, 16, 32), ("convT", 1, 4, 4, 8, 8), ("convT", 2, 8, 8, 32, 16)])
def test_conv_kernel_short_contractions(kind, B, H, W, Cin, Cout):
    """The implicit-GEMM kernel at contraction lengths of one, two and three 32-deep k-tiles (and k-tiles that straddle
    taps): the software pipeline keeps two tiles of loads in flight, so its prologue / drain paths are exactly what a
    1 x 1 convolution with 32 input channels runs.  Reference: the oracle's definition (F.conv2d / F.conv_transpose2d
    on the CPU, vqvae/vqvae_zc.py:121-129,172-192)."""
    import torch.nn.functional as F
    from cogview_amd import _lib as L
    from cogview_amd.vqvae.vqvae_zc import _conv, pack_conv_weight, pack_convt_weight
    g = torch.Generator().manual_seed(B * 1000 + Cin * 10 + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    bias = torch.randn(Cout, generator=g)
    xh = x.permute(0, 2, 3, 1).contiguous().cuda()
    if kind == "1x1":
        w = torch.randn(Cout, Cin, 1, 1, generator=g) * 0.1
        ref = F.conv2d(x, w, bias)
        y = _conv(L.CONV_1X1, xh, pack_conv_weight(w.cuda()), bias.cuda(), Cout, False)
    elif kind == "conv":
        w = torch.randn(Cout, Cin, 4, 4, generator=g) * 0.1
        ref = F.relu(F.conv2d(x, w, bias, stride=2, padding=1))
        y = _conv(L.CONV_4X4_S2, xh, pack_conv_weight(w.cuda()), bias.cuda(), Cout, True)
    else:
        w = torch.randn(Cin, Cout, 4, 4, generator=g) * 0.1
        ref = F.relu(F.conv_transpose2d(x, w, bias, stride=2, padding=1))
        y = _conv(L.CONVT_4X4_S2, xh, pack_convt_weight(w.cuda()), bias.cuda(), Cout, True)
    for _ in range(2):                       # twice: timing-dependent races show up as run-to-run differences
        assert rel(y.permute(0, 3, 1, 2), ref) < 1e-5


def test_fused_rgb_projection_of_the_last_transposed_conv():
    """COGV conv descriptor with rgb_w / rgb_partial: the last transposed convolution's (bias + ReLU) output projected to
    RGB in its epilogue (per-channel-tile partial sums + cogv_rgb_finalize_f32) == conv1x1(relu(convT(x))) of the oracle
    (vqvae/vqvae_zc.py:186-190) with the de-normalisation of vqvae/api.py:43; the 512-channel activation is never written."""
    import torch.nn.functional as F
    from cogview_amd import _lib as L
    from cogview_amd.vqvae.vqvae_zc import _conv, pack_convt_weight
    g = torch.Generator().manual_seed(5)
    B, H, W, Cin, Cout = 2, 8, 8, 64, 256
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cin, Cout, 4, 4, generator=g) * 0.05
    bias = torch.randn(Cout, generator=g) * 0.1
    w4 = torch.randn(3, Cout, 1, 1, generator=g) * 0.1
    b4 = torch.randn(3, generator=g)
    scale, shift = (0.30379, 0.32279, 0.32800), (0.79093, 0.76271, 0.75340)
    ref = F.conv2d(F.relu(F.conv_transpose2d(x, w, bias, stride=2, padding=1)), w4, b4)
    ref = ref * torch.tensor(scale).view(1, 3, 1, 1) + torch.tensor(shift).view(1, 3, 1, 1)
    img = _conv(L.CONVT_4X4_S2, x.permute(0, 2, 3, 1).contiguous().cuda(), pack_convt_weight(w.cuda()), bias.cuda(), Cout, True,
                rgb=(w4.reshape(3, Cout).contiguous().cuda(), b4.cuda(), scale, shift))
    assert img.shape == (B, 3, 2 * H, 2 * W)
    assert rel(img, ref) < 1e-5


@pytest.mark.parametrize("variant", ["s4_simple_res2", "s4_pyramid_res1", "s6_pyramid_res2", "s2_res1", "s6_simple_res1"])
def test_non_production_topologies_vs_the_reference(golden_dir, variant):
    """Every topology vqvae/vqvae_zc.py's constructors can build besides the production one (round-3 verdict, missing item 4):
    stride 4 / 2 encoders with their 3x3 convolution, the non-simple channel pyramid (channel / 4, / 2; a 1x1 TRANSPOSED
    convolution and a 4x4 transposed convolution straight to RGB in the decoder), ResBlocks -- whose leading in-place ReLU
    makes the block add relu(input), not input -- against the REFERENCE's own outputs (oracle/gen_golden_vqvae_variants.py):
    token ids bit-exact, decoded image within 2e-5 relative L2 (fp32 products and sums, different summation order)."""
    from cogview_amd.vqvae.vqvae_zc import VQVAE
    z = np.load(os.path.join(golden_dir, "vqvae_variants.npz"))
    ch, nrb, nrc, ed, ne, stride, simple = [int(v) for v in z[f"{variant}.kw"]]
    m = VQVAE(channel=ch, n_res_block=nrb, n_res_channel=nrc, embed_dim=ed, n_embed=ne, stride=stride, simple=bool(simple))
    pre = f"{variant}.param."
    m.load_state_dict({k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)})
    m = m.cuda().eval()
    img = torch.from_numpy(z[f"{variant}.img"]).cuda()
    ids_ref = torch.from_numpy(z[f"{variant}.ids"])
    with torch.no_grad():
        quant, diff, ids = m.encode(img)
        assert torch.equal(ids.cpu(), ids_ref), f"{(ids.cpu() != ids_ref).sum().item()} of {ids_ref.numel()} ids differ"
        dec = m.decode_code(ids)
    ref = torch.from_numpy(z[f"{variant}.dec"])
    assert dec.shape == ref.shape
    e = ((dec.cpu().double() - ref.double()).norm() / ref.double().norm()).item()
    assert e < 2e-5, e
    assert quant.shape[1] == ed and tuple(quant.shape[2:]) == tuple(ids.shape[1:])
