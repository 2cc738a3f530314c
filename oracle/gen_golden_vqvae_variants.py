"""Generate tests/golden/vqvae_variants.npz by running the REFERENCE's VQVAE (vqvae/vqvae_zc.py, imports unmodified) on the
topologies its constructors can build besides the production one: stride 4 / 2, the non-simple channel pyramid, ResBlocks.
Run in the build container only:   python oracle/gen_golden_vqvae_variants.py
Per variant: constructor arguments, state dict, a random image batch, the token ids and the decoded image of those ids."""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
VARIANTS = {
    "s4_simple_res2": dict(channel=32, n_res_block=2, n_res_channel=16, embed_dim=16, n_embed=64, stride=4, simple=True),
    "s4_pyramid_res1": dict(channel=32, n_res_block=1, n_res_channel=8, embed_dim=16, n_embed=64, stride=4, simple=False),
    "s6_pyramid_res2": dict(channel=32, n_res_block=2, n_res_channel=16, embed_dim=16, n_embed=64, stride=6, simple=False),
    "s2_res1": dict(channel=16, n_res_block=1, n_res_channel=8, embed_dim=8, n_embed=32, stride=2, simple=True),
    "s6_simple_res1": dict(channel=32, n_res_block=1, n_res_channel=16, embed_dim=16, n_embed=64, stride=6, simple=True),
}


def main():
    sys.path.insert(0, REF)
    from vqvae.vqvae_zc import VQVAE
    arrs = {}
    for i, (name, kw) in enumerate(VARIANTS.items()):
        torch.manual_seed(100 + i)
        m = VQVAE(**kw).eval()
        with torch.no_grad():
            for p in m.parameters():                     # biases start at zero-ish scale: make them matter
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn_like(p))
        img = torch.randn(2, 3, 48, 48, generator=torch.Generator().manual_seed(7 + i))
        with torch.no_grad():
            _, _, ids = m.encode(img.clone())
            dec = m.decode_code(ids)
        for k, v in m.state_dict().items():
            arrs[f"{name}.param.{k}"] = v.numpy()
        arrs[f"{name}.img"], arrs[f"{name}.ids"], arrs[f"{name}.dec"] = img.numpy(), ids.numpy(), dec.numpy()
        arrs[f"{name}.kw"] = np.array([kw["channel"], kw["n_res_block"], kw["n_res_channel"], kw["embed_dim"], kw["n_embed"],
                                       kw["stride"], int(kw["simple"])])
        print(name, tuple(ids.shape), tuple(dec.shape))
    path = os.path.join(OUT, "vqvae_variants.npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
