"""Production tokenizer API (vqvae/api.py:12-44): new_model / img2code / code2img."""
import math

import torch

from .vqvae_zc import VQVAE

_STD = (0.30379, 0.32279, 0.32800)
_MEAN = (0.79093, 0.76271, 0.75340)


def new_model():
    """The pretrained tokenizer's architecture (vqvae/api.py:12-20)."""
    return VQVAE(channel=512, n_res_block=0, n_res_channel=32, embed_dim=256, n_embed=8192, stride=6)


def img2code(model, img):
    """[b, 3, H, W] normalised image -> LongTensor [b, (H/8)*(W/8)]."""
    with torch.no_grad():
        ids = model.encode_ids(img)
    return ids.view(img.shape[0], -1)


def code2img(model, code):
    """[b, h, w] codes (or [1, h*w]: the reference takes the square root of the TOTAL element count, so flat
    codes are only valid for batch 1 -- vqvae/api.py:38-40) -> de-normalised image [b, 3, 8h, 8w]; the
    per-channel `* std + mean` of api.py:43 is fused into the last kernel."""
    if len(code.shape) == 2:
        s = int(math.sqrt(len(code.view(-1))) + 1e-5)
        code = code.view(code.shape[0], s, s)
    with torch.no_grad():
        return model.decode_code(code, scale=_STD, shift=_MEAN)
