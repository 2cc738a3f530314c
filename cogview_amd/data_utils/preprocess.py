"""The img2code stage of the reference's preprocessing (preprocess/preprocess_text_image_data.py:28-64: images in batches
of 128 through the VQ-VAE encoder, one row of text ids + 1024 codes per image), writing the CompactBinaryDataset file the
trainer reads instead of LMDB pickles (no lmdb module in this image).  The encoder is the HIP VQ-VAE path."""
import numpy as np
import torch

from .datasets import write_compact_binary


@torch.no_grad()
def images_to_compact_binary(model, images, text_ids, path, batch_size=128, img2code=None):
    """images: tensor [n, 3, 256, 256] (normalised as vqvae/api.py expects) or an iterable of such batches;
    text_ids: n id lists.  Returns the number of rows written."""
    if img2code is None:
        from ..vqvae import img2code as _img2code
        img2code = _img2code
    if isinstance(images, torch.Tensor):
        images = [images[i:i + batch_size] for i in range(0, images.shape[0], batch_size)]
    done = 0
    for chunk in images:
        codes = img2code(model, chunk)
        codes = codes.reshape(codes.shape[0], -1).cpu().numpy()
        write_compact_binary(path, text_ids[done:done + codes.shape[0]], codes, append=done > 0)
        done += codes.shape[0]
    assert done == len(text_ids)
    return done
