"""Token datasets in the reference's on-disk formats.

CompactBinaryDataset (data_utils/datasets.py:63-81,119-128): a flat int32 file of rows [64 text ids padded with -1 |
1024 image codes]; a sample is  [ROI1] text  [BASE] [BOI1] codes [EOI1]  padded with [PAD] to max_position_embeddings,
with a loss mask over the unpadded part.  The LMDB-backed variants need the `lmdb` module, which this image does not
have; their row processing (TokenizedDataset) is exposed for any sequence-like backing store.
"""
import random

import numpy as np
from torch.utils.data import Dataset

from ..generation.id_space import IdSpace


class BinaryDataset(Dataset):
    """Fixed-length rows of a flat binary file, memory-mapped (or read at once with preload=True)."""

    def __init__(self, path, process_fn, length_per_sample=64 + 1024, dtype='int32', preload=False, **kwargs):
        assert length_per_sample is not None
        self.length_per_sample, self.dtype, self.process_fn = length_per_sample, np.dtype(dtype), process_fn
        if preload:
            self.bin = np.fromfile(path, dtype=self.dtype).reshape(-1, length_per_sample)
        else:
            import os
            n = os.path.getsize(path) // self.dtype.itemsize
            self.bin = np.memmap(path, dtype=self.dtype, mode='r', shape=(n // length_per_sample, length_per_sample))

    def __len__(self):
        return self.bin.shape[0]

    def __getitem__(self, index):
        return self.process_fn(self.bin[index])


class RandomMappingDataset(Dataset):
    """data_utils/configure_data.py:276-291: 200x virtual length; index i maps to a sample drawn by an rng seeded with i
    (so every epoch-less pass is a deterministic shuffle with replacement)."""

    def __init__(self, ds, **kwargs):
        self.wrapped_data = ds

    def __len__(self):
        return len(self.wrapped_data) * 200

    def __getitem__(self, index):
        rng = random.Random(index)
        rng = np.random.RandomState(seed=[rng.randint(0, 2 ** 32 - 1) for _ in range(16)])
        return self.wrapped_data[rng.randint(len(self.wrapped_data))]


def TextCodeTemplate(text, code, tokenizer=None):
    """data_utils/templates.py:53-66 for id arrays: [ROI1] + text ids + wrapped image code."""
    tokenizer = tokenizer if tokenizer is not None else IdSpace()
    text_ids = np.concatenate((np.array([tokenizer['[ROI1]']]), np.asarray(text)), axis=0)
    return np.concatenate((text_ids, tokenizer.wrap_code(np.asarray(code))), axis=0)


def get_dataset_by_type(dataset_type, path, args, tokenizer=None, **kwargs):
    """data_utils/datasets.py:83-128.  args: .max_position_embeddings (and .finetune /
    .max_position_embeddings_finetune).  'TokenizedDataset' takes any indexable `path` object holding id arrays."""
    tokenizer = tokenizer if tokenizer is not None else IdSpace()
    ml = args.max_position_embeddings
    if getattr(args, 'finetune', False) and getattr(args, 'max_position_embeddings_finetune', 0) > ml:
        ml = args.max_position_embeddings_finetune
    pad = tokenizer['[PAD]']

    def pad_to_len(ret):
        n = len(ret)
        if n < ml:
            return np.concatenate((ret, np.full(ml - n, pad, dtype=np.asarray(ret).dtype)), axis=0), n
        return ret[:ml], ml

    def with_mask(ret):
        ret, n = pad_to_len(ret)
        return {'text': ret, 'loss_mask': np.array([1] * n + [0] * (len(ret) - n))}

    if dataset_type == 'CompactBinaryDataset':
        def process_fn(row):
            text, code = row[:64].astype(np.int64), row[64:].astype(np.int64)        # must be 64 + 1024
            return with_mask(TextCodeTemplate(text[text > -1], code, tokenizer))
        return BinaryDataset(path, process_fn, **kwargs)
    if dataset_type == 'TokenizedDataset':
        class _Rows(Dataset):
            def __len__(self):
                return len(path)

            def __getitem__(self, i):
                return with_mask(np.asarray(path[i]).flatten())
        return _Rows()
    raise NotImplementedError(f"{dataset_type}: the LMDB-backed datasets need the lmdb module (not in this image)")


def write_compact_binary(path, text_ids, codes, append=False):
    """Write rows in the CompactBinaryDataset layout: text_ids = sequence of id lists (<= 64 ids each, padded with -1),
    codes [n, 1024] image codes."""
    codes = np.asarray(codes)
    assert codes.ndim == 2 and codes.shape[1] == 1024 and len(text_ids) == codes.shape[0]
    rows = np.full((codes.shape[0], 64 + 1024), -1, dtype=np.int32)
    for i, t in enumerate(text_ids):
        t = np.asarray(t, dtype=np.int32)[:64]
        rows[i, :len(t)] = t
    rows[:, 64:] = codes.astype(np.int32)
    with open(path, 'ab' if append else 'wb') as f:
        rows.tofile(f)
    return rows.shape[0]
