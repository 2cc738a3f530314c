"""Small helpers with the reference's names and semantics (mpu/utils.py:19-75)."""
import torch


def ensure_divisibility(numerator, denominator):
    assert numerator % denominator == 0, '{} is not divisible by {}'.format(numerator, denominator)


def divide(numerator, denominator):
    ensure_divisibility(numerator, denominator)
    return numerator // denominator


def split_tensor_along_last_dim(tensor, num_partitions, contiguous_split_chunks=False):
    last = tensor.dim() - 1
    size = divide(tensor.size()[last], num_partitions)
    chunks = torch.split(tensor, size, dim=last)
    if contiguous_split_chunks:
        return tuple(c.contiguous() for c in chunks)
    return chunks


class VocabUtility:
    """Vocabulary range [first, last) owned by `rank` (mpu/utils.py:55-75)."""

    @staticmethod
    def vocab_range_from_per_partition_vocab_size(per_partition_vocab_size, rank, world_size):
        first = rank * per_partition_vocab_size
        return first, first + per_partition_vocab_size

    @staticmethod
    def vocab_range_from_global_vocab_size(global_vocab_size, rank, world_size):
        return VocabUtility.vocab_range_from_per_partition_vocab_size(
            divide(global_vocab_size, world_size), rank, world_size)
