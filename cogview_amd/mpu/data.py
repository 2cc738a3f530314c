"""broadcast_data: model-parallel rank 0 loads the batch and broadcasts it inside its group
(mpu/data.py:33-116).  Two broadcasts: the packed sizes, then one flat payload."""
import torch

from .initialize import get_model_parallel_group, get_model_parallel_rank, get_model_parallel_src_rank

_MAX_DATA_DIM = 5


def _device():
    return torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')


def _check_data_types(keys, data, target_dtype):
    for key in keys:
        assert data[key].dtype == target_dtype, '{} has data type {} which is different than {}'.format(
            key, data[key].dtype, target_dtype)


def _build_key_size_numel_dictionaries(keys, data):
    sizes = [0] * (_MAX_DATA_DIM * len(keys))
    if get_model_parallel_rank() == 0:
        for k, key in enumerate(keys):
            assert data[key].dim() < _MAX_DATA_DIM, 'you should increase MAX_DATA_DIM'
            for i, s in enumerate(data[key].size()):
                sizes[k * _MAX_DATA_DIM + i] = s
    sizes_dev = torch.tensor(sizes, dtype=torch.int64, device=_device())
    torch.distributed.broadcast(sizes_dev, get_model_parallel_src_rank(), group=get_model_parallel_group())
    sizes_cpu = sizes_dev.cpu().tolist()
    key_size, key_numel, total = {}, {}, 0
    for k, key in enumerate(keys):
        size, numel, i = [], 1, 0
        while i < _MAX_DATA_DIM and sizes_cpu[k * _MAX_DATA_DIM + i] > 0:
            size.append(sizes_cpu[k * _MAX_DATA_DIM + i])
            numel *= size[-1]
            i += 1
        key_size[key], key_numel[key] = size, numel
        total += numel
    return key_size, key_numel, total


def broadcast_data(keys, data, datatype):
    key_size, key_numel, total = _build_key_size_numel_dictionaries(keys, data)
    if get_model_parallel_rank() == 0:
        _check_data_types(keys, data, datatype)
        flat = torch.cat([data[key].contiguous().view(-1) for key in keys], dim=0).to(_device())
    else:
        flat = torch.empty(total, device=_device(), dtype=datatype)
    torch.distributed.broadcast(flat, get_model_parallel_src_rank(), group=get_model_parallel_group())
    out, offset = {}, 0
    for key in keys:
        out[key] = flat.narrow(0, offset, key_numel[key]).view(key_size[key])
        offset += key_numel[key]
    return out
