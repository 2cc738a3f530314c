"""Checkpoint files in the reference's format (utils.py:158-380).

    <dir>/latest_checkpointed_iteration.txt            "<iteration>" or "release"
    <dir>/<iteration | release>/mp_rank_XX_model_states.pt
        {'iteration', 'module' (model.state_dict()), 'optimizer', 'lr_scheduler',
         'random_rng_state', 'np_rng_state', 'torch_rng_state', 'cuda_rng_state', 'rng_tracker_states'}

The file name and the 'module' key are also what DeepSpeed writes (`mp_rank_00_model_states.pt`), so the released
cogview-base weights -- a DeepSpeed checkpoint, generate_samples.py:56-61 -- load through load_checkpoint unchanged:
cogview_amd.model.GPT2Model keeps the reference's parameter names and layouts (fused QKV rows [q; k; v] per partition,
mpu/layers.py:64-71).  Host-side logic only; the DeepSpeed *engine* paths (args.deepspeed) are not reproduced.
"""
import contextlib
import os
import random
import sys

import numpy as np
import torch

from . import mpu


def _dp_rank():
    return mpu.get_data_parallel_rank() if mpu.model_parallel_is_initialized() else 0


def _mp_rank():
    return mpu.get_model_parallel_rank() if mpu.model_parallel_is_initialized() else 0


def _barrier():
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()


def _global_rank():
    return torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0


def print_rank_0(message):
    if _global_rank() == 0:
        print(message, flush=True)


def get_checkpoint_name(checkpoints_path, iteration, release=False, zero=False):
    d = 'release' if release else '{:d}'.format(iteration)
    if zero:
        d += '_zero_dp_rank_{}'.format(_dp_rank())
    return os.path.join(checkpoints_path, d, 'mp_rank_{:02d}_model_states.pt'.format(_mp_rank()))


def ensure_directory_exists(filename):
    os.makedirs(os.path.dirname(filename), exist_ok=True)


def get_checkpoint_tracker_filename(checkpoints_path):
    return os.path.join(checkpoints_path, 'latest_checkpointed_iteration.txt')


def _unwrap(model):
    from .model.distributed import DistributedDataParallel
    while isinstance(model, DistributedDataParallel):
        model = model.module
    return model


def save_checkpoint(iteration, model, optimizer, lr_scheduler, args):
    """utils.py:188-234 (the non-DeepSpeed branch): data-parallel rank 0 of every model-parallel rank writes its file,
    global rank 0 then moves the tracker."""
    if getattr(args, 'deepspeed', False):
        raise NotImplementedError("the DeepSpeed engine is not reproduced; checkpoints are written in the same layout")
    # sharded exchange: step() all-gathers the updated parameters on a side stream and each layer of the NEXT forward
    # waits for its own region -- nothing orders that against a state_dict() taken through the unwrapped module, so
    # wait for the whole arena here (no-op for the all-reduce exchange)
    wrapped = model
    while wrapped is not None:
        if getattr(wrapped, 'shard', None) is not None:
            wrapped.shard.wait_upto(wrapped.arena.total)
        wrapped = getattr(wrapped, 'module', None)
    model = _unwrap(model)
    if optimizer is not None and not getattr(args, 'no_save_optim', False) and hasattr(optimizer, 'consolidate_state'):
        optimizer.consolidate_state()        # collective: sharded optimizer state -> full state on every rank
    if _dp_rank() == 0:
        name = get_checkpoint_name(args.save, iteration)
        print('global rank {} is saving checkpoint at iteration {:7d} to {}'.format(_global_rank(), iteration, name))
        sd = {'iteration': iteration, 'module': model.state_dict()}
        if not getattr(args, 'no_save_optim', False):
            if optimizer is not None:
                sd['optimizer'] = optimizer.state_dict()
            if lr_scheduler is not None:
                sd['lr_scheduler'] = lr_scheduler.state_dict()
        if not getattr(args, 'no_save_rng', False):
            sd['random_rng_state'] = random.getstate()
            sd['np_rng_state'] = np.random.get_state()
            sd['torch_rng_state'] = torch.get_rng_state()
            if torch.cuda.is_available():
                sd['cuda_rng_state'] = torch.cuda.get_rng_state()
            sd['rng_tracker_states'] = mpu.get_cuda_rng_tracker().get_states()
            # the counter-based dropout generator's default (seed, offset) pair: what torch.cuda's generator state is
            # to the reference's hidden / embedding dropout (not covered by 'cuda_rng_state' here)
            st = mpu.random.get_default_state()
            sd['cogv_default_dropout_state'] = (st.seed, st.offset)
        ensure_directory_exists(name)
        torch.save(sd, name)
        print('  successfully saved {}'.format(name))
    _barrier()
    if _global_rank() == 0:
        with open(get_checkpoint_tracker_filename(args.save), 'w') as f:
            f.write(str(iteration))
    _barrier()


def get_checkpoint_iteration(args):
    """utils.py:255-281: (iteration, release, success) from the tracker file."""
    tracker = get_checkpoint_tracker_filename(args.load)
    if not os.path.isfile(tracker):
        print_rank_0('WARNING: could not find the metadata file {} '.format(tracker))
        print_rank_0('    will not load any checkpoints and will start from random')
        return 0, False, False
    with open(tracker, 'r') as f:
        meta = f.read().strip()
    try:
        iteration, release = int(meta), False
    except ValueError:
        iteration, release = 0, meta == 'release'
        if not release:
            raise ValueError('invalid metadata file {}: {!r}'.format(tracker, meta))
    assert iteration > 0 or release, 'error parsing metadata file {}'.format(tracker)
    return iteration, release, True


def extend_position_embedding(weight, length):
    """utils.py:284-288: tile a position table to a multiple of its length."""
    ori_length, hidden_size = weight.shape
    assert length % ori_length == 0
    return weight.expand(length // ori_length, -1, -1).reshape(length, hidden_size)


@contextlib.contextmanager
def reference_class_names():
    """While a checkpoint is unpickled: `fp16.loss_scaler.{LossScaler, DynamicLossScaler}` -- the path under which the reference
    (and this package once cogview_amd.bind_reference_names() has run) pickles the scaler object inside FP16_Optimizer's state
    (fp16/fp16.py:336-360) -- resolves to the mirror's classes even in a process that has not bound the reference's names."""
    from . import fp16 as _fp16
    added = []
    for name, mod in (("fp16", _fp16), ("fp16.loss_scaler", _fp16.loss_scaler)):
        if name not in sys.modules:
            sys.modules[name] = mod
            added.append(name)
    try:
        yield
    finally:
        for name in added:
            sys.modules.pop(name, None)


def load_checkpoint(model, optimizer, lr_scheduler, args, load_optimizer_states=True):
    """utils.py:290-380 (the non-DeepSpeed branch; it also reads the model-states file a DeepSpeed run wrote).
    Returns the iteration to resume from (0 for --finetune / release checkpoints or when there is none)."""
    iteration, release, success = get_checkpoint_iteration(args)
    if not success:
        return 0
    name = get_checkpoint_name(args.load, iteration, release)
    if _dp_rank() == 0:
        print('global rank {} is loading checkpoint {}'.format(_global_rank(), name))
    with reference_class_names():
        sd = torch.load(name, map_location='cpu', weights_only=False)
    model = _unwrap(model)
    if 'module' not in sd:
        raise KeyError('a metadata file exists but {} holds no model ("module")'.format(name))
    model.load_state_dict(sd['module'])
    finetune = getattr(args, 'finetune', False)
    optimizer_restored = False
    if not release and not finetune and not getattr(args, 'no_load_optim', False):
        if optimizer is not None and load_optimizer_states:
            if 'optimizer' not in sd:
                raise KeyError('{} holds no optimizer state: pass --no-load-optim or --finetune'.format(name))
            optimizer.load_state_dict(sd['optimizer'])
            optimizer_restored = True
        if lr_scheduler is not None:
            if 'lr_scheduler' in sd:
                lr_scheduler.load_state_dict(sd['lr_scheduler'])
            elif 'client_lr_scheduler' in sd:            # DeepSpeed's name for it (utils.py:307-309)
                lr_scheduler.load_state_dict(sd['client_lr_scheduler'])
    if optimizer is not None and not optimizer_restored and hasattr(optimizer, '_model_params_to_master_params'):
        # weights loaded without optimizer state (--finetune, --no-load-optim, a release file): the fp32 master copy
        # still holds the random initialisation and the first step would write it back over the loaded weights.
        # The reference refreshes the masters in this case (utils.py:300-301 refresh_fp32_params).
        optimizer._model_params_to_master_params()
    if finetune or release:
        iteration = 0
    else:
        iteration = sd['iteration'] if 'iteration' in sd else sd['total_iters']     # (older checkpoints)
    if not release and not finetune and not getattr(args, 'no_load_rng', False) and 'random_rng_state' in sd:
        random.setstate(sd['random_rng_state'])
        np.random.set_state(sd['np_rng_state'])
        torch.set_rng_state(sd['torch_rng_state'])
        if torch.cuda.is_available() and 'cuda_rng_state' in sd:
            torch.cuda.set_rng_state(sd['cuda_rng_state'])
        mpu.get_cuda_rng_tracker().set_states(sd['rng_tracker_states'])
        if 'cogv_default_dropout_state' in sd:
            seed, offset = sd['cogv_default_dropout_state']
            mpu.random.set_default_state(mpu.random._State(seed, offset))
    if _dp_rank() == 0:
        print('  successfully loaded {}'.format(name))
    return iteration
