"""RNG-state tracker and activation checkpointing with the reference's API (mpu/random.py:85-384).

MI355X-first difference: dropout in this package is COUNTER-BASED.  Every dropout site draws a fresh
`stream_id` from the current RNG state (a (seed, offset) pair of Python ints) and the HIP kernels derive the
mask from (seed, stream_id, element index).  So an "RNG state" is two integers, forking / restoring a state is
free (the reference copies 816-byte CUDA generator states through the host), and activation-checkpoint
recompute replays identical masks simply by restoring the offsets -- the property the reference gets from
saving/restoring CUDA RNG states in CheckpointFunction (mpu/random.py:308-310,353-355).

Two states are tracked exactly as model_parallel_cuda_manual_seed documents (mpu/random.py:198-233):
  default state          seed            same inside a model-parallel group (hidden / embedding dropout)
  'model-parallel-rng'   seed+2718+rank  different inside a model-parallel group (attention dropout)
"""
import contextlib

import torch

from .initialize import (get_data_parallel_rank, get_model_parallel_group, get_model_parallel_rank, mp_rank_or_0,
                         mp_world_size_or_1)

_MODEL_PARALLEL_RNG_TRACKER_NAME = 'model-parallel-rng'

PARTITION_ACTIVATIONS = False


class _State:
    __slots__ = ("seed", "offset")

    def __init__(self, seed, offset=0):
        self.seed, self.offset = int(seed), int(offset)

    def clone(self):
        return _State(self.seed, self.offset)


_DEFAULT_STATE = _State(torch.initial_seed() & 0x7FFFFFFFFFFFFFFF)
_CURRENT = [_DEFAULT_STATE]          # the state dropout sites draw from


def next_dropout_stream():
    """(seed, stream_id) for one dropout site; advances the current state."""
    st = _CURRENT[0]
    st.offset += 1
    return st.seed, st.offset


def manual_seed(seed):
    """Seed of the default (data-parallel) dropout state; the analogue of torch.cuda.manual_seed."""
    _DEFAULT_STATE.seed, _DEFAULT_STATE.offset = int(seed), 0


def get_default_state():
    return _DEFAULT_STATE.clone()


def set_default_state(state):
    _DEFAULT_STATE.seed, _DEFAULT_STATE.offset = state.seed, state.offset


_DEFAULT_STATE_KEY = "cogview-amd-default-dropout-state"


class CudaRNGStatesTracker:
    """Named dropout RNG states (API of mpu/random.py:85-180)."""

    def __init__(self):
        self.states_ = {}
        self.seeds_ = set()

    def reset(self):
        self.states_ = {}
        self.seeds_ = set()

    def get_states(self):
        """{name: int64 tensor [seed, offset]} -- built-in types only: utils.save_checkpoint (the reference's :219-225 as well
        as this package's) pickles the result as sd['rng_tracker_states'], and a file that names a class of this package could
        not be opened by the reference at all (torch.load unpickles the whole dictionary, whatever --no-load-rng says).  The
        reference stores byte tensors of torch.cuda.get_rng_state() here; the dropout states of the two stacks are not
        interchangeable (different generators): pass --no-load-rng when a checkpoint crosses over."""
        out = {name: torch.tensor([st.seed, st.offset], dtype=torch.int64) for name, st in self.states_.items()}
        # the default (data-parallel) dropout state rides along under a reserved name: the reference's save_checkpoint knows
        # nothing of it (it saves torch.cuda.get_rng_state(), which no kernel here reads), and a resume through the reference's
        # utils.py would otherwise restart the hidden-dropout streams
        out[_DEFAULT_STATE_KEY] = torch.tensor([_DEFAULT_STATE.seed, _DEFAULT_STATE.offset], dtype=torch.int64)
        return out

    def set_states(self, states):
        """Accepts what get_states() returns, and the _State objects files written before round 5's end carry."""
        new = {}
        for name, st in states.items():
            if name == _DEFAULT_STATE_KEY:
                seed, offset = (int(v) for v in st.tolist())
                set_default_state(_State(seed, offset))
                continue
            if isinstance(st, _State):
                new[name] = st.clone()
            else:
                if isinstance(st, torch.Tensor) and (st.dtype == torch.uint8 or st.numel() != 2):
                    # a checkpoint written by the REFERENCE: torch.cuda.get_rng_state() ByteTensors (mpu/random.py:163-168), a
                    # generator state no kernel here can resume.  The reference's own load_checkpoint tells the user the same
                    # thing when its states do not fit (utils.py:361-366)
                    raise ValueError("rng tracker state '{}' is a torch CUDA generator state ({} x {}), not this package's "
                                     "(seed, offset) pair: the checkpoint was written by another implementation -- load it "
                                     "with --no-load-rng".format(name, st.numel(), st.dtype))
                seed, offset = (int(v) for v in (st.tolist() if isinstance(st, torch.Tensor) else st))
                new[name] = _State(seed, offset)
        self.states_ = new

    def add(self, name, seed):
        if seed in self.seeds_:
            raise Exception('seed {} already exists'.format(seed))
        self.seeds_.add(seed)
        if name in self.states_:
            raise Exception('cuda rng state {} already exists'.format(name))
        self.states_[name] = _State(seed)

    @contextlib.contextmanager
    def fork(self, name=_MODEL_PARALLEL_RNG_TRACKER_NAME):
        if name not in self.states_:
            raise Exception('cuda rng state {} is not added'.format(name))
        prev = _CURRENT[0]
        _CURRENT[0] = self.states_[name]
        try:
            yield
        finally:
            _CURRENT[0] = prev


_CUDA_RNG_STATE_TRACKER = CudaRNGStatesTracker()


def get_cuda_rng_tracker():
    return _CUDA_RNG_STATE_TRACKER


def model_parallel_cuda_manual_seed(seed):
    """mpu/random.py:198-233: default state <- seed, model-parallel state <- seed + 2718 + mp_rank."""
    offset = seed + 2718
    model_parallel_seed = offset + mp_rank_or_0()
    if torch.distributed.is_initialized() and torch.distributed.get_rank() == 0:
        print('> initializing model parallel cuda seeds on global rank {}, model parallel rank {}, and data '
              'parallel rank {} with model parallel seed: {} and data parallel seed: {}'.format(
                  torch.distributed.get_rank(), get_model_parallel_rank(), get_data_parallel_rank(),
                  model_parallel_seed, seed), flush=True)
    _CUDA_RNG_STATE_TRACKER.reset()
    manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    _CUDA_RNG_STATE_TRACKER.add(_MODEL_PARALLEL_RNG_TRACKER_NAME, model_parallel_seed)


def attention_dropout_stream():
    """Stream for attention dropout: drawn under the model-parallel fork when the tracker is seeded
    (mpu/sparse_transformer.py:667-669), else from the default state."""
    if _MODEL_PARALLEL_RNG_TRACKER_NAME in _CUDA_RNG_STATE_TRACKER.states_:
        with _CUDA_RNG_STATE_TRACKER.fork():
            return next_dropout_stream()
    return next_dropout_stream()


def detach_variable(inputs):
    out = []
    for inp in inputs:
        if not isinstance(inp, torch.Tensor):
            out.append(inp)
            continue
        x = inp.detach()
        x.requires_grad = inp.requires_grad
        out.append(x)
    return tuple(out)


def partition_activation(t):
    """mpu/random.py:298-310 (with `partition_activations_in_checkpoint(True)`): a checkpointed activation is kept as this
    model-parallel rank's 1/p of its elements.  Returns (what to save, original shape or None when kept whole)."""
    mp = mp_world_size_or_1()
    if not PARTITION_ACTIVATIONS or mp == 1 or t.numel() % mp or not t.is_floating_point():
        return t, None
    n = t.numel() // mp
    r = mp_rank_or_0()
    return t.detach().contiguous().view(-1)[r * n:(r + 1) * n].clone(), tuple(t.shape)


def gather_activation(part, shape):
    """mpu/random.py:248-266 (get_full_inputs): rebuild the full activation from every model-parallel rank's piece."""
    if shape is None:
        return part
    mp = mp_world_size_or_1()
    full = torch.empty(part.numel() * mp, dtype=part.dtype, device=part.device)
    pieces = list(full.chunk(mp))
    torch.distributed.all_gather(pieces, part.contiguous(), group=get_model_parallel_group())
    return full.view(shape)


class CheckpointFunction(torch.autograd.Function):
    """Re-entrant activation checkpoint (mpu/random.py:273-372): forward under no_grad keeping only the
    inputs and the RNG states; backward restores the states, recomputes with grad, and backpropagates."""

    @staticmethod
    def forward(ctx, run_function, *args):
        ctx.run_function = run_function
        ctx.fwd_cpu_rng_state = torch.get_rng_state()
        ctx.fwd_default_state = get_default_state()
        ctx.fwd_tracker_states = get_cuda_rng_tracker().get_states()
        ctx.tensor_idx = [i for i, a in enumerate(args) if isinstance(a, torch.Tensor)]
        ctx.other = [None if isinstance(a, torch.Tensor) else a for a in args]
        # activation partitioning (mpu/random.py:298-310): every tensor argument but the last one (the mask) is kept as
        # this rank's slice and all-gathered again in backward
        saved, ctx.part_shapes, ctx.part_grad = [], [], []
        for k, i in enumerate(ctx.tensor_idx):
            t = args[i]
            if k < len(ctx.tensor_idx) - 1:
                piece, shape = partition_activation(t)
            else:
                piece, shape = t, None
            saved.append(piece)
            ctx.part_shapes.append(shape)
            ctx.part_grad.append(t.requires_grad)
        ctx.save_for_backward(*saved)
        with torch.no_grad():
            outputs = run_function(*args)
        return outputs

    @staticmethod
    def backward(ctx, *grads):
        if not torch.autograd._is_checkpoint_valid():
            raise RuntimeError("Checkpointing is not compatible with .grad(), please use .backward() if possible")
        args = list(ctx.other)
        for i, t, shape, rg in zip(ctx.tensor_idx, ctx.saved_tensors, ctx.part_shapes, ctx.part_grad):
            if shape is not None:
                t = gather_activation(t, shape)
                t.requires_grad = rg
            args[i] = t
        detached = detach_variable(tuple(args))
        bwd_cpu = torch.get_rng_state()
        bwd_default = get_default_state()
        bwd_tracker = get_cuda_rng_tracker().get_states()
        torch.set_rng_state(ctx.fwd_cpu_rng_state)
        set_default_state(ctx.fwd_default_state)
        get_cuda_rng_tracker().set_states(ctx.fwd_tracker_states)
        with torch.enable_grad():
            outputs = ctx.run_function(*detached)
        torch.set_rng_state(bwd_cpu)
        set_default_state(bwd_default)
        get_cuda_rng_tracker().set_states(bwd_tracker)
        if isinstance(outputs, torch.Tensor):
            outputs = (outputs,)
        pairs = [(o, g) for o, g in zip(outputs, grads) if isinstance(o, torch.Tensor) and o.requires_grad]
        from .. import functional as F_                       # (functional imports mpu: resolved at call time)
        with F_.nested_backward():                            # deferred weight gradients of the recomputed layers join the outer pass' queue
            torch.autograd.backward([o for o, _ in pairs], [g for _, g in pairs])
        return (None,) + tuple(inp.grad if isinstance(inp, torch.Tensor) else None for inp in detached)


def checkpoint(function, *args):
    return CheckpointFunction.apply(function, *args)


def partition_activations_in_checkpoint(partition_activation):
    """mpu/random.py:380-384.  With it on, a checkpointed layer input is kept as each model-parallel rank's 1/p slice and
    all-gathered over the model-parallel group when the layer is recomputed (a memory device for 32 GB GPUs: one 4B layer
    input is 267 MB at b = 24 -- 13 GB over 48 layers -- against 288 GB of HBM3E, so nothing here turns it on)."""
    global PARTITION_ACTIVATIONS
    PARTITION_ACTIVATIONS = bool(partition_activation)
