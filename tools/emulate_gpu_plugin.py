"""Development aid: run the HOST-LEVEL part of the GPU-marked tests on the CPU emulation of the kernels (tests/cpu_ops.py).

    python -m pytest -p tools.emulate_gpu_plugin tests/test_checkpoint_gpu.py tests/test_model_gpu.py -m gpu -q

A pytest plugin, loaded only when named with -p: `cogview_amd.ops`' entry points are replaced by their torch-CPU restatements,
device='cuda' / .cuda() / .to('cuda') resolve to the CPU (a TorchFunctionMode), torch.cuda's stream / event objects are inert and
process groups come up over gloo.  What it is for: the build container has no GPU, and a round's GPU minutes run out -- a change to
the host path (optimizer, checkpointing, data-parallel wrapper, dropout-state bookkeeping) can still be held to the assertions
the GPU tests make about it.  What it is NOT: evidence about the kernels -- a test that passes here has not touched
libcogview_hip.so; the driver's GPU run never loads this plugin.  Forms the emulation does not have (decode / matrix-vector
kernels, sparse training, arbitrary mask tensors, stored keep bits, VQ-VAE convolutions) and tests that spawn their own worker
processes fail here by construction."""
import contextlib, sys, os
import torch
from torch.overrides import TorchFunctionMode
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def _is_cuda_dev(d):
    return (isinstance(d, str) and d.startswith("cuda")) or (isinstance(d, torch.device) and d.type == "cuda")

class _Mode(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if _is_cuda_dev(kwargs.get("device")):
            kwargs["device"] = "cpu"
        name = getattr(func, "__name__", "")
        if name == "cuda" and args and isinstance(args[0], torch.Tensor):
            return args[0]
        if name == "to" and len(args) > 1 and _is_cuda_dev(args[1]):
            args = (args[0], "cpu") + tuple(args[2:])
        return func(*args, **kwargs)

class _Inert:
    def __init__(self, *a, **k): pass
    def wait_stream(self, s): pass
    def wait_event(self, e): pass
    def record(self, s=None): pass
    def synchronize(self): pass
    def query(self): return True
    cuda_stream = 0

def pytest_configure(config):
    from tests import cpu_ops
    cpu_ops.install()
    torch.Tensor.is_cuda = property(lambda self: True)
    torch.nn.Module.cuda = lambda self, device=None: self
    torch.cuda.is_available = lambda: True
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.current_device = lambda: 0
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.Stream = _Inert; torch.cuda.Event = _Inert
    torch.cuda.current_stream = lambda *a, **k: _Inert()
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    torch.cuda.manual_seed = lambda *a, **k: None
    torch.cuda.get_rng_state = lambda *a, **k: torch.zeros(16, dtype=torch.uint8)
    torch.cuda.set_rng_state = lambda *a, **k: None
    torch.cuda.empty_cache = lambda: None
    import torch.distributed as dist
    real_init = dist.init_process_group
    def init(backend=None, *a, **k):
        return real_init("gloo", *a, **k)
    dist.init_process_group = init
    real_new = dist.new_group
    dist.new_group = lambda *a, **k: real_new(*a, **{**k, "backend": "gloo"})
    config._emu_mode = _Mode()
    config._emu_mode.__enter__()

def pytest_unconfigure(config):
    config._emu_mode.__exit__(None, None, None)
