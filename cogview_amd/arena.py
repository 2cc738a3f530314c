"""Flat parameter / gradient arena.

MI355X-first memory layout: all parameters of a module live in ONE contiguous 16-bit buffer (each tensor
256-byte aligned) and all gradients in a second buffer with the same offsets; `param.data` / `param.grad`
are views.  288 GB of HBM3E per GPU makes the extra fp32 master / Adam-state copies (16 B per parameter,
62.9 GB for the 4B model) resident without sharding, and the flat layout turns every per-tensor loop of the
reference (772 tensors at 48 layers) into one kernel or one collective:
    zero_grad            one memset
    DP all-reduce        contiguous slices in layer order (overlappable with backward)
    overflow + norm      cogv_grad_stats over a chunk table
    AdamW + cast         cogv_adamw_step over the same chunk table
"""
import torch

ALIGN = 128          # elements (256 B)
CHUNK = 65536        # elements per optimizer chunk


class ParamArena:
    def __init__(self, params, dtype, device):
        self.params = list(params)
        self.offsets, total = [], 0
        for p in self.params:
            self.offsets.append(total)
            total += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = total
        self.data = torch.zeros(total, dtype=dtype, device=device)
        self.grad = torch.zeros(total, dtype=dtype, device=device)
        for p, off in zip(self.params, self.offsets):
            view = self.data[off:off + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.grad[off:off + p.numel()].view(p.shape)
            p._cogv_arena = (self, off)
        self._tables = {}
        self._fresh = None
        # Lazy zero_grad contract (see zero_grad): it is sound only while EVERY gradient of the arena is produced by a
        # kernel that consults functional.grad_accumulate / ensure_zeroed.  `lazy_ok` is the owner's declaration of that
        # (flatten_module copies module._cogv_lazy_zero_grad: GPT2Model sets it, any other module does not), and the
        # hook below is the guard: a gradient that arrives through autograd's AccumulateGrad (a plain torch op on a
        # parameter) while gradients are marked untouched would be added onto stale data -- fail loudly instead.
        self.lazy_ok = False
        for p in self.params:
            if p.requires_grad and hasattr(p, "register_post_accumulate_grad_hook"):
                p.register_post_accumulate_grad_hook(self._autograd_accumulated)

    def _autograd_accumulated(self, p):
        if self._fresh is not None and id(p) in self._fresh:
            raise RuntimeError(
                "a gradient of shape %s reached its arena parameter through autograd's AccumulateGrad while a lazy "
                "zero_grad was pending: it was added onto stale data.  Lazy zero_grad is only valid for modules whose "
                "parameter gradients all come from this package's fused backward kernels; call optimizer.zero_grad() "
                "(a memset) for this model, or set COGV_LAZY_ZERO_GRAD=0" % (tuple(p.shape),))

    def zero_grad(self, lazy=False):
        """lazy=False: one memset.  lazy=True: no memset -- every gradient is only MARKED untouched, and the backward
        kernel that produces it first overwrites instead of accumulating (take_fresh); whatever is still untouched when
        the gradients are consumed is zeroed then (finish_lazy).  Saves the 7.9-GB memset and the 7.9-GB read of the
        accumulate epilogues per step at 4B; `param.grad` holds stale values between the call and the backward pass."""
        if lazy and not self.lazy_ok:
            raise RuntimeError("lazy zero_grad on an arena whose module did not declare _cogv_lazy_zero_grad")
        if lazy:
            self._fresh = {id(p) for p in self.params}
        else:
            self._fresh = None
            self.grad.zero_()
        for p, off in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + off * self.grad.element_size():
                p.grad = self.grad[off:off + p.numel()].view(p.shape)

    def take_fresh(self, params):
        """True when EVERY gradient of `params` is still untouched since a lazy zero_grad: the caller's kernel must then
        write (not accumulate) all of them.  Otherwise the untouched ones among them are zeroed here and the caller
        accumulates.  Either way they count as written from now on."""
        fresh = self._fresh
        if not fresh:
            return False
        ids = [id(p) for p in params]
        hit = [i in fresh for i in ids]
        for i in ids:
            fresh.discard(i)
        if all(hit):
            return True
        for p, h in zip(params, hit):
            if h:
                p.grad.zero_()
        return False

    def ensure_zeroed(self, p):
        """For kernels that can only add (scatter-add with atomics): zero p's gradient now if it is still untouched."""
        if self._fresh and id(p) in self._fresh:
            self._fresh.discard(id(p))
            p.grad.zero_()

    def finish_lazy(self):
        """Zero the gradients no kernel has written since a lazy zero_grad (parameters that took no part in the
        backward pass).  Called by whoever consumes the gradients (optimizer statistics, data-parallel exchange)."""
        fresh = self._fresh
        if fresh:
            for p in self.params:
                if id(p) in fresh:
                    p.grad.zero_()
        self._fresh = None

    def slice_of(self, params):
        """(start, end) element range covering `params` (which must be contiguous in the arena)."""
        want = {id(p) for p in params}
        idx = [i for i, q in enumerate(self.params) if id(q) in want]
        assert idx and idx == list(range(idx[0], idx[-1] + 1)), "parameters are not contiguous in the arena"
        last = idx[-1]
        return self.offsets[idx[0]], self.offsets[last] + (self.params[last].numel() + ALIGN - 1) // ALIGN * ALIGN

    def chunk_table(self, group_of, norm_of, owned=None):
        """Device chunk table for cogv_grad_stats / cogv_adamw_step.
        group_of(param) -> hyper-parameter group index; norm_of(param) -> bool (counted in the global norm).
        owned: optional sorted list of disjoint element ranges [(s, e)] (multiples of 8): only the parts of the parameters
        inside them are listed (a data-parallel rank's share when the optimizer is sharded)."""
        starts, lens, groups, norms = [], [], [], []
        oi = 0
        for p, off in zip(self.params, self.offsets):
            g, n = group_of(p), norm_of(p)
            spans = [(off, off + p.numel())]
            if owned is not None:
                spans = []
                while oi < len(owned) and owned[oi][1] <= off:
                    oi += 1
                j = oi
                while j < len(owned) and owned[j][0] < off + p.numel():
                    a, b = max(owned[j][0], off), min(owned[j][1], off + p.numel())
                    if b > a:
                        spans.append((a, b))
                    j += 1
            for a, b in spans:
                for c0 in range(a, b, CHUNK):
                    starts.append(c0)
                    lens.append(min(CHUNK, b - c0))
                    groups.append(g)
                    norms.append(1 if n else 0)
        if not starts:                       # a rank may own nothing of a tiny model: one empty chunk keeps the kernels happy
            starts, lens, groups, norms = [0], [0], [0], [0]
        dev = self.data.device
        return (torch.tensor(starts, dtype=torch.int64, device=dev), torch.tensor(lens, dtype=torch.int32, device=dev),
                torch.tensor(groups, dtype=torch.uint8, device=dev), torch.tensor(norms, dtype=torch.uint8, device=dev))


def flatten_module(module, dtype=None):
    """Move every parameter of `module` (already on the GPU, already in its 16-bit dtype) into one arena.
    Tied parameters appear once.  Returns the arena (also stored as module._cogv_arena)."""
    params, seen = [], set()
    for p in module.parameters():
        if id(p) not in seen:
            seen.add(id(p))
            params.append(p)
    assert params, "module has no parameters"
    dtype = dtype or params[0].dtype
    assert all(p.dtype == dtype for p in params), "arena needs a single parameter dtype"
    assert all(p.is_cuda for p in params), "arena needs GPU parameters"
    arena = ParamArena(params, dtype, params[0].device)
    arena.lazy_ok = bool(getattr(module, "_cogv_lazy_zero_grad", False))
    module._cogv_arena = arena
    return arena


def arena_of(params):
    """The arena that holds exactly these parameters (None if they are not all in one arena)."""
    a = None
    for p in params:
        e = getattr(p, "_cogv_arena", None)
        if e is None:
            return None
        if a is None:
            a = e[0]
        elif a is not e[0]:
            return None
    return a
