"""clip_grad_norm with the semantics of mpu/grads.py:28-74: global norm over the model-parallel group with
non-model-parallel parameters counted once (on model-parallel rank 0), in-place scaling by
max_norm / (norm + 1e-6) when that is < 1.  The reference syncs with the host once per tensor
(`.norm().item()`); here the sum of squares is one device kernel per tensor (or ONE kernel when the
gradients live in a flat arena -- see fp16.FP16_Optimizer) and a single host read."""
import torch

from .. import ops
from .initialize import get_model_parallel_group, mp_rank_or_0, mp_world_size_or_1

inf = float('inf')


def _chunk_tables(grads):
    """[(flat view, chunk_start, chunk_len, chunk_norm)]: ONE chunk table per underlying storage -- the gradients of a model
    built on the flat arena (cogview_amd/arena.py) are views of one buffer, so the whole norm is one cogv_grad_stats launch
    (round 3 issued two launches per parameter: ~1.5k at 48 layers).  A tensor whose offset in its storage is not a
    multiple of 8 elements (the kernel's 16-byte vector loads) gets a table of its own on a contiguous copy."""
    by_storage, out = {}, []
    for g in grads:
        st = g.untyped_storage()
        off = g.storage_offset()
        if g.is_contiguous() and off % 8 == 0 and st.data_ptr() % 16 == 0:
            by_storage.setdefault((st.data_ptr(), g.dtype), (st, [])).__getitem__(1).append((off, g.numel()))
        else:
            gc = g.contiguous().view(-1)
            if gc.data_ptr() % 16:                  # a contiguous view at an odd offset: .contiguous() returned the view itself
                gc = gc.clone()
            # one workgroup reduces one chunk with fp32 partial sums: cut a large loose tensor into chunks of <= 1M elements
            # (multiples of 8) so that the reduction is parallel and its accumulation error stays bounded
            n, step = gc.numel(), 1 << 20
            out.append((gc, [(o, min(step, n - o)) for o in range(0, n, step)] or [(0, 0)]))
    for (_, dtype), (st, chunks) in by_storage.items():
        flat = torch.empty(0, dtype=dtype, device=grads[0].device).set_(st, 0, (st.nbytes() // dtype.itemsize,))
        out.append((flat, sorted(chunks)))
    dev = grads[0].device
    return [(flat, torch.tensor([c[0] for c in ch], dtype=torch.int64, device=dev),
             torch.tensor([c[1] for c in ch], dtype=torch.int32, device=dev),
             torch.ones(len(ch), dtype=torch.uint8, device=dev)) for flat, ch in out]


def clip_grad_norm(parameters, max_norm, norm_type=2):
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    parameters = [p for p in parameters if p.grad is not None]
    max_norm, norm_type = float(max_norm), float(norm_type)
    if not parameters:
        return 0.0
    dev = parameters[0].grad.device
    # DEVICE-ONLY for the two norms the training scripts use (2 and inf): they are HIP kernels (cogv_grad_stats / cogv_absmax) for
    # 16-bit and fp32 gradients alike, and gradients on the CPU raise CogviewHipError there -- by design: a silent torch
    # fallback on the product path would void the parity claims made for it (the CPU statement of this function lives with the
    # test infrastructure, not here).  Only the general p-norm below, which nothing on the training path asks for, is torch.
    mp = mp_world_size_or_1()
    if norm_type == inf:
        slot = ops.new_absmax_slot(dev)
        for p in parameters:
            g = p.grad.data
            g = g.contiguous()
            if g.dtype == torch.float32 and (g.numel() % 4 or g.data_ptr() % 16):
                torch.maximum(slot, g.abs().max().reshape(1), out=slot)      # an odd-sized fp32 tensor: cogv_absmax reads 16 B
            else:
                ops.absmax(g, slot)
        total = slot.clone()
        if mp > 1:
            torch.distributed.all_reduce(total, op=torch.distributed.ReduceOp.MAX, group=get_model_parallel_group())
        total_norm = total.item()
    elif norm_type == 2.0:
        stats = torch.zeros(2, dtype=torch.float64, device=dev)
        counted = [p.grad.data for p in parameters if getattr(p, 'model_parallel', False) or mp_rank_or_0() == 0]
        if counted:                                   # 16-bit model gradients and fp32 master gradients alike: cogv_grad_stats
            for flat, cs, cl, cn in _chunk_tables(counted):
                ops.grad_stats(flat, cs, cl, cn, stats)
        if mp > 1:
            torch.distributed.all_reduce(stats, group=get_model_parallel_group())
        total_norm = float(stats[0].item()) ** 0.5
    else:
        # general p-norm (mpu/grads.py:59-69): sum of |g|^p over the counted parameters, summed over the model-parallel
        # group, then the p-th root.  Nothing on the training path asks for it (the scripts clip the 2-norm), so this is
        # the reference's own formulation on torch reductions: one device reduction per tensor, ONE host read.
        acc = torch.zeros(1, dtype=torch.float64, device=dev)
        for p in parameters:
            if getattr(p, 'model_parallel', False) or mp_rank_or_0() == 0:
                acc += p.grad.data.double().abs().pow(norm_type).sum()
        if mp > 1:
            torch.distributed.all_reduce(acc, group=get_model_parallel_group())
        total_norm = float(acc.item()) ** (1.0 / norm_type)
    clip_coef = max_norm / (total_norm + 1e-6)
    if clip_coef < 1:
        for p in parameters:
            g = p.grad.data
            if g.dtype == torch.float32 or g.numel() % 8 or not g.is_contiguous() or g.data_ptr() % 16:
                g.mul_(clip_coef)
            else:
                g.copy_(ops.scale(g, clip_coef))
    return total_norm
