"""The reference's train step (pretrain_gpt2.py:256-448) over this package's modules.

`get_batch`, `forward_step`, `backward_step`, `train_step` keep the reference's dataflow and return values;
what changes is where bytes move:
  * the attention mask is passed in the reference's integer "sep" form (sep = 0: left-to-right) instead of a
    materialised [1,1,s,s] tensor (mpu/sparse_transformer.py:477-489 accepts both);
  * logits stay in 16 bits between the LM head and the fused cross entropy (FP16_Module(keep_half_outputs));
  * the loss all-reduces for logging are issued only when `log` is set.
"""
import os

import torch

from . import mpu
from .functional import vocab_parallel_cross_entropy

IMG_TXT_SEP = 8192     # image tokens are ids [0, 8192): data_utils/unified_tokenizer.py:32-67, pretrain_gpt2.py:304-306


def get_batch(text, loss_mask, fp16=True):
    """pretrain_gpt2.py:256-289 for a batch already on the GPU: text [b, s+1] int64, loss_mask [b, s+1]."""
    tokens_ = text.long()
    labels = tokens_[:, 1:].contiguous()
    loss_mask = loss_mask[:, 1:].contiguous().float()
    tokens = tokens_[:, :-1].contiguous()
    b, s = tokens.shape
    position_ids = torch.arange(s, dtype=torch.long, device=tokens.device).unsqueeze(0).expand(b, s)
    attention_mask = 0                                     # left-to-right, "sep" form
    return tokens, labels, loss_mask, attention_mask, position_ids


def forward_step(batch, model, txt_loss_scale=1.0, is_sparse=0, mems=(), log=True, world_size=1):
    """pretrain_gpt2.py:292-341.  Returns (loss, mems, img_loss, txt_loss)."""
    tokens, labels, loss_mask, attention_mask, position_ids = batch
    img_indices_bool = tokens.detach() < IMG_TXT_SEP
    txt_indices_bool = (~img_indices_bool) & (loss_mask > 0)
    logits, *mems = model(tokens, position_ids, attention_mask, txt_indices_bool, img_indices_bool, is_sparse, *mems)
    losses = vocab_parallel_cross_entropy(logits, labels, inplace_backward=logits.dtype != torch.float32)
    loss_mask = loss_mask.clone()
    loss_mask[txt_indices_bool] *= txt_loss_scale
    loss_mask = loss_mask.view(-1)
    losses = losses.view(-1) * loss_mask
    loss = torch.sum(losses) / loss_mask.sum()
    img_loss = txt_loss = None
    if log:
        ib, tb = img_indices_bool.view(-1), txt_indices_bool.view(-1)
        ld = losses.detach()
        img_loss = (ld * ib).sum() / ib.sum().clamp(min=1)
        txt_loss = (ld * tb).sum() / tb.sum().clamp(min=1) / txt_loss_scale
        if world_size > 1:
            torch.distributed.all_reduce(img_loss)
            torch.distributed.all_reduce(txt_loss)
            img_loss, txt_loss = img_loss / world_size, txt_loss / world_size
    return loss, mems, img_loss, txt_loss


def backward_step(optimizer, model, lm_loss, clip_grad=1.0, fp16=True, world_size=1, reduce_loss=False):
    """pretrain_gpt2.py:344-391 (non-DeepSpeed branch).  Returns the loss averaged over all ranks when
    `reduce_loss` is set (the reference's lm_loss_reduced, :361-365), else the local loss."""
    # The backward pass starts right here, so the memset can be skipped: gradients are marked untouched and the first
    # kernel that produces each one overwrites it (arena.ParamArena.zero_grad(lazy=True)); parameters the pass did not
    # reach are zeroed afterwards.  Optimizers without a flat arena take the ordinary path.
    lazy = getattr(optimizer, 'lazy_zero_grad_ok', False) and os.environ.get('COGV_LAZY_ZERO_GRAD', '1') != '0'
    if lazy:
        optimizer.zero_grad(lazy=True)
    else:
        optimizer.zero_grad()
    if fp16:
        optimizer.backward(lm_loss, update_master_grads=False)
    else:
        lm_loss.backward()
    if lazy:
        optimizer.finish_lazy_zero_grad()
    reduced = lm_loss.detach().clone().view(1)
    if reduce_loss and world_size > 1:
        torch.distributed.all_reduce(reduced)
        reduced /= world_size
    ddp = model if hasattr(model, 'allreduce_params') else None
    if ddp is not None:
        ddp.allreduce_params(reduce_after=False)
    if fp16:
        optimizer.update_master_grads()
    if clip_grad > 0:
        if fp16:
            optimizer.clip_master_grads(clip_grad)
        else:
            mpu.clip_grad_norm(model.parameters(), clip_grad)
    return reduced


def train_step(batch, model, optimizer, lr_scheduler=None, clip_grad=1.0, txt_loss_scale=1.0, fp16=True, log=False,
               world_size=1, is_sparse=0, check_forward_nan=False):
    """pretrain_gpt2.py:406-448.  Returns (loss, skipped_iter).
    check_forward_nan: the reference's guard (:414-416) -- a non-finite sum of the all-reduced image + text losses skips
    the optimizer step WITHOUT touching the loss scale (all ranks agree because the value is all-reduced).  It needs the
    partial losses, i.e. it implies `log`.  The reference reads the value on the host between forward and backward; with the
    fused optimizer (one host read per step: the gradient statistics) the flag rides in that read instead -- backward is
    enqueued at once (the GPU queue is not drained at the forward / backward boundary: 0.8 ms per 4B step,
    profiles/r04_step_idle_gaps_4B_fp16.txt) and on a non-finite forward its gradients are simply discarded: same
    parameters, same loss scale, same return value as the reference's early return."""
    log = log or check_forward_nan
    lm_loss, _, img_loss, txt_loss = forward_step(batch, model, txt_loss_scale, is_sparse=is_sparse, log=log,
                                                  world_size=world_size)
    tot, deferred = None, False
    if check_forward_nan:
        tot = img_loss + txt_loss
        deferred = fp16 and hasattr(optimizer, 'defer_forward_nan_flag')
        if deferred:
            optimizer.defer_forward_nan_flag(tot)
        elif not bool(torch.isfinite(tot).all().item()):
            print('Skipping backward and optimizer step for nan or inf in forwarding!')
            if hasattr(model, 'needs_reduction'):
                model.needs_reduction = False
            return tot.detach(), 1
    overflow_before = getattr(optimizer, 'overflow', False)
    lm_loss = backward_step(optimizer, model, lm_loss, clip_grad, fp16, world_size=world_size, reduce_loss=log)
    if deferred and optimizer.forward_was_nan():
        print('Skipping backward and optimizer step for nan or inf in forwarding!')
        # the reference returns before backward: its optimizer never sees these gradients.  Here they were produced and
        # measured (update_master_grads) before the flag reached the host -- put the optimizer's visible state back: the
        # overflow flag (saved by state_dict()) as it was, the statistics of the discarded gradients forgotten
        optimizer.overflow = overflow_before
        if hasattr(optimizer, '_stats_valid'):
            optimizer._stats_valid = False
        return tot.detach(), 1
    optimizer.step()
    skipped = 0
    if not (fp16 and optimizer.overflow):
        if lr_scheduler is not None:
            lr_scheduler.step()
    else:
        skipped = 1
    return lm_loss.view(()), skipped
