"""Training entry point over this package -- the non-DeepSpeed path of the reference's pretrain_gpt2.py:58-819 reduced to
what the hot path consumes: build GPT2Model -> FP16_Module -> data-parallel wrapper, FusedAdam inside FP16_Optimizer,
AnnealingLR, CompactBinaryDataset batches sliced per data-parallel rank, the train loop with logging and checkpoints in
the reference's formats.  Flag names are the reference's (arguments.py); flags of subsystems that are out of scope
(DeepSpeed, tensorboard, the text tokenizer, LMDB data) are not defined.

    python -m torch.distributed.run --nproc-per-node 8 -m cogview_amd.pretrain_gpt2 \\
        --num-layers 48 --hidden-size 2560 --num-attention-heads 40 --max-position-embeddings 1089 --vocab-size 58240 \\
        --batch-size 24 --train-iters 300000 --lr 1e-4 --lr-decay-style cosine --warmup 0.02 --fp16 \\
        --train-data /data/cogview/train.bin --save /ckpt --save-interval 1000
"""
import argparse
import os
import random
import time

import numpy as np
import torch

from . import mpu, training, utils
from .data_utils import RandomMappingDataset, get_dataset_by_type
from .fp16 import FP16_Module, FP16_Optimizer
from .learning_rates import AnnealingLR
from .model import GPT2Model, PyTorchDistributedDataParallel, gpt2_get_params_for_weight_decay_optimization
from .optim import FusedAdam


def get_args(argv=None):
    p = argparse.ArgumentParser(description="CogView GPT pre-training on MI355X (cogview_amd)")
    g = p.add_argument_group("model")
    g.add_argument("--num-layers", type=int, default=24)
    g.add_argument("--hidden-size", type=int, default=1024)
    g.add_argument("--num-attention-heads", type=int, default=16)
    g.add_argument("--max-position-embeddings", type=int, default=1089)
    g.add_argument("--vocab-size", type=int, default=58240)
    g.add_argument("--attention-dropout", type=float, default=0.1)
    g.add_argument("--hidden-dropout", type=float, default=0.1)
    g.add_argument("--max-memory-length", type=int, default=0)
    g.add_argument("--is-sparse", type=int, default=0, choices=[0, 1, 2])
    g.add_argument("--query-window", type=int, default=128)
    g.add_argument("--key-window-times", type=int, default=6)
    g.add_argument("--num-pivot", type=int, default=768)
    g = p.add_argument_group("precision")
    g.add_argument("--fp16", action="store_true")
    g.add_argument("--bf16", action="store_true", help="extension: bf16 compute with fp32 masters")
    g.add_argument("--loss-scale", type=float, default=None)
    g.add_argument("--loss-scale-window", type=float, default=1000)
    g.add_argument("--min-scale", type=float, default=1)
    g.add_argument("--hysteresis", type=int, default=2)
    g = p.add_argument_group("train")
    g.add_argument("--batch-size", type=int, default=4, help="per data-parallel rank")
    g.add_argument("--weight-decay", type=float, default=0.01)
    g.add_argument("--checkpoint-activations", action="store_true")
    g.add_argument("--checkpoint-num-layers", type=int, default=1)
    g.add_argument("--clip-grad", type=float, default=1.0)
    g.add_argument("--train-iters", type=int, default=1000000)
    g.add_argument("--log-interval", type=int, default=100)
    g.add_argument("--exit-interval", type=int, default=None)
    g.add_argument("--seed", type=int, default=1234)
    g.add_argument("--txt-loss-scale", type=float, default=1.0)
    g.add_argument("--lr", type=float, default=1e-4)
    g.add_argument("--lr-decay-iters", type=int, default=None)
    g.add_argument("--lr-decay-style", type=str, default="linear", choices=["constant", "linear", "cosine", "exponential"])
    g.add_argument("--lr-decay-ratio", type=float, default=0.1)
    g.add_argument("--warmup", type=float, default=0.01)
    g.add_argument("--save", type=str, default=None)
    g.add_argument("--save-interval", type=int, default=5000)
    g.add_argument("--no-save-optim", action="store_true")
    g.add_argument("--no-save-rng", action="store_true")
    g.add_argument("--load", type=str, default=None)
    g.add_argument("--no-load-optim", action="store_true")
    g.add_argument("--no-load-rng", action="store_true")
    g.add_argument("--finetune", action="store_true")
    g.add_argument("--distributed-backend", default="nccl", choices=["nccl", "gloo"])
    g.add_argument("--model-parallel-size", type=int, default=1)
    g = p.add_argument_group("data")
    g.add_argument("--train-data", nargs="+", default=None, help="CompactBinaryDataset files (64 text ids + 1024 codes per row)")
    g.add_argument("--dataset-type", type=str, default="CompactBinaryDataset")
    g.add_argument("--num-workers", type=int, default=2)
    args = p.parse_args(argv)
    args.deepspeed = False
    args.rank = int(os.getenv("RANK", "0"))
    args.world_size = int(os.getenv("WORLD_SIZE", "1"))
    args.local_rank = int(os.getenv("LOCAL_RANK", "0"))
    args.dynamic_loss_scale = args.loss_scale is None
    args.iteration = 0
    return args


def initialize_distributed(args):
    """pretrain_gpt2.py:613-634: one process per GPU, RCCL process group, model-parallel grid."""
    torch.cuda.set_device(args.local_rank % max(1, torch.cuda.device_count()))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "6000")
    torch.distributed.init_process_group(backend=args.distributed_backend, world_size=args.world_size, rank=args.rank,
                                         init_method="env://")
    mpu.initialize_model_parallel(args.model_parallel_size)


def set_random_seed(seed):
    """pretrain_gpt2.py:643-651."""
    if seed is not None and seed > 0:
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        mpu.model_parallel_cuda_manual_seed(seed)


def get_model(args):
    """pretrain_gpt2.py:58-105."""
    utils.print_rank_0('building CogView2 model ...')
    model = GPT2Model(num_layers=args.num_layers, vocab_size=args.vocab_size, hidden_size=args.hidden_size,
                      num_attention_heads=args.num_attention_heads, embedding_dropout_prob=args.hidden_dropout,
                      attention_dropout_prob=args.attention_dropout, output_dropout_prob=args.hidden_dropout,
                      max_sequence_length=args.max_position_embeddings, max_memory_length=args.max_memory_length,
                      checkpoint_activations=args.checkpoint_activations,
                      checkpoint_num_layers=args.checkpoint_num_layers, parallel_output=True,
                      query_window=args.query_window, key_window_times=args.key_window_times, num_pivot=args.num_pivot)
    if mpu.get_data_parallel_rank() == 0:
        print(' > number of parameters on model parallel rank {}: {}'.format(
            mpu.get_model_parallel_rank(), sum(p.nelement() for p in model.parameters())), flush=True)
    model.cuda(torch.cuda.current_device())
    if args.fp16 or args.bf16:
        model = FP16_Module(model, dtype=torch.bfloat16 if args.bf16 else torch.float16, keep_half_outputs=True)
    i = torch.cuda.current_device()
    return PyTorchDistributedDataParallel(model, device_ids=[i], output_device=i,
                                          process_group=mpu.get_data_parallel_group())


def get_optimizer_param_groups(model):
    """pretrain_gpt2.py:108-121."""
    while hasattr(model, "module"):
        model = model.module
    groups = gpt2_get_params_for_weight_decay_optimization(model)
    for group in groups:
        for param in group['params']:
            if not hasattr(param, 'model_parallel'):
                param.model_parallel = False
    return groups


def get_optimizer(param_groups, args):
    """pretrain_gpt2.py:124-158: FusedAdam (decoupled weight decay) inside the loss-scaling wrapper."""
    optimizer = FusedAdam(param_groups, lr=args.lr, weight_decay=args.weight_decay)
    if args.fp16 or args.bf16:
        optimizer = FP16_Optimizer(optimizer, static_loss_scale=args.loss_scale if args.loss_scale else 1.0,
                                   dynamic_loss_scale=args.dynamic_loss_scale and not args.bf16,
                                   dynamic_loss_args={'scale_window': args.loss_scale_window, 'min_scale': args.min_scale,
                                                      'delayed_shift': args.hysteresis})
    return optimizer


def get_learning_rate_scheduler(optimizer, args):
    """pretrain_gpt2.py:161-181."""
    num_iters = max(1, args.lr_decay_iters if args.lr_decay_iters is not None else args.train_iters)
    return AnnealingLR(optimizer, start_lr=args.lr, warmup_iter=args.warmup * num_iters, num_iters=num_iters,
                       decay_style=args.lr_decay_style, last_iter=-1, decay_ratio=args.lr_decay_ratio)


def setup_model_and_optimizer(args):
    """pretrain_gpt2.py:184-209."""
    model = get_model(args)
    optimizer = lr_scheduler = None
    if args.train_data is not None:
        optimizer = get_optimizer(get_optimizer_param_groups(model), args)
        lr_scheduler = get_learning_rate_scheduler(optimizer, args)
    return model, optimizer, lr_scheduler


class DistributedBatchSampler(torch.utils.data.Sampler):
    """data_utils/samplers.py:106-169 for the training case (drop_last, no wrap): the GLOBAL batch of
    batch_size * data-parallel world indices is formed first, then every rank keeps its contiguous slice."""

    def __init__(self, sampler, global_batch_size, rank, world_size, start_iter=0):
        self.sampler, self.batch_size, self.rank, self.world_size = sampler, global_batch_size, rank, world_size
        self.start_iter = start_iter

    def __iter__(self):
        batch, i = [], 0
        for idx in self.sampler:
            batch.append(idx)
            if len(batch) == self.batch_size:
                if i >= self.start_iter * self.batch_size:
                    lo = self.rank * self.batch_size // self.world_size
                    hi = (self.rank + 1) * self.batch_size // self.world_size
                    yield batch[lo:hi]
                    self.start_iter = 0
                i += len(batch)
                batch = []

    def __len__(self):
        return len(self.sampler) // self.batch_size


def make_data_iterator(args):
    """configure_data.py: ConcatDataset of the given files -> RandomMappingDataset (x200 virtual length, index-seeded
    shuffle) -> global batches sliced by data-parallel rank; resumes at args.iteration."""
    sets = [get_dataset_by_type(args.dataset_type, path, args) for path in args.train_data]
    ds = RandomMappingDataset(torch.utils.data.ConcatDataset(sets))
    dp_world, dp_rank = mpu.get_data_parallel_world_size(), mpu.get_data_parallel_rank()
    sampler = DistributedBatchSampler(torch.utils.data.SequentialSampler(ds), args.batch_size * dp_world, dp_rank, dp_world,
                                      start_iter=args.iteration)
    loader = torch.utils.data.DataLoader(ds, batch_sampler=sampler, num_workers=args.num_workers, pin_memory=True)
    return iter(loader)


def get_batch(data_iterator, args):
    """pretrain_gpt2.py:256-289 (the model-parallel broadcast of the batch included)."""
    keys = ['text', 'loss_mask']
    data = next(data_iterator) if data_iterator is not None and mpu.get_model_parallel_rank() == 0 else None
    data_b = mpu.broadcast_data(keys, data, torch.int64)
    return training.get_batch(data_b['text'].cuda(non_blocking=True), data_b['loss_mask'].cuda(non_blocking=True))


def train(model, optimizer, lr_scheduler, train_data_iterator, args):
    """pretrain_gpt2.py:483-566."""
    model.train()
    total = total_img = total_txt = 0.0
    skipped_iters, t0 = 0, time.time()
    half = args.fp16 or args.bf16
    while args.iteration < args.train_iters:
        batch = get_batch(train_data_iterator, args)
        log = (args.iteration + 1) % args.log_interval == 0
        # the reference's train_step (:406-448): forward, the nan / inf guard on the all-reduced partial losses,
        # backward + gradient exchange + clip, optimizer step; the logged loss is the mean over all ranks (:361-365)
        lm_loss, _, img_loss, txt_loss = training.forward_step(batch, model, args.txt_loss_scale, args.is_sparse, log=True,
                                                               world_size=args.world_size)
        forward_ok = bool(torch.isfinite(img_loss + txt_loss).all().item())
        if not forward_ok:
            # the reference's train_step returns early (:414-416) but its train loop still counts, logs, saves and
            # honours exit_interval for that iteration (:505-566)
            print('Skipping backward and optimizer step for nan or inf in forwarding!')
            if hasattr(model, 'needs_reduction'):
                model.needs_reduction = False
            skipped_iters += 1
            lm_loss = (img_loss + txt_loss).detach()
        else:
            lm_loss = training.backward_step(optimizer, model, lm_loss, args.clip_grad, half, world_size=args.world_size,
                                             reduce_loss=True)
            optimizer.step()
            if half and optimizer.overflow:
                skipped_iters += 1
            else:
                lr_scheduler.step()
        args.iteration += 1
        total += lm_loss.detach().float().view(())
        if log:
            total_img, total_txt = img_loss.float().item(), txt_loss.float().item()
            dt = (time.time() - t0) * 1000.0 / args.log_interval
            tok = args.batch_size * (args.max_position_embeddings - 1) * mpu.get_data_parallel_world_size() / (dt / 1000.0)
            s = ' iteration {:8d}/{:8d} | elapsed time per iteration (ms): {:.1f} | tokens/s {:.0f} | learning rate {:.3E}'.format(
                args.iteration, args.train_iters, dt, tok, optimizer.param_groups[0]['lr'])
            s += ' | lm loss {:.6E} | img loss {:.6E} | txt loss {:.6E}'.format(total.item() / args.log_interval, total_img, total_txt)
            if half:
                s += ' | loss scale {:.1f} | skipped {}'.format(optimizer.loss_scale, skipped_iters)
            utils.print_rank_0(s)
            total, t0 = 0.0, time.time()
        if args.save and args.save_interval and args.iteration % args.save_interval == 0:
            utils.save_checkpoint(args.iteration, model, optimizer, lr_scheduler, args)
        if args.exit_interval and args.iteration % args.exit_interval == 0:
            torch.distributed.barrier()
            utils.print_rank_0('exiting the program at iteration {}'.format(args.iteration))
            return args.iteration, skipped_iters
    return args.iteration, skipped_iters


def main(argv=None):
    """pretrain_gpt2.py:697-819."""
    args = get_args(argv)
    initialize_distributed(args)
    set_random_seed(args.seed)
    model, optimizer, lr_scheduler = setup_model_and_optimizer(args)
    if args.load is not None:
        args.iteration = utils.load_checkpoint(model, optimizer, lr_scheduler, args)
    iterator = make_data_iterator(args) if args.train_data is not None and mpu.get_model_parallel_rank() == 0 else None
    if args.train_data is not None and args.train_iters > 0:
        iteration, skipped = train(model, optimizer, lr_scheduler, iterator, args)
        if args.save and iteration % max(1, args.save_interval) != 0:
            utils.save_checkpoint(iteration, model, optimizer, lr_scheduler, args)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
