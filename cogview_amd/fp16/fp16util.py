"""Helpers kept from fp16/fp16util.py that the reference's callers use (prep_param_lists & friends)."""
import torch
from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors

from .. import mpu


def prep_param_lists(model, flat_master=False):
    model_params = [p for p in model.parameters() if p.requires_grad]
    if flat_master:
        master = torch.nn.Parameter(_flatten_dense_tensors([p.data for p in model_params]).float())
        master.requires_grad = True
        if master.grad is None:
            master.grad = master.new(*master.size())
        return model_params, [master]
    masters = [p.clone().float().detach() for p in model_params]
    for p in masters:
        p.requires_grad = True
    return model_params, masters


def model_grads_to_master_grads(model_params, master_params, flat_master=False):
    if flat_master:
        master_params[0].grad.data.copy_(_flatten_dense_tensors([p.grad.data for p in model_params]))
        return
    for model, master in zip(model_params, master_params):
        if model.grad is not None:
            if master.grad is None:
                master.grad = master.data.new(*master.data.size())
            master.grad.data.copy_(model.grad.data)
        else:
            master.grad = None


def master_params_to_model_params(model_params, master_params, flat_master=False):
    if flat_master:
        for model, master in zip(model_params, _unflatten_dense_tensors(master_params[0].data, model_params)):
            model.data.copy_(master)
    else:
        for model, master in zip(model_params, master_params):
            model.data.copy_(master.data)


def to_python_float(t):
    return t.item() if hasattr(t, 'item') else t[0]


clip_grad_norm = mpu.clip_grad_norm
