from .fp16util import (clip_grad_norm, master_params_to_model_params, model_grads_to_master_grads,
                       prep_param_lists, to_python_float)
from .fp16 import FP16_Module, FP16_Optimizer, fp16_to_fp32, fp32_to_fp16
from .loss_scaler import DynamicLossScaler, LossScaler
