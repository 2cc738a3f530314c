"""One call that binds the reference's scripts to this package: the reference imports `mpu`, `model`,
`fp16`, `vqvae` and `apex.optimizers.FusedAdam` by those top-level names (pretrain_gpt2.py:34-43, generate_samples.py:36-43,
utils.py:25-27, data_utils/vqvae_tokenizer.py:29); `bind_reference_names()` registers this package's mirrors under them.

Beyond the four aliases it also
  * registers every sub-module under the reference's dotted name (`fp16.loss_scaler`, `mpu.layers`, ...): `from mpu.layers import
    ...` style imports and, more importantly, UNPICKLING resolve to the mirror instead of importing a second copy of the file;
  * lets the loss scalers pickle under the reference's path `fp16.loss_scaler.{LossScaler, DynamicLossScaler}`:
    FP16_Optimizer.state_dict() carries the scaler OBJECT (fp16/fp16.py:336-360 -- so does the mirror, for the file format's
    sake), utils.save_checkpoint pickles it, and a checkpoint written here must open in the reference (and the reverse) without
    this package on the path.  The attribute names are the reference's (tests/test_reference_drivers_cpu.py loads files both ways).
"""
import importlib
import pkgutil
import sys
import types


def bind_reference_names(apex=True):
    from . import fp16, model, mpu, optim, vqvae
    for name, pkg in (("mpu", mpu), ("model", model), ("fp16", fp16), ("vqvae", vqvae)):
        sys.modules[name] = pkg
        for info in pkgutil.iter_modules(pkg.__path__):
            sys.modules[name + "." + info.name] = importlib.import_module(pkg.__name__ + "." + info.name)
    for cls in (fp16.loss_scaler.LossScaler, fp16.loss_scaler.DynamicLossScaler):
        cls.__module__ = "fp16.loss_scaler"
    if apex:
        a = sys.modules.get("apex") or types.ModuleType("apex")
        o = types.ModuleType("apex.optimizers")
        o.FusedAdam = optim.FusedAdam                     # pretrain_gpt2.py:43
        a.optimizers = o
        sys.modules["apex"], sys.modules["apex.optimizers"] = a, o
    return {"mpu": mpu, "model": model, "fp16": fp16, "vqvae": vqvae}
