"""In-kernel phase times of the dK.dV attention kernel at the bench shape (probe build: COGV_VARIANT=ts COGV_HIPCC_EXTRA="-DCOGV_ATTN_TS"
python cogview_amd/csrc/build.py; run with COGVIEW_HIP_LIB=build/ab/libcogview_ts.so).  GPU box."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cogview_amd import _lib as L, ops
s, drop, H, bb = 1088, (0.1, 1, 2), 40, 24
dt = torch.float16
qkv = torch.randn(bb, s, 3 * H * 64, device="cuda", dtype=dt)
q, k, v = [qkv[:, :, i * H * 64:(i + 1) * H * 64].view(bb, s, H, 64) for i in range(3)]
do = torch.randn(bb, s, H, 64, device="cuda", dtype=dt)
dqkv = torch.empty_like(qkv)
outs = dict(dq=dqkv[:, :, :H * 64].view(bb, s, H, 64), dk=dqkv[:, :, H * 64:2 * H * 64].view(bb, s, H, 64), dv=dqkv[:, :, 2 * H * 64:].view(bb, s, H, 64))
cs = torch.zeros(3 * H * 64, device="cuda", dtype=dt)
o, lse, bits = ops.attention_fwd(q, k, v, dropout=drop, keep_bits=True)
run = lambda: ops.attention_bwd(do, q, k, v, o, lse, dropout=drop, keep_bits=bits, colsum_out=cs, **outs)
for _ in range(3):
    run()
torch.cuda.synchronize()
fn = L.lib().cogv_debug_attn_ts
fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 8)()
assert fn(buf, 1) == 0
n = 5
import time
t0 = time.time()
for _ in range(n):
    run()
torch.cuda.synchronize()
print(f"backward (D + dQ + dK.dV) with the probe build: {(time.time() - t0) / n * 1e6:.0f} us per call")
assert fn(buf, 0) == 0
t = [int(x) for x in buf]
waves, iters = t[7], t[6]
names = ["DMA wait + barrier", "DMA issue", "stage compute (both halves)", "epilogue", "prologue", "whole wave"]
print(f"waves {waves / n:.0f} per launch, wave-iterations {iters / n:.0f} per launch ({iters / waves:.2f} per wave)")
for i, nm in enumerate(names):
    print(f"  {nm:30s} {t[i] / waves:10.0f} cycles per wave   {100.0 * t[i] / t[5]:5.1f} % of the wave's life" + (f"   {t[i] / iters:8.0f} per iteration" if i < 3 else ""))
