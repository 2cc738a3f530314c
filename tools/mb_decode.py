"""Single-token decode latency of the 4B model with a 1024-position memory: the captured HIP-graph decode step
(generation.GraphDecoder), eager K/V-cache memories (kv_cache=True) and the reference-style layer-input memories (every
step re-projects K and V of the whole memory)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29589")
import torch, torch.distributed as dist
dist.init_process_group("nccl", init_method="env://", world_size=1, rank=0)
from cogview_amd import mpu
from cogview_amd.fp16 import FP16_Module
from cogview_amd.model import GPT2Model
mpu.initialize_model_parallel(1); torch.manual_seed(1); mpu.model_parallel_cuda_manual_seed(1)
L, h, heads, V = 48, 2560, 40, 58240
pre, steps = 1024, 24
tokens = torch.randint(0, 58219, (1, pre + steps), device="cuda")
pos = torch.arange(pre + steps, device="cuda").unsqueeze(0)
from cogview_amd.generation import GraphDecoder
model = FP16_Module(GPT2Model(L, V, h, heads, 0.1, 0.1, 0.1, 1089, 1089, False).cuda(), dtype=torch.bfloat16, keep_half_outputs=True).eval()
B = int(os.environ.get("MB_DECODE_BATCH", "1"))
tokens, pos = tokens.expand(B, -1).contiguous(), pos.expand(B, -1).contiguous()
dec = GraphDecoder(model, batch=B, capacity=1152)
with torch.no_grad():
    dec.prefill(tokens[:, :pre], pos[:, :pre])
    for mode in ("eager fixed-capacity step", "captured graph"):
        if mode == "captured graph":
            dec.capture()
        for t in range(4):
            dec.step(tokens[:, pre:pre + 1], pos[:, pre:pre + 1])
        torch.cuda.synchronize(); t0 = time.time()
        for t in range(20):
            lg = dec.step(tokens[:, pre:pre + 1], pos[:, pre:pre + 1])
            nxt = lg[:, -1].float().argmax(-1)                   # consume the logits on the device, as a sampler would
        torch.cuda.synchronize(); dt = (time.time() - t0) / 20
        print(f"GraphDecoder {mode}: decode {dt*1e3:.2f} ms/token at memory length ~{pre}" + (f" (batch {B})" if B > 1 else ""), flush=True)
del model, dec
torch.cuda.empty_cache()
if os.environ.get("MB_DECODE_GRAPH_ONLY") == "1":
    sys.exit(0)
for kv in (True, False):
    model = FP16_Module(GPT2Model(L, V, h, heads, 0.1, 0.1, 0.1, 1089, 1089, False, kv_cache=kv).cuda(), dtype=torch.bfloat16,
                        keep_half_outputs=True).eval()
    with torch.no_grad():
        torch.cuda.synchronize(); t0 = time.time()
        logits, *mems = model(tokens[:, :pre], pos[:, :pre], 0, None, None, 0)
        torch.cuda.synchronize(); t_pre = time.time() - t0
        for t in range(pre, pre + 4):                       # warm-up steps
            logits, *mems = model(tokens[:, t:t + 1], pos[:, t:t + 1], 0, None, None, 0, *mems)
        torch.cuda.synchronize(); t0 = time.time()
        for t in range(pre + 4, pre + steps):
            logits, *mems = model(tokens[:, t:t + 1], pos[:, t:t + 1], 0, None, None, 0, *mems)
        torch.cuda.synchronize(); dt = (time.time() - t0) / (steps - 4)
    print(f"kv_cache={kv}: prefix of {pre} tokens {t_pre*1e3:.1f} ms; decode {dt*1e3:.2f} ms/token at memory length ~{pre}", flush=True)
    del model, mems, logits
    torch.cuda.empty_cache()
