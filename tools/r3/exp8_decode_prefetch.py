"""Captured 4B decode step with and without the side-stream cache warming of functional.decode_chain
(COGV_DECODE_PREFETCH), same process, same model, same cache state.
HISTORICAL: the code it drives (cogv_prefetch, functional._Warm) exists only in commit 6d23bbd -- the measurement
(profiles/r03_decode_prefetch_ab.log) was 2-3.6x SLOWER and the code was reverted."""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29591")
import torch, torch.distributed as dist
dist.init_process_group("nccl", init_method="env://", world_size=1, rank=0)
from cogview_amd import mpu, functional as F_
from cogview_amd.fp16 import FP16_Module
from cogview_amd.model import GPT2Model
from cogview_amd.generation import GraphDecoder
mpu.initialize_model_parallel(1); torch.manual_seed(1); mpu.model_parallel_cuda_manual_seed(1)
L, h, heads, V = 48, 2560, 40, 58240
pre = 1024
tokens = torch.randint(0, 58219, (1, pre + 1), device="cuda")
pos = torch.arange(pre + 1, device="cuda").unsqueeze(0)
model = FP16_Module(GPT2Model(L, V, h, heads, 0.1, 0.1, 0.1, 1089, 1089, False).cuda(), dtype=torch.bfloat16, keep_half_outputs=True).eval()
dec = GraphDecoder(model, batch=1, capacity=1152)
base = None
with torch.no_grad():
    dec.prefill(tokens[:, :pre], pos[:, :pre])
    for wgs in [0, 32, 0, 16, 64, 128]:
        try:
            F_._DECODE_PREFETCH = wgs
            dec.length = pre
            dec.graph = None
            dec.capture()
            dec.length = pre
            first = dec.step(tokens[:, pre:], pos[:, pre:]).float().clone()
            for _ in range(3):
                dec.length = pre
                dec.step(tokens[:, pre:], pos[:, pre:])
            torch.cuda.synchronize(); t0 = time.time()
            for _ in range(30):
                dec.length = pre
                lg = dec.step(tokens[:, pre:], pos[:, pre:])
                nxt = lg[:, -1].float().argmax(-1)
            torch.cuda.synchronize(); dt = (time.time() - t0) / 30
            if base is None:
                base = first
            print(f"prefetch workgroups {wgs:4d}: captured decode {dt*1e3:.3f} ms/token; logits equal to the first run: {torch.equal(first, base)}", flush=True)
        except Exception:
            traceback.print_exc()
            print(f"prefetch workgroups {wgs}: FAILED", flush=True)
