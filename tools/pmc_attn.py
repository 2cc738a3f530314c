import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cogview_amd import ops
DT = torch.float16 if os.environ.get("PMC_ATTN_DTYPE", "bf16") == "fp16" else torch.bfloat16
b, H, s = 24, 40, 1088
qkv = torch.randn(b, s, 3 * H * 64, device="cuda", dtype=DT)
q, k, v = [qkv[:, :, i * H * 64:(i + 1) * H * 64].view(b, s, H, 64) for i in range(3)]
do = torch.randn(b, s, H, 64, device="cuda", dtype=DT)
drop = (0.1, 1, 2)
for _ in range(2):
    o, lse, bits = ops.attention_fwd(q, k, v, dropout=drop, keep_bits=True)       # the training path: stored keep bits
    ops.attention_bwd(do, q, k, v, o, lse, dropout=drop, keep_bits=bits)
torch.cuda.synchronize()
