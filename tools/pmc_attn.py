import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cogview_amd import ops
b, H, s = 24, 40, 1088
qkv = torch.randn(b, s, 3 * H * 64, device="cuda", dtype=torch.bfloat16)
q, k, v = [qkv[:, :, i * H * 64:(i + 1) * H * 64].view(b, s, H, 64) for i in range(3)]
do = torch.randn(b, s, H, 64, device="cuda", dtype=torch.bfloat16)
drop = (0.1, 1, 2)
for _ in range(2):
    o, lse = ops.attention_fwd(q, k, v, dropout=drop)
    ops.attention_bwd(do, q, k, v, o, lse, dropout=drop)
torch.cuda.synchronize()
