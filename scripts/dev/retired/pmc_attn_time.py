"""Dev: a few launches of both attn_time_bwd variants for rocprofv3 --pmc passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oa-transformer_amd")); sys.path.insert(0, ROOT)
import torch
from OATrans.ops import hip
B, T, N, H = 32, 8, 196, 12
D = H * 64; M = B * T * N + B; Mp = (M + 255) // 256 * 256
qkv = torch.zeros(Mp, 3 * D, dtype=torch.bfloat16, device="cuda"); qkv[:M] = torch.randn(M, 3 * D, device="cuda").bfloat16()
out = torch.zeros(Mp, D, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(Mp, H, device="cuda")
dout = torch.zeros(Mp, D, dtype=torch.bfloat16, device="cuda"); dout[:M] = torch.randn(M, D, device="cuda").bfloat16()
dqkv = torch.zeros(Mp, 3 * D, dtype=torch.bfloat16, device="cuda"); side = torch.zeros(B, H, 3, 64, device="cuda")
sc = 0.125
hip.attn_cls_fwd(qkv, out, lse, B, T, N, H, D, sc); hip.attn_time_fwd(qkv, out, lse, B, T, N, H, D, sc)
for var in (1, 0):
    hip.lib().oat_attn_time_set_variant(var)
    for _ in range(3):
        hip.attn_time_bwd(qkv, out, lse, dout, dqkv, side, B, T, N, H, D, sc)
torch.cuda.synchronize()
