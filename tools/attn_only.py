"""one shape of the two self-attention kernels, a few launches each (for rocprofv3 --pmc passes): python tools/attn_only.py [B N heads]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
from udifftext_amd import ops
B, N, H = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (8, 4096, 5)
dev = torch.device("cuda", 0)
C = H * 64
qkv = torch.randn((B, N, 3 * C), device=dev).bfloat16()
data = (torch.randn((B * N, 3 * C), device=dev) * 8).to(torch.float8_e4m3fn).view(torch.uint8)
scale = torch.full(((3 * C + 127) // 128, B * N), 0x7f7f7f7f, dtype=torch.int32, device=dev)
for _ in range(4):
    ops.attention_rowv(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], H, 0.125)
    ops.attention_mx8(ops.Mx8Act(data, scale), B, H, 0.125, 32.0)
torch.cuda.synchronize()
