"""one convolution / GEMM / self-attention shape launched 20 times (for rocprofv3 --pmc passes):
python tools/pmc_one_conv.py conv B H C N [up] | gemm M N K [geglu] | attn B heads N"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
from udifftext_amd import lib as L, ops, packing
dev = torch.device("cuda", 0)
kind = sys.argv[1]
if kind == "conv":
    B, H, C, N = [int(v) for v in sys.argv[2:6]]
    up = len(sys.argv) > 6 and sys.argv[6] == "1"
    x = torch.randn((B, H, H, C), device=dev).bfloat16()
    w = packing.pack_conv(torch.randn((N, C, 3, 3), device=dev) / math.sqrt(C * 9))
    b = torch.zeros((N,), device=dev)
    for _ in range(20):
        ops.conv2d(x, w, b, upsample=up)
elif kind == "attn":
    B, Hh, N = [int(v) for v in sys.argv[2:5]]
    qkv = torch.randn((B, N, 3 * Hh * 64), device=dev).bfloat16()
    Cc = Hh * 64
    for _ in range(20):
        ops.attention_rowv(qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:], Hh, 0.125)
else:
    M, N, K = [int(v) for v in sys.argv[2:5]]
    g = len(sys.argv) > 5 and sys.argv[5] == "1"
    x = torch.randn((M, K), device=dev).bfloat16()
    w = torch.randn((N, K), device=dev) / math.sqrt(K)
    bb = torch.randn((N,), device=dev)
    wp, bp = packing.pack_geglu(w, bb) if g else (packing.pack_linear(w), bb)
    for _ in range(20):
        ops.linear(x, wp, bp, flags=(L.GEMM_GEGLU if g else 0))
torch.cuda.synchronize()
