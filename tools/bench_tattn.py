"""Fused text cross-attention launch (udt_tattn_fused) at the UNet's four levels, 8 samples (4 with zero context), in a
dependent chain inside a hipGraph (out of launch i is x of launch i + 1).   python tools/bench_tattn.py"""
import math, os, sys
sys.path.insert(0, os.environ.get("UDT_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
from udifftext_amd import ops, packing

dev = torch.device("cuda", 0)
torch.manual_seed(0)
for N, heads in [(4096, 5), (1024, 10), (256, 20), (64, 20)]:
    B, Lc, Dc, C = 8, 12, 2048, heads * 64
    x = [(torch.randn((B, N, C), device=dev) * 1.5).bfloat16() for _ in range(2)]
    ctx = torch.randn((B, Lc, Dc), device=dev).bfloat16()
    kvw = packing.pack_linear(torch.randn((2 * C, Dc), device=dev) / math.sqrt(Dc))
    kv = ops.linear(ctx.reshape(B * Lc, Dc), kvw).reshape(B, Lc, 2 * C)
    tabs = ops.tattn_prepare(kv, packing.pack_linear(torch.randn((C, C), device=dev) / math.sqrt(C)),
                             packing.pack_linear(torch.randn((C, C), device=dev) / math.sqrt(C)), torch.ones(C, device=dev),
                             torch.zeros(C, device=dev), heads, 64 ** -0.5)
    bo = torch.zeros(C, device=dev)
    fns = [(lambda i=i: ops.tattn_fused(x[i % 2], tabs, bo, heads, 4, 1e-5, out=x[(i + 1) % 2])) for i in range(20)]
    for f in fns[:2]:
        f()
    torch.cuda.synchronize()
    s = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for f in fns:
            f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"tattn_fused B=8 n={N:5d} C={C:5d}: {e0.elapsed_time(e1) / 100 * 1e3:7.1f} us", flush=True)
