"""Micro-benchmark of the hot kernels at the UNet's shapes (config #2: 8 samples in flight)."""
import math, os, sys, time
sys.path.insert(0, os.environ.get("UDT_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
from udifftext_amd import ops, packing, lib as L

dev = torch.device("cuda", 0)

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

def conv_case(B, H, C1, C2, N, ups=False, k=3):
    x = torch.randn((B, H, H, C1), device=dev).bfloat16()
    x2 = torch.randn((B, H, H, C2), device=dev).bfloat16() if C2 else None
    w = packing.pack_conv(torch.randn((N, C1 + C2, k, k), device=dev) / math.sqrt((C1 + C2) * k * k), [C1, C2] if C2 else None)
    b = torch.zeros((w.shape[0],), device=dev)
    Ho = H * 2 if ups else H
    out = torch.empty((B, Ho, Ho, w.shape[0]), dtype=torch.bfloat16, device=dev)
    ms = timeit(lambda: ops.conv2d(x, w, b, ksize=k, upsample=ups, x2=x2, out=out, n_out=w.shape[0]))
    fl = 2.0 * B * Ho * Ho * N * (C1 + C2) * k * k
    print(f"conv{k}x{k} B{B} {H}x{H} {C1}+{C2}->{N} ups={ups}: {ms*1e3:8.1f} us  {fl/ms/1e9:8.1f} TF/s")

def lin_case(M, N, K, flags=0):
    x = torch.randn((M, K), device=dev).bfloat16()
    w = packing.pack_linear(torch.randn((N, K), device=dev) / math.sqrt(K))
    b = torch.zeros((N,), device=dev)
    ncols = N // 2 if flags & L.GEMM_GEGLU else N
    out = torch.empty((M, ncols), dtype=torch.bfloat16, device=dev)
    ms = timeit(lambda: ops.linear(x, w, b, out=out, flags=flags))
    print(f"linear {M}x{N}x{K} flags={flags}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TF/s")

def attn_case(B, H, N):
    C = H * 64
    qkv = torch.randn((B, N, 3 * C), device=dev).bfloat16()
    vt = torch.randn((B, C, N), device=dev).bfloat16()
    out = torch.empty((B, N, C), dtype=torch.bfloat16, device=dev)
    ms = timeit(lambda: ops.attention(qkv[..., :C], qkv[..., C:2*C], vt, H, 0.125, out=out))
    print(f"attn B{B} H{H} N{N}: {ms*1e3:8.1f} us  {4.0*B*H*N*N*64/ms/1e9:8.1f} TF/s")
    ms = timeit(lambda: ops.attention_rowv(qkv[..., :C], qkv[..., C:2*C], qkv[..., 2*C:], H, 0.125, out=out))
    print(f"attn (row-major V) B{B} H{H} N{N}: {ms*1e3:8.1f} us  {4.0*B*H*N*N*64/ms/1e9:8.1f} TF/s")

B = 8
conv_case(B, 64, 320, 0, 320)
conv_case(B, 64, 640, 320, 320)
conv_case(B, 32, 640, 0, 640)
conv_case(B, 32, 1280, 640, 640)
conv_case(B, 16, 1280, 0, 1280)
conv_case(B, 16, 1280, 1280, 1280)
conv_case(B, 8, 1280, 0, 1280)
conv_case(B, 8, 1280, 1280, 1280)
conv_case(B, 32, 640, 0, 640, ups=True)
conv_case(B, 64, 64, 0, 320)
conv_case(B, 64, 320, 0, 4)
lin_case(B * 4096, 320, 320)
lin_case(B * 4096, 640, 320)
lin_case(B * 4096, 2560, 320, L.GEMM_GEGLU)
lin_case(B * 4096, 320, 1280)
lin_case(B * 1024, 640, 640)
lin_case(B * 1024, 5120, 640, L.GEMM_GEGLU)
lin_case(B * 1024, 640, 2560)
lin_case(B * 256, 1280, 1280)
lin_case(B * 256, 10240, 1280, L.GEMM_GEGLU)
lin_case(B * 256, 1280, 5120)
lin_case(B * 64, 1280, 1280)
lin_case(4096, 4096, 4096)
lin_case(8192, 8192, 8192)
attn_case(B, 5, 4096)
attn_case(B, 10, 1024)
attn_case(B, 20, 256)
attn_case(B, 20, 64)
# VAE-ish
conv_case(4, 512, 128, 0, 128)
conv_case(4, 256, 256, 0, 256)
conv_case(4, 128, 512, 0, 512)
# norms
x = torch.randn((B, 4096, 320), device=dev).bfloat16(); g = torch.ones(320, device=dev); bb = torch.zeros(320, device=dev)
o = torch.empty_like(x)
ms = timeit(lambda: ops.group_norm(x, g, bb, 32, 1e-5, True, out=o)); print(f"gn+silu 8x4096x320: {ms*1e3:.1f} us  {x.numel()*2*3/ms/1e6:.1f} GB/s (3 passes)")
ms = timeit(lambda: ops.layer_norm(x, g, bb, 1e-5, out=o)); print(f"layernorm 32768x320: {ms*1e3:.1f} us  {x.numel()*2*2/ms/1e6:.1f} GB/s")
