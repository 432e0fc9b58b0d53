"""One 3x3 convolution / GEMM alone on the chip, planned for 1/s of the CUs (udt_gemm_desc.cu_share = s): how much of a
launch's time scales with the workgroup count.  usage: python tools/bench_cu_share.py"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from udifftext_amd import ops, packing

dev = torch.device("cuda", 0)


def timed(fn, iters=40):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


print("conv B H C->N         " + "".join(f"share={s:<12d}" for s in (1, 2, 3, 4, 8)))
for B, H, C, N in [(8, 64, 320, 320), (8, 64, 640, 320), (8, 32, 640, 640), (8, 32, 1280, 640), (8, 16, 1280, 1280),
                   (8, 16, 2560, 1280), (8, 8, 1280, 1280), (1, 512, 128, 128), (1, 256, 256, 256)]:
    x = torch.randn((B, H, H, C), device=dev).bfloat16()
    w = packing.pack_conv(torch.randn((N, C, 3, 3), device=dev) / math.sqrt(C * 9))
    b = torch.zeros((N,), device=dev)
    out = torch.empty((B, H, H, N), dtype=torch.bfloat16, device=dev)
    row = f"{B:2d} {H:3d} {C:4d}->{N:4d}    "
    for s in (1, 2, 3, 4, 8):
        with ops.launch_context(cu_share=s):
            us = timed(lambda: ops.conv2d(x, w, b, out=out))
        row += f"{us:6.1f} us {2.0 * B * H * H * N * C * 9 / us / 1e6:5.0f}TF "
    print(row, flush=True)
print("gemm M N K")
for M, N, K in [(32768, 320, 320), (32768, 640, 320), (32768, 320, 1280), (8192, 640, 2560), (2048, 1280, 5120), (2048, 2560, 1280)]:
    x = torch.randn((M, K), device=dev).bfloat16()
    w = packing.pack_linear(torch.randn((N, K), device=dev) / math.sqrt(K))
    b = torch.zeros((N,), device=dev)
    row = f"{M:6d} {N:5d} {K:5d}   "
    for s in (1, 2, 3, 4, 8):
        with ops.launch_context(cu_share=s):
            us = timed(lambda: ops.linear(x, w, b))
        row += f"{us:6.1f} us {2.0 * M * N * K / us / 1e6:5.0f}TF "
    print(row, flush=True)
