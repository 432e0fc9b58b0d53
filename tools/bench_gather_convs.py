"""The UNet's gathered 3x3 convolutions (nearest x2 upsampling folded into the gather, stride 2) next to a staged-patch
convolution of the same FLOPs — how much the gather form (gemm8<CONV>) costs against conv3p."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from udifftext_amd import ops, packing

dev = torch.device("cuda", 0)


def timed(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for share in (1, 3):
    print(f"cu_share {share}")
    for B, Hin, C, N, mode in [(8, 32, 640, 640, "up"), (8, 16, 1280, 1280, "up"), (8, 8, 1280, 1280, "up"),
                               (8, 64, 320, 320, "s2"), (8, 32, 640, 640, "s2"), (8, 16, 1280, 1280, "s2")]:
        x = torch.randn((B, Hin, Hin, C), device=dev).bfloat16()
        w = packing.pack_conv(torch.randn((N, C, 3, 3), device=dev) / math.sqrt(C * 9))
        b = torch.zeros((N,), device=dev)
        with ops.launch_context(cu_share=share):
            if mode == "up":
                Ho = 2 * Hin
                us = timed(lambda: ops.conv2d(x, w, b, upsample=True))
                xr = torch.randn((B, Ho, Ho, C), device=dev).bfloat16()
                ref = timed(lambda: ops.conv2d(xr, w, b))
            else:
                Ho = Hin // 2
                us = timed(lambda: ops.conv2d(x, w, b, stride=2))
                xr = torch.randn((B, Ho, Ho, C), device=dev).bfloat16()
                ref = timed(lambda: ops.conv2d(xr, w, b))
        fl = 2.0 * B * Ho * Ho * N * C * 9
        print(f"  {mode} {B}x{Hin}x{Hin} {C}->{N} (out {Ho}x{Ho}): this launch {us:7.1f} us {fl/us/1e6:5.0f} TF | staged-patch conv of the same output {ref:7.1f} us {fl/ref/1e6:5.0f} TF", flush=True)
