"""The UNet's 1x1 skip convolutions over a channel concat (two sources, gather kernel) against a plain GEMM over the
materialised concat — what reading two sources through the gather path costs."""
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


for share in (1, 3):
    print(f"cu_share {share}")
    for B, H, C1, C2, N in [(8, 64, 320, 320, 320), (8, 64, 640, 320, 320), (8, 32, 640, 320, 640), (8, 32, 640, 640, 640),
                            (8, 32, 1280, 640, 640), (8, 16, 1280, 640, 1280), (8, 16, 1280, 1280, 1280), (8, 8, 1280, 1280, 1280)]:
        x1 = torch.randn((B, H, H, C1), device=dev).bfloat16()
        x2 = torch.randn((B, H, H, C2), device=dev).bfloat16()
        w = torch.randn((N, C1 + C2, 1, 1), device=dev) / math.sqrt(C1 + C2)
        wp = packing.pack_conv(w, [C1, C2])
        wl = packing.pack_linear(w.reshape(N, C1 + C2))
        b = torch.zeros((N,), device=dev)
        w1, w2 = wl[:, :C1].contiguous(), wl[:, C1:].contiguous()
        xc = torch.cat([x1, x2], 3).reshape(-1, C1 + C2).contiguous()
        res = torch.randn((B * H * H, N), device=dev).bfloat16()
        with ops.launch_context(cu_share=share):
            t_conv = timed(lambda: ops.conv2d(x1, wp, b, ksize=1, pad=(0, 0), x2=x2))
            t_lin = timed(lambda: ops.linear(xc, wl, b))
            t_two = timed(lambda: ops.linear(x2.reshape(-1, C2), w2, None, residual=ops.linear(x1.reshape(-1, C1), w1, b)))
        print(f"  {B}x{H}x{H} {C1}+{C2}->{N}: two-source gather {t_conv:6.1f} us | plain GEMM on the concat {t_lin:6.1f} us | two chained plain GEMMs {t_two:6.1f} us", flush=True)
