"""run every lean kernel shape of the B=2 512x512 path several times and compare bitwise"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
from udifftext_amd import lib as L, ops, packing
dev = torch.device("cuda", 0)
torch.manual_seed(0)
convs = [(4, 64, 320, 320, 0), (4, 64, 640, 320, 0), (4, 64, 960, 320, 0), (4, 32, 320, 640, 0), (4, 32, 640, 640, 0), (4, 32, 1280, 640, 0), (4, 32, 960, 640, 0),
         (4, 16, 640, 1280, 0), (4, 16, 1280, 1280, 0), (4, 16, 2560, 1280, 0), (4, 16, 1920, 1280, 0), (4, 8, 1280, 1280, 0), (4, 8, 2560, 1280, 0),
         (4, 8, 1280, 1280, 1), (4, 16, 1280, 1280, 1), (4, 32, 640, 640, 1),
         (2, 512, 128, 128, 0), (2, 256, 128, 256, 0), (2, 256, 256, 256, 0), (2, 128, 256, 512, 0), (2, 128, 512, 512, 0), (2, 64, 512, 512, 0),
         (2, 64, 512, 512, 1), (2, 128, 512, 512, 1), (2, 256, 256, 256, 1)]
bad = 0
for B, H, C, N, up in convs:
    x = torch.randn((B, H, H, C), device=dev).bfloat16()
    w = packing.pack_conv(torch.randn((N, C, 3, 3), device=dev) / math.sqrt(C * 9))
    b = torch.randn((N,), device=dev)
    Ho = 2 * H if up else H
    r = torch.randn((B, Ho, Ho, N), device=dev).bfloat16()
    outs = [ops.conv2d(x, w, b, residual=r, upsample=bool(up)).clone() for _ in range(6)]
    torch.cuda.synchronize()
    nd = sum(int(not torch.equal(outs[0], o)) for o in outs[1:])
    mx = max((outs[0].float() - o.float()).abs().max().item() for o in outs[1:])
    bad += nd
    print(f"conv B{B} {H}x{H} {C}->{N} up={up}: {nd} of 5 repeats differ (max {mx:.3e})", flush=True)
for M, N, K, geglu in [(16384, 960, 320, 0), (16384, 2560, 320, 1), (16384, 320, 1280, 0), (4096, 640, 2560, 0), (1024, 1280, 5120, 0), (256, 1280, 5120, 0),
                       (256, 1280, 1280, 0), (1024, 3840, 1280, 0), (8192, 512, 512, 0)]:
    x = torch.randn((M, K), device=dev).bfloat16()
    w = torch.randn((N, K), device=dev) / math.sqrt(K)
    b = torch.randn((N,), device=dev)
    if geglu:
        wp, bp = packing.pack_geglu(w, b)
    else:
        wp, bp = packing.pack_linear(w), b
    outs = [ops.linear(x, wp, bp, flags=(L.GEMM_GEGLU if geglu else 0)).clone() for _ in range(6)]
    torch.cuda.synchronize()
    nd = sum(int(not torch.equal(outs[0], o)) for o in outs[1:])
    bad += nd
    print(f"gemm {M}x{N}x{K} geglu={geglu}: {nd} of 5 repeats differ", flush=True)
print("NONDETERMINISTIC" if bad else "all deterministic")
