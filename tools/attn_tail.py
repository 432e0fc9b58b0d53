"""the two self-attention kernels at N = 4096 for head counts that fill the chip a whole number of times or not (workgroups = 32 * B * heads;
1024 co-resident at 4 per CU): microseconds per launch and per 1024 workgroups."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
from udifftext_amd import ops
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
def t(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
for B, H in ((8, 2), (8, 3), (8, 4), (8, 5), (8, 6), (8, 8), (8, 10), (8, 12)):
    C = H * 64
    qkv = torch.randn((B, N, 3 * C), device=dev).bfloat16()
    data = (torch.randn((B * N, 3 * C), device=dev) * 8).to(torch.float8_e4m3fn).view(torch.uint8)
    scale = torch.full(((3 * C + 127) // 128, B * N), 0x7f7f7f7f, dtype=torch.int32, device=dev)
    o = torch.empty((B, N, C), device=dev, dtype=torch.bfloat16)
    tb = t(lambda: ops.attention_rowv(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], H, 0.125, out=o))
    t8 = t(lambda: ops.attention_mx8(ops.Mx8Act(data, scale), B, H, 0.125, 32.0, out=o))
    wgs = (N // 128) * B * H
    print(f"B={B} H={H:2d}: {wgs:5d} workgroups  bf16 {tb:7.1f} us ({tb * 1024 / wgs:6.1f} per 1024)   e4m3 {t8:7.1f} us ({t8 * 1024 / wgs:6.1f} per 1024)")
