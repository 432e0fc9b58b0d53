import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
from udifftext_amd import ops, packing
dev = torch.device("cuda", 0)
M, N, K = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (8192, 8192, 8192))]
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
x = torch.randn((M, K), device=dev).bfloat16()
w = packing.pack_linear(torch.randn((N, K), device=dev) / math.sqrt(K))
out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
for _ in range(2): ops.linear(x, w, out=out)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters): ops.linear(x, w, out=out)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / iters
print(f"gemm {M}x{N}x{K}: {ms*1e3:.1f} us {2.0*M*N*K/ms/1e9:.1f} TF/s")
