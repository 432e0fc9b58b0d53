import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
from udifftext_amd import ops, packing
from udifftext_amd import lib as L
for item in os.environ.get("DBG", "").split(","):
    if item:
        k, v = item.split("="); L.check(L.load().udt_debug_set(k.encode(), int(v)), "dbg")
dev = torch.device("cuda", 0)
B, H, C, N = [int(v) for v in sys.argv[1:5]]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 50
x = torch.randn((B, H, H, C), device=dev).bfloat16()
w = packing.pack_conv(torch.randn((N, C, 3, 3), device=dev) / math.sqrt(C * 9))
b = torch.zeros((N,), device=dev)
out = torch.empty((B, H, H, N), dtype=torch.bfloat16, device=dev)
for _ in range(2): ops.conv2d(x, w, b, out=out)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters): ops.conv2d(x, w, b, out=out)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / iters
print(f"dbg={os.environ.get('DBG','')} conv B{B} {H}x{H} {C}->{N}: {ms*1e3:.1f} us {2.0*B*H*H*N*C*9/ms/1e9:.1f} TF/s")
