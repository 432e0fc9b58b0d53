"""time + check udt_gemm on a list of MxNxK shapes (plain linear, bf16).  UDT_GEMM_IMPL selects the kernel family."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
from udifftext_amd import ops, packing
dev = torch.device("cuda", 0)
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(4096, 4096, 4096), (8192, 8192, 8192)]
from udifftext_amd import lib as L
for item in os.environ.get("DBG", "").split(","):
    if item:
        k, v = item.split("="); L.check(L.load().udt_debug_set(k.encode(), int(v)), "dbg")
torch.manual_seed(0)
for (M, N, K) in shapes:
    x = torch.randn((M, K), device=dev).bfloat16()
    wt = (torch.randn((N, K), device=dev) / math.sqrt(K))
    w = packing.pack_linear(wt)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    bias = torch.randn((N,), device=dev) if os.environ.get("BIAS") else None
    res = torch.randn((M, N), device=dev).bfloat16() if os.environ.get("RES") else None
    for _ in range(3): ops.linear(x, w, bias, residual=res, out=out)
    torch.cuda.synchronize()
    rows = torch.randint(0, M, (64,), device=dev)
    ref = x[rows].float() @ wt.bfloat16().float().t()
    if bias is not None: ref = ref + bias
    if res is not None: ref = ref + res[rows].float()
    err = ((out[rows].float() - ref).norm() / ref.norm()).item()
    iters = max(5, int(2e12 / (2.0 * M * N * K)))
    iters = min(iters, 200)
    best = 1e9
    for rep in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters): ops.linear(x, w, bias, residual=res, out=out)
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters)
    print(f"dbg={os.environ.get('DBG','')} impl={os.environ.get('UDT_GEMM_IMPL','8')} gemm {M}x{N}x{K}: {best*1e3:8.1f} us {2.0*M*N*K/best/1e9:8.1f} TF/s  relerr {err:.2e}", flush=True)
