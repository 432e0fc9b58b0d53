"""LayerNorm-folded projections of the 64 x 64 level (K = 320): the row-resident kernel (rowres.h) against the tiled lean kernels
on the same packed weights.   python tools/bench_rowres.py"""
import math, os, sys
sys.path.insert(0, os.environ.get("UDT_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
from udifftext_amd import lib as L, ops, packing

dev = torch.device("cuda", 0)
lib = L.load()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"{'shape':34s} {'tiled (lean)':>22s} {'row-resident':>22s}")
for M, N, K, geglu in [(32768, 2560, 320, True), (32768, 960, 320, False), (65536, 2560, 320, True), (65536, 960, 320, False),
                       (16384, 2560, 320, True), (16384, 960, 320, False), (8192, 2560, 320, True)]:
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn((M, K), generator=g).to(dev).bfloat16()
    w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(dev)
    b = torch.randn((N,), generator=g).to(dev)
    gamma, beta = torch.ones((K,), device=dev), torch.zeros((K,), device=dev)
    wf, cf, sf = packing.pack_ln_linear(w, b, gamma, beta, geglu=geglu)
    fl = L.GEMM_GEGLU if geglu else 0
    out = torch.empty((M, N // 2 if geglu else N), dtype=torch.bfloat16, device=dev)
    res = []
    for on in (0, 1):
        L.check(lib.udt_debug_set(b"rowres", on), "debug_set")
        res.append(timed(lambda: ops.ln_linear(x, wf, cf, sf, flags=fl, out=out)))
    L.check(lib.udt_debug_set(b"rowres", 1), "debug_set")
    fl_ = 2.0 * M * N * K
    print(f"{M:6d} x {N:5d} x {K:4d} {'geglu' if geglu else 'plain'}  " + "  ".join(f"{t:9.1f} us {fl_ / t / 1e6:6.0f} TF" for t in res), flush=True)
