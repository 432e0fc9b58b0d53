"""3x3 convolutions on the 8 x 8 maps per split-K setting (library under UDT_ROOT).   python tools/bench_conv8.py"""
import math, os, sys
sys.path.insert(0, os.environ.get("UDT_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
from udifftext_amd import lib as L, ops, packing

dev = torch.device("cuda", 0)
lib = L.load()


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for B, C, N in [(8, 1280, 1280), (8, 2560, 1280), (16, 1280, 1280)]:
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn((B, 8, 8, C), generator=g).to(dev).bfloat16()
    w = packing.pack_conv((torch.randn((N, C, 3, 3), generator=g) / math.sqrt(9 * C)).to(dev))
    b = torch.randn((N,), generator=g).to(dev)
    rv = torch.randn((B, N), generator=g).to(dev)
    out = torch.empty((B, 8, 8, N), dtype=torch.bfloat16, device=dev)
    row = []
    for sk in (-1, 2, 4, 5, 7, 10):
        L.check(lib.udt_debug_set(b"lean_splitk", sk), "debug_set")
        t = timed(lambda: ops.conv2d(x, w, b, ksize=3, rowvec=rv, out=out, colstats=True))
        row.append(f"sk={sk:2d}: {t:6.1f}")
    L.check(lib.udt_debug_set(b"lean_splitk", -1), "debug_set")
    print(f"B={B} 8x8 {C}->{N}   " + "   ".join(row), flush=True)
