"""wide.h wconv3_kernel (256 pixels x 160 channels per workgroup, one per CU) against the lean 128-pixel kernel on the UNet's
3x3 shapes: `indep` = 20 back-to-back launches on rotating buffers inside a hipGraph, `chain` = 20 DEPENDENT launches (each
reads the previous output or takes it as residual: what a network step looks like).   python tools/bench_wide_conv.py"""
import math, os, sys
sys.path.insert(0, os.environ.get("UDT_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
from udifftext_amd import lib as L, ops, packing

dev = torch.device("cuda", 0)
torch.manual_seed(0)


def dbg(k, v):
    L.check(L.load().udt_debug_set(k.encode(), int(v)), "udt_debug_set " + k)


def graph_time(fn_list, reps=5):
    for f in fn_list[:3]:
        f()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for f in fn_list:
            f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(fn_list)) * 1e3


SHAPES = [(8, 64, 320, 320), (8, 64, 640, 320), (8, 64, 960, 320), (8, 32, 640, 640), (8, 32, 1280, 640), (8, 32, 320, 640),
          (8, 16, 1280, 1280), (8, 16, 2560, 1280), (8, 16, 640, 1280), (16, 96, 320, 320), (16, 48, 640, 640), (2, 64, 320, 320),
          (4, 64, 320, 320)]
if len(sys.argv) > 1:
    SHAPES = SHAPES[:int(sys.argv[1])]
print(f"{'conv (B HxW C->N)':26s}" + "".join(f"{h:>24s}" for h in ("lean indep", "wide indep", "lean chain", "wide chain", "lean chain+stats", "wide chain+stats")))
for B, H, C, N in SHAPES:
    xs = [torch.randn((B, H, H, C), device=dev).bfloat16() for _ in range(4)]
    w = packing.pack_conv(torch.randn((N, C, 3, 3), device=dev) / math.sqrt(C * 9))
    b = torch.zeros((N,), device=dev)
    rv = torch.randn((B, N), device=dev)
    outs = [torch.empty((B, H, H, N), dtype=torch.bfloat16, device=dev) for _ in range(4)]
    fl = 2.0 * B * H * H * N * C * 9
    row = f"{B:2d} {H:3d}x{H:<3d} {C:4d}->{N:4d}   "
    cols = []
    for regime in ("indep", "chain", "chain+stats"):
        for mode in (0, 1):
            dbg("wide_conv", mode)
            st = regime.endswith("stats")
            if regime == "indep":
                fns = [(lambda i=i: ops.conv2d(xs[i % 4], w, b, out=outs[i % 4], rowvec=rv)) for i in range(20)]
            else:
                fns = [(lambda i=i: ops.conv2d(xs[0], w, b, residual=outs[i % 2], out=outs[(i + 1) % 2], rowvec=rv, colstats=st)) for i in range(20)]
            us = graph_time(fns)
            cols.append(f"{us:8.1f} us {fl / us / 1e6:5.0f} TF")
    print(row + "".join(f"{c:>24s}" for c in cols), flush=True)
dbg("wide_conv", -1)
