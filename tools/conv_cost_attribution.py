"""Cost attribution of the lean 3x3 convolution's main loop (measurement builds only: UDT_EXTRA_FLAGS=-DUDT_MEASURE):
time per launch with parts of the loop switched off (results are WRONG in those modes).
  bit 0: no weight-tile DMA   bit 1: no patch DMA   bit 2: no MFMA (fragment reads kept)   bit 3: no LDS fragment reads
python tools/conv_cost_attribution.py"""
import math, os, sys
sys.path.insert(0, os.environ.get("UDT_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import udifftext_amd
from udifftext_amd import lib as L, ops, packing

dev = torch.device("cuda", 0)


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


MODES = [(0, "full"), (1, "no W dma"), (2, "no patch dma"), (3, "no dma"), (4, "no mfma"), (8, "no lds reads"), (12, "dma only"),
         (11, "mfma only"), (7, "lds reads only"), (15, "barriers only")]
dbg("lean_conv", 1)
WIDE = len(sys.argv) > 1 and sys.argv[1] == "wide"          # the wide kernel (wide.h) instead of the lean one
dbg("wide_conv", 1 if WIDE else 0)
print(f"{'conv':26s}" + "".join(f"{n:>15s}" for _, n in MODES))
for B, H, C, N in ([(8, 64, 320, 320), (8, 64, 960, 320), (8, 32, 1280, 640)] if WIDE else
                   [(8, 32, 640, 640), (8, 64, 320, 320), (8, 16, 1280, 1280), (8, 8, 1280, 1280), (16, 32, 640, 640), (13, 32, 640, 640)]):
    xs = [torch.randn((B, H, H, C), device=dev).bfloat16() for _ in range(4)]
    w = packing.pack_conv(torch.randn((N, C, 3, 3), device=dev) / math.sqrt(C * 9))
    b = torch.zeros((N,), device=dev)
    outs = [torch.empty((B, H, H, N), dtype=torch.bfloat16, device=dev) for _ in range(4)]
    row = f"{B:2d} {H:3d}x{H:<3d} {C:4d}->{N:4d}  "
    for m, _ in MODES:
        dbg("lconv_dbg", m)
        fns = [(lambda i=i: ops.conv2d(xs[i % 4], w, b, out=outs[i % 4])) for i in range(20)]
        row += f"{graph_time(fns):12.1f} us"
    dbg("lconv_dbg", 0)
    print(row, flush=True)
