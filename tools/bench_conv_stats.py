"""Cost of the statistics-emitting epilogue of the lean 3x3 convolution / GEMM: the same launch with and without
udt_gemm_desc.colstats, 20 launches on rotating buffers in one hipGraph.   python tools/bench_conv_stats.py"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
from udifftext_amd import ops, packing

dev = torch.device("cuda", 0)


def graph_time(fns, reps=5):
    for f in fns[:4]:
        f()
    torch.cuda.synchronize()
    s = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for f in fns:
            f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(fns)) * 1e3


for B, H, C, N in [(8, 64, 320, 320), (8, 32, 640, 640), (8, 16, 1280, 1280), (8, 8, 1280, 1280)]:
    xs = [torch.randn((B, H, H, C), device=dev).bfloat16() for _ in range(4)]
    w = packing.pack_conv(torch.randn((N, C, 3, 3), device=dev) / math.sqrt(9 * C))
    b = torch.zeros((N,), device=dev)
    res = [torch.randn((B, H, H, N), device=dev).bfloat16() for _ in range(4)]
    t = {}
    for st in (False, True):
        fns = [(lambda i=i: ops.conv2d(xs[i % 4], w, b, ksize=3, residual=res[i % 4], colstats=st)) for i in range(20)]
        t[st] = graph_time(fns)
    print(f"conv3x3 B{B} {H}x{H} {C}->{N}: plain {t[False]:7.1f} us   with statistics {t[True]:7.1f} us   (+{t[True] - t[False]:.1f})", flush=True)
for M, N, K, rpb in [(32768, 320, 320, 4096), (8192, 640, 640, 1024), (2048, 1280, 1280, 256)]:
    xs = [torch.randn((M, K), device=dev).bfloat16() for _ in range(4)]
    w = packing.pack_linear(torch.randn((N, K), device=dev) / math.sqrt(K))
    res = [torch.randn((M, N), device=dev).bfloat16() for _ in range(4)]
    t = {}
    for st in (False, True):
        fns = [(lambda i=i: ops.linear(xs[i % 4], w, None, residual=res[i % 4], rows_per_batch=rpb, colstats=st)) for i in range(20)]
        t[st] = graph_time(fns)
    print(f"linear {M}x{N}x{K}: plain {t[False]:7.1f} us   with statistics {t[True]:7.1f} us   (+{t[True] - t[False]:.1f})", flush=True)
