"""The UNet's plain-GEMM shapes at BASELINE config #2 (8 samples per call), timed back to back inside hipGraphs (20
launches of the same problem on ROTATING buffers, so every launch reads HBM-cold operands like the real step), for a
list of udt_debug_set settings:   python tools/bench_gemm_shapes.py n_block=-1 n_block=0 n_block=2 ...
Prints one row per shape: microseconds and TFLOP/s per setting."""
import math, os, sys
sys.path.insert(0, os.environ.get("UDT_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import udifftext_amd
from udifftext_amd import lib as L, ops, packing

dev = torch.device("cuda", 0)
GEGLU, TRANS = L.GEMM_GEGLU, L.GEMM_TRANSPOSED
# (M, N, K, flags, residual, rows_per_batch, name)
SHAPES = [
    (32768, 2560, 320, GEGLU, False, 0, "L0 geglu"), (32768, 320, 1280, 0, True, 0, "L0 ff-out"),
    (32768, 320, 320, 0, True, 0, "L0 to_out/proj"), (32768, 640, 320, 0, False, 0, "L0 q|k"), (32768, 960, 320, 0, False, 0, "L0 q|k|v"),
    (32768, 320, 320, TRANS, False, 4096, "L0 v^T"), (16384, 320, 320, 0, False, 0, "L0 t_attn q"),
    (8192, 5120, 640, GEGLU, False, 0, "L1 geglu"), (8192, 640, 2560, 0, True, 0, "L1 ff-out"),
    (8192, 640, 640, 0, True, 0, "L1 to_out/proj"), (8192, 1280, 640, 0, False, 0, "L1 q|k"), (8192, 1920, 640, 0, False, 0, "L1 q|k|v"),
    (4096, 640, 640, 0, False, 0, "L1 t_attn q"),
    (2048, 10240, 1280, GEGLU, False, 0, "L2 geglu"), (2048, 1280, 5120, 0, True, 0, "L2 ff-out"),
    (2048, 1280, 1280, 0, True, 0, "L2 to_out/proj"), (2048, 2560, 1280, 0, False, 0, "L2 q|k"), (2048, 3840, 1280, 0, False, 0, "L2 q|k|v"),
    (1024, 1280, 1280, 0, False, 0, "L2 t_attn q"), (512, 1280, 1280, 0, True, 0, "L3 to_out/proj"),
    (512, 10240, 1280, GEGLU, False, 0, "L3 geglu"), (512, 1280, 5120, 0, True, 0, "L3 ff-out"), (512, 3840, 1280, 0, False, 0, "L3 q|k|v"),
]
settings = sys.argv[1:] or ["n_block=-1"]
NBUF = 4


def apply(spec):
    for k in ("no_xchg", "no_epi", "no_store", "no_res", "no_bias", "no_fast"):      # measurement builds only: start clean
        L.load().udt_debug_set(k.encode(), 0)
    for k in ("lean", "lean_splitk", "n_block"):
        L.load().udt_debug_set(k.encode(), -1)
    for item in spec.split(","):
        k, v = item.split("=")
        L.check(L.load().udt_debug_set(k.encode(), int(v)), "udt_debug_set")


def time_shape(M, N, K, flags, res, rpb):
    xs = [torch.randn((M, K), device=dev).bfloat16() for _ in range(NBUF)]
    w = torch.randn((N, K), device=dev) / math.sqrt(K)
    if flags & GEGLU:
        wp, b = packing.pack_geglu(w, torch.randn((N,), device=dev))
    else:
        wp, b = packing.pack_linear(w), torch.randn((N,), device=dev)
    if flags & TRANS:
        b = None
    rs = [torch.randn((M, N), device=dev).bfloat16() for _ in range(NBUF)] if res else [None] * NBUF
    run = lambda i: ops.linear(xs[i % NBUF], wp, b, residual=rs[i % NBUF], flags=flags, rows_per_batch=rpb)
    for i in range(3):
        run(i)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for i in range(20):
            run(i)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 100 * 1e3


print("shape".ljust(44) + "".join(s.rjust(22) for s in settings))
for M, N, K, flags, res, rpb, name in SHAPES:
    row = f"{name:16s} {M:6d}x{N:5d}x{K:5d} fl={flags:<3d}"
    for spec in settings:
        apply(spec)
        us = time_shape(M, N, K, flags, res, rpb)
        row += f"{us:10.1f} us {2.0 * M * N * K / us / 1e6:6.0f} TF"
    print(row, flush=True)
apply("n_block=-1,lean=-1,lean_splitk=-1")
