"""Where the hand-written kernels stand against the vendor libraries ON THE SAME SHAPES AND BOX: PyTorch-ROCm's own bf16 ops
(torch.nn.functional.linear -> hipBLASLt / rocBLAS, conv2d channels_last -> MIOpen, scaled_dot_product_attention -> its flash
backend) next to udifftext_amd's, each as 20 launches on rotating buffers inside one hipGraph (HBM-cold operands like inside a
step).  The vendor ops get the plain problem (no fused bias / residual / GEGLU epilogue), ours run the launch the UNet uses.
This is a yardstick for the roofline fractions in DESIGN.md, not part of the product path.   python tools/bench_vs_vendor.py"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import udifftext_amd
from udifftext_amd import ops, packing

dev = torch.device("cuda", 0)
torch.manual_seed(0)
NB = 4


def graph_time(fns, reps=5):
    for f in fns[:NB]:
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


def row(name, flops, t_ours, t_vendor):
    print(f"{name:44s} ours {t_ours:8.1f} us {flops / t_ours / 1e6:6.0f} TF   vendor {t_vendor:8.1f} us {flops / t_vendor / 1e6:6.0f} TF   "
          f"vendor/ours {t_vendor / t_ours:5.2f}", flush=True)


print("# linears (bf16, [M, K] x [N, K]^T)")
for M, N, K, name in [(32768, 320, 320, "L0 to_out / proj"), (32768, 960, 320, "L0 q|k|v"), (32768, 320, 1280, "L0 ff-out"),
                      (32768, 2560, 320, "L0 ff-in (ours: + GEGLU)"), (8192, 640, 640, "L1 to_out / proj"), (8192, 640, 2560, "L1 ff-out"),
                      (8192, 5120, 640, "L1 ff-in (ours: + GEGLU)"), (2048, 1280, 1280, "L2 to_out / proj"), (2048, 1280, 5120, "L2 ff-out"),
                      (2048, 10240, 1280, "L2 ff-in (ours: + GEGLU)"), (512, 1280, 1280, "L3 to_out / proj")]:
    geglu = "GEGLU" in name
    xs = [torch.randn((M, K), device=dev).bfloat16() for _ in range(NB)]
    w = torch.randn((N, K), device=dev) / math.sqrt(K)
    b = torch.zeros((N,), device=dev)
    wb = w.bfloat16()
    if geglu:
        wp, bp = packing.pack_geglu(w, b)
        outs = [torch.empty((M, N // 2), dtype=torch.bfloat16, device=dev) for _ in range(NB)]
        ours = [(lambda i=i: ops.linear(xs[i % NB], wp, bp, flags=udifftext_amd.lib.GEMM_GEGLU, out=outs[i % NB])) for i in range(20)]
    else:
        wp = packing.pack_linear(w)
        outs = [torch.empty((M, N), dtype=torch.bfloat16, device=dev) for _ in range(NB)]
        ours = [(lambda i=i: ops.linear(xs[i % NB], wp, b, out=outs[i % NB])) for i in range(20)]
    vouts = [torch.empty((M, N), dtype=torch.bfloat16, device=dev) for _ in range(NB)]
    vend = [(lambda i=i: torch.matmul(xs[i % NB], wb.t(), out=vouts[i % NB])) for i in range(20)]
    row(f"{name} {M}x{N}x{K}", 2.0 * M * N * K, graph_time(ours), graph_time(vend))

print("# 3x3 convolutions (bf16 NHWC, pad 1)")
for B, H, C, N, name in [(8, 64, 320, 320, "L0"), (8, 32, 640, 640, "L1"), (8, 16, 1280, 1280, "L2"), (8, 8, 1280, 1280, "L3")]:
    xs = [torch.randn((B, H, H, C), device=dev).bfloat16() for _ in range(NB)]
    w4 = torch.randn((N, C, 3, 3), device=dev) / math.sqrt(9 * C)
    w = packing.pack_conv(w4)
    b = torch.zeros((N,), device=dev)
    outs = [torch.empty((B, H, H, N), dtype=torch.bfloat16, device=dev) for _ in range(NB)]
    ours = [(lambda i=i: ops.conv2d(xs[i % NB], w, b, ksize=3, out=outs[i % NB])) for i in range(20)]
    xcl = [x.permute(0, 3, 1, 2) for x in xs]                         # NCHW views of NHWC storage = channels_last
    wcl = w4.bfloat16().contiguous(memory_format=torch.channels_last)
    vend = [(lambda i=i: F.conv2d(xcl[i % NB], wcl, None, padding=1)) for i in range(20)]
    row(f"{name} conv3x3 B{B} {H}x{H} {C}->{N}", 2.0 * B * H * H * N * C * 9, graph_time(ours), graph_time(vend))

print("# self-attention (bf16, head_dim 64)")
for B, Hh, N in [(8, 5, 4096), (8, 10, 1024), (8, 20, 256)]:
    qkv = [torch.randn((B, N, 3 * Hh * 64), device=dev).bfloat16() for _ in range(NB)]
    Cc = Hh * 64
    outs = [torch.empty((B, N, Cc), dtype=torch.bfloat16, device=dev) for _ in range(NB)]
    ours = [(lambda i=i: ops.attention_rowv(qkv[i % NB][..., :Cc], qkv[i % NB][..., Cc:2 * Cc], qkv[i % NB][..., 2 * Cc:], Hh, 0.125,
                                            out=outs[i % NB])) for i in range(20)]
    heads = [[t[..., j * Cc:(j + 1) * Cc].reshape(B, N, Hh, 64).permute(0, 2, 1, 3) for j in range(3)] for t in qkv]
    vend = [(lambda i=i: F.scaled_dot_product_attention(*heads[i % NB])) for i in range(20)]
    row(f"attention B{B} H{Hh} N{N}", 4.0 * B * Hh * N * N * 64, graph_time(ours), graph_time(vend))
