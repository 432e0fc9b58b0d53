"""The transformer linears of a config #2 UNet call (8 samples) at the 32x32 / 16x16 / 8x8 levels: the bf16 launch the default path
makes vs the MX8 launch config #5 makes for the same layer (incl. what the epilogues emit for the next layer), timed back to back
inside hipGraphs on rotating buffers like tools/bench_gemm_shapes.py.  One row per layer: microseconds, TFLOP/s, ratio."""
import math, os, sys
sys.path.insert(0, os.environ.get("UDT_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import udifftext_amd
from udifftext_amd import lib as L, ops, packing
import mx8_ref

dev = torch.device("cuda", 0)
for item in sys.argv[1:]:                      # udt_debug_set knobs, e.g. lean_krot=0
    k, v = item.split("=")
    L.check(L.load().udt_debug_set(k.encode(), int(v)), "udt_debug_set")
GEGLU = L.GEMM_GEGLU
NBUF = 4
# (name, M, N, K, kind): kind in plain (to_out / proj_out: residual), emit (ff-out: residual + MX8 twin), proj_in (bf16 in, MX8 twin +
# row statistics out), qkv (LayerNorm-folded), geglu (LayerNorm-folded GEGLU, hidden as MX8 only)
LAYERS = []
for lvl, M, C in (("L0", 32768, 320), ("L1", 8192, 640), ("L2", 2048, 1280), ("L3", 512, 1280)):
    if C % 128 != 0:
        continue
    LAYERS += [(f"{lvl} proj_in", M, C, C, "proj_in"), (f"{lvl} q|k|v", M, 3 * C, C, "qkv"), (f"{lvl} to_out", M, C, C, "plain"),
               (f"{lvl} geglu", M, 8 * C, C, "geglu"), (f"{lvl} ff-out", M, C, 4 * C, "emit"), (f"{lvl} proj_out", M, C, C, "plain")]


def graph_time(run):
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


print(f"{'layer':14s} {'M x N x K':>20s} {'bf16':>18s} {'mx8':>18s}  ratio")
tot_b = tot_m = 0.0
for name, M, N, K, kind in LAYERS:
    xs = [torch.randn((M, K), device=dev) for _ in range(NBUF)]
    xb = [x.bfloat16() for x in xs]
    acts = []
    for x in xs:
        q, s = mx8_ref.encode(x)
        P = K // 64
        st = torch.stack([x.reshape(M, P, 64).sum(dim=2).t(), x.reshape(M, P, 64).pow(2).sum(dim=2).t()], dim=2).contiguous()
        acts.append(ops.Mx8Act(q, s, st))
    w = torch.randn((N, K), device=dev) / math.sqrt(K)
    b = torch.randn((N,), device=dev)
    gamma, beta = torch.ones((K,), device=dev), torch.zeros((K,), device=dev)
    rs = [torch.randn((M, N), device=dev).bfloat16() for _ in range(NBUF)] if kind in ("plain", "emit") else [None] * NBUF
    if kind == "geglu":
        wf, c, s_ = packing.pack_ln_linear(w, b, gamma, beta, geglu=True)
        run_b = lambda i: ops.ln_linear(xb[i % NBUF], wf, c, s_, flags=GEGLU)
        wq, cs, c8, s8 = packing.pack_ln_linear_mx8(w, b, gamma, beta, geglu=True)
        run_m = lambda i: ops.linear_mx8(acts[i % NBUF], wq, cs, ln_c=c8, ln_s=s8, flags=GEGLU, emit_q8=True, want_bf16=False)
    elif kind == "qkv":
        wf, c, s_ = packing.pack_ln_linear(w, None, gamma, beta)
        run_b = lambda i: ops.ln_linear(xb[i % NBUF], wf, c, s_)
        wq, cs, c8, s8 = packing.pack_ln_linear_mx8(w, None, gamma, beta)
        run_m = lambda i: ops.linear_mx8(acts[i % NBUF], wq, cs, ln_c=c8, ln_s=s8)
    elif kind == "proj_in":
        wp = packing.pack_linear(w)
        run_b = lambda i: ops.linear(xb[i % NBUF], wp, b)
        run_m = lambda i: ops.linear(xb[i % NBUF], wp, b, emit_q8=True, emit_rowstats=True)       # (bf16 operands, emitting epilogue)
    else:
        wp = packing.pack_linear(w)
        run_b = lambda i: ops.linear(xb[i % NBUF], wp, b, residual=rs[i % NBUF])
        wq, cs = packing.pack_linear_fp8(w)
        run_m = lambda i: ops.linear_mx8(acts[i % NBUF], wq, cs, b, residual=rs[i % NBUF], emit_q8=(kind == "emit"))
    tb, tm = graph_time(run_b), graph_time(run_m)
    fl = 2.0 * M * N * K
    tot_b += tb; tot_m += tm
    print(f"{name:14s} {M:6d}x{N:5d}x{K:5d} {tb:8.1f} us {fl / tb / 1e6:5.0f} TF {tm:8.1f} us {fl / tm / 1e6:5.0f} TF  {tb / tm:5.2f}x", flush=True)
print(f"{'sum':14s} {'':20s} {tot_b:8.1f} us {'':8s} {tot_m:8.1f} us {'':8s}  {tot_b / tot_m:5.2f}x")
