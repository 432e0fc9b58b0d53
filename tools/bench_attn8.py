"""attn1 of a config #2 UNet call (8 samples) at the four levels: q|k|v projection + self-attention as config #5 runs them with bf16
attention (projection -> bf16 q|k|v; attn_d64_v2_kernel, O also as MX8 where to_out is e4m3) and with the e4m3 attention
(projection -> MX8 q|k|v only; attn_d64_mx8_kernel), each launch timed inside hipGraphs on rotating buffers.  Microseconds."""
import math, os, sys
sys.path.insert(0, os.environ.get("UDT_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import udifftext_amd
from udifftext_amd import lib as L, ops, packing
import mx8_ref

dev = torch.device("cuda", 0)
for item in sys.argv[1:]:
    k, v = item.split("=")
    L.check(L.load().udt_debug_set(k.encode(), int(v)), "udt_debug_set")
NBUF = 4
VM = 32.0


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


print(f"{'level':6s} {'B x N x C':>16s} | {'qkv bf16':>9s} {'qkv->mx8':>9s} | {'attn bf16':>9s} {'attn e4m3':>9s} | {'sum bf16':>9s} {'sum e4m3':>9s} ratio")
tb = t8 = 0.0
for lvl, B, N, C, count in (("L0", 8, 4096, 320, 5), ("L1", 8, 1024, 640, 5), ("L2", 8, 256, 1280, 5), ("L3", 8, 64, 1280, 1)):
    M, heads = B * N, C // 64
    xs = [torch.randn((M, C), device=dev) for _ in range(NBUF)]
    w = torch.randn((3 * C, C), device=dev) / math.sqrt(C)
    gamma, beta = torch.ones((C,), device=dev), torch.zeros((C,), device=dev)
    mx_in = C % 128 == 0
    if mx_in:
        acts = []
        for x in xs:
            q, s = mx8_ref.encode(x)
            P = C // 64
            st = torch.stack([x.reshape(M, P, 64).sum(dim=2).t(), x.reshape(M, P, 64).pow(2).sum(dim=2).t()], dim=2).contiguous()
            acts.append(ops.Mx8Act(q, s, st))
        wq, cs, c, sv = packing.pack_ln_linear_mx8(w, None, gamma, beta)
        outs = [torch.empty((M, 3 * C), dtype=torch.bfloat16, device=dev) for _ in range(NBUF)]
        f_b = lambda i: ops.linear_mx8(acts[i % NBUF], wq, cs, ln_c=c, ln_s=sv, out=outs[i % NBUF])
        f_8 = lambda i: ops.linear_mx8(acts[i % NBUF], wq, cs, ln_c=c, ln_s=sv, emit_q8=True, want_bf16=False, q8_fixed=(2 * C, VM))
    else:
        xb = [x.bfloat16() for x in xs]
        wf, c, sv = packing.pack_ln_linear(w, None, gamma, beta)
        outs = [torch.empty((M, 3 * C), dtype=torch.bfloat16, device=dev) for _ in range(NBUF)]
        f_b = lambda i: ops.ln_linear(xb[i % NBUF], wf, c, sv, out=outs[i % NBUF])
        f_8 = lambda i: ops.ln_linear(xb[i % NBUF], wf, c, sv, emit_q8=True, want_bf16=False, q8_fixed=(2 * C, VM))
    qkvs = [f_b(i).reshape(B, N, 3 * C) for i in range(NBUF)]
    q8s = [f_8(i) for i in range(NBUF)]
    oo = [torch.empty((B, N, C), dtype=torch.bfloat16, device=dev) for _ in range(NBUF)]
    a_b = lambda i: ops.attention_rowv(qkvs[i % NBUF][..., :C], qkvs[i % NBUF][..., C:2 * C], qkvs[i % NBUF][..., 2 * C:], heads, 0.125,
                                       out=oo[i % NBUF], emit_q8=mx_in)
    a_8 = lambda i: ops.attention_mx8(q8s[i % NBUF], B, heads, 0.125, VM, out=oo[i % NBUF], emit_q8=mx_in)
    t = [graph_time(f) for f in (f_b, f_8, a_b, a_8)]
    sb, s8 = t[0] + t[2], t[1] + t[3]
    tb += count * sb
    t8 += count * s8
    print(f"{lvl:6s} {B:3d} x {N:4d} x {C:4d} | {t[0]:9.1f} {t[1]:9.1f} | {t[2]:9.1f} {t[3]:9.1f} | {sb:9.1f} {s8:9.1f} {s8 / sb:.3f}")
print(f"per UNet call (5 + 5 + 5 + 1 blocks): bf16 attention {tb:.0f} us, e4m3 attention {t8:.0f} us ({t8 / tb:.3f})")
