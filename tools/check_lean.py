"""Correctness of the lean GEMM family (csrc/lean.h) against the 8-wave kernels and a torch fp32 reference, on the
UNet's shapes plus ragged ones:  python tools/check_lean.py [lean configs ...]   (default 1 2 3)"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
from udifftext_amd import lib as L, ops, packing

dev = torch.device("cuda", 0)
GEGLU = L.GEMM_GEGLU
torch.manual_seed(0)


def dbg(k, v):
    L.check(L.load().udt_debug_set(k.encode(), int(v)), "udt_debug_set " + k)


def rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30)).item(), (a - b).abs().max().item()


CASES = [  # M, N, K, flags, residual, rowvec rows_per_batch
    (32768, 320, 320, 0, True, 0), (32768, 960, 320, 0, False, 0), (32768, 2560, 320, GEGLU, False, 0),
    (8192, 640, 2560, 0, True, 0), (2048, 1280, 5120, 0, True, 0), (512, 1280, 1280, 0, True, 0),
    (512, 10240, 1280, GEGLU, False, 0), (1000, 328, 192, 0, True, 0), (130, 136, 64, 0, False, 0),
    (4096, 640, 640, 0, True, 1024), (300, 1280, 1280, 0, False, 100),
]
cfgs = [int(a) for a in sys.argv[1:]] or [1, 2, 3]
bad = 0
for M, N, K, fl, res, rpb in CASES:
    x = torch.randn((M, K), device=dev).bfloat16()
    w = torch.randn((N, K), device=dev) / math.sqrt(K)
    b = torch.randn((N,), device=dev)
    if fl & GEGLU:
        wp, bp = packing.pack_geglu(w, b)
    else:
        wp, bp = packing.pack_linear(w), b
    r = torch.randn((M, N), device=dev).bfloat16() if res else None
    rv = torch.randn((M // rpb, N), device=dev) if rpb else None
    kw = dict(residual=r, rowvec=rv, rows_per_batch=rpb, flags=fl)
    dbg("lean", 0)
    ref8 = ops.linear(x, wp, bp, **kw)
    # torch reference
    y = x.float() @ w.bfloat16().float().t() + b
    if fl & GEGLU:
        h = N // 2
        y = y[:, :h] * torch.nn.functional.gelu(y[:, h:])
    if rv is not None:
        y = y + rv.repeat_interleave(rpb, 0)
    if r is not None:
        y = y + r.float()
    e8 = rel(ref8, y)
    line = f"{M:6d}x{N:5d}x{K:5d} fl={fl} res={int(res)} rpb={rpb:4d}  gemm8 vs torch {e8[0]:.2e}"
    for c in cfgs:
        for sk in (-1, 3):
            dbg("lean", c); dbg("lean_splitk", sk)
            out = ops.linear(x, wp, bp, **kw)
            torch.cuda.synchronize()
            e = rel(out, y)
            d8 = (out.float() - ref8.float()).abs().max().item()
            ok = e[0] < 6e-3 and math.isfinite(e[0])
            bad += 0 if ok else 1
            line += f" | lean{c}{'s' if sk > 0 else ' '} {e[0]:.2e} d8 {d8:.1e}{'' if ok else ' BAD'}"
    dbg("lean_splitk", -1)
    print(line, flush=True)

# two-source 1x1 convolutions (skip concat in front of a ResBlock's skip_connection) on the lean GEMM family
import torch.nn.functional as F
for B, H, C1, C2, N in [(8, 64, 320, 320, 320), (8, 32, 640, 320, 640), (8, 16, 1280, 1280, 1280), (8, 8, 1280, 1280, 1280), (2, 16, 128, 64, 192)]:
    x1 = torch.randn((B, H, H, C1), device=dev).bfloat16(); x2 = torch.randn((B, H, H, C2), device=dev).bfloat16()
    w4 = torch.randn((N, C1 + C2, 1, 1), device=dev) / math.sqrt(C1 + C2)
    w = packing.pack_conv(w4, [C1, C2]); b = torch.randn((N,), device=dev)
    y = F.conv2d(torch.cat([x1, x2], -1).float().permute(0, 3, 1, 2), w4.bfloat16().float(), b).permute(0, 2, 3, 1)
    line = f"conv1x1 B{B} {H}x{H} {C1}+{C2}->{N}:"
    for c in [0] + cfgs[:1]:
        dbg("lean", c)
        o = ops.conv2d(x1, w, b, ksize=1, x2=x2)
        torch.cuda.synchronize()
        e = rel(o, y)
        ok = e[0] < 6e-3 and math.isfinite(e[0]); bad += 0 if ok else 1
        line += f"  lean={c} {e[0]:.2e}{'' if ok else ' BAD'}"
    print(line, flush=True)

# LayerNorm-folded form against layer_norm + linear and a torch fp32 reference
for M, N, K, fl in [(32768, 960, 320, 0), (8192, 5120, 640, GEGLU), (2048, 3840, 1280, 0), (777, 2560, 320, GEGLU), (512, 1280, 1280, 0)]:
    x = (torch.randn((M, K), device=dev) * 1.5 + 0.3).bfloat16()
    w = torch.randn((N, K), device=dev) / math.sqrt(K)
    b = torch.randn((N,), device=dev)
    gamma = 1 + 0.1 * torch.randn((K,), device=dev)
    beta = 0.05 * torch.randn((K,), device=dev)
    xn = torch.nn.functional.layer_norm(x.float(), (K,), gamma, beta, 1e-5)
    y = xn @ w.t() + b
    if fl & GEGLU:
        h = N // 2
        y = y[:, :h] * torch.nn.functional.gelu(y[:, h:])
    dbg("lean", 0)
    wp, bp = (packing.pack_geglu(w, b) if fl & GEGLU else (packing.pack_linear(w), b))
    two = ops.linear(ops.layer_norm(x, gamma, beta), wp, bp, flags=fl)
    e2 = rel(two, y)
    line = f"LN {M:6d}x{N:5d}x{K:5d} fl={fl}  layernorm+gemm8 vs torch {e2[0]:.2e}"
    for c in cfgs:
        dbg("lean", c)
        wf, cf, sf = packing.pack_ln_linear(w, b, gamma, beta, geglu=bool(fl & GEGLU))
        out = ops.ln_linear(x, wf, cf, sf, flags=fl)
        torch.cuda.synchronize()
        e = rel(out, y)
        ok = e[0] < 1e-2 and math.isfinite(e[0])
        bad += 0 if ok else 1
        line += f" | lean{c} {e[0]:.2e}{'' if ok else ' BAD'}"
    print(line, flush=True)
dbg("lean", -1)
print("FAILED" if bad else "ALL OK", bad)
