"""VAE mid-block attention (one head, 512 dims): the flash kernel (udt_attn512_fwd) vs the query-block GEMM -> softmax -> GEMM
form, per map size.   python tools/bench_attn512.py"""
import os, sys
sys.path.insert(0, os.environ.get("UDT_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
import udifftext_amd.lib
from udifftext_amd import ops

dev = torch.device("cuda", 0)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for B, N in [(4, 4096), (1, 4096), (2, 4096), (4, 1024), (1, 1024), (4, 9216), (1, 9216)]:
    qkv = torch.randn((B, N, 1536), device=dev).bfloat16()
    q, k, v = qkv[..., :512], qkv[..., 512:1024], qkv[..., 1024:]
    vt = v.permute(0, 2, 1).contiguous()
    o = torch.empty((B, N, 512), dtype=torch.bfloat16, device=dev)

    def block_form():
        for q0 in range(0, N, 1024):
            q1 = min(N, q0 + 1024)
            s = ops.bmm_nt(q[:, q0:q1], k, alpha=512 ** -0.5)
            ops.softmax_rows_(s)
            ops.bmm_nt(s, vt, out=o[:, q0:q1])

    fl = 4.0 * B * N * N * 512
    t_f = timed(lambda: ops.attention_d512(q, k, v, 512 ** -0.5, out=o))
    t_n = timed(lambda: ops.attention_d512(q, k, v, 512 ** -0.5, out=o, key_split=False))
    t_b = timed(block_form)
    ws = udifftext_amd.lib.load().udt_attn512_workspace_bytes(B, N, N)
    print(f"B={B} N={N:5d}: flash {t_f:9.1f} us {fl / t_f / 1e6:6.0f} TF (key split: {'%.1f MB scratch' % (ws / 1e6) if ws else 'not planned'})"
          f"    flash, no key split {t_n:9.1f} us    block form {t_b:9.1f} us {fl / t_b / 1e6:6.0f} TF", flush=True)
