"""cost attribution of the row-resident GEMM's loop: times the launch in the library under UDT_ROOT (one measurement build per
compile-time mask: UDT_EXTRA_FLAGS="-DUDT_MEASURE -DRR_DBG_MASK=<bits>", wrong results).  python tools/rowres_cost_attribution.py <label>"""
import math, os, sys
sys.path.insert(0, os.environ.get("UDT_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
from udifftext_amd import lib as L, ops, packing

dev = torch.device("cuda", 0)
lib = L.load()
label = sys.argv[1] if len(sys.argv) > 1 else ""
for M, N, K, geglu in [(32768, 2560, 320, True), (32768, 960, 320, False)]:
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn((M, K), generator=g).to(dev).bfloat16()
    w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(dev)
    b = torch.randn((N,), generator=g).to(dev)
    wf, cf, sf = packing.pack_ln_linear(w, b, torch.ones((K,), device=dev), torch.zeros((K,), device=dev), geglu=geglu)
    fl = L.GEMM_GEGLU if geglu else 0
    out = torch.empty((M, N // 2 if geglu else N), dtype=torch.bfloat16, device=dev)
    import ctypes as C
    d = ops.gemm_desc(a=x.data_ptr(), w=wf.data_ptr(), bias=cf.data_ptr(), residual=None, out=out.data_ptr(), M=M, N=N, K=K, lda=K,
                      ldo=out.stride(0), ldr=0, flags=fl, ln_colsum=sf.data_ptr(), ln_eps=1e-5)
    st = torch.cuda.current_stream().cuda_stream
    fn = lambda: lib.udt_ln_gemm_fwd(C.byref(d), None, 0, st)          # (bare C-ABI call: ~3 us of host time)
    assert fn() == 0
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{label:32s} {M} x {N} x {K} {'geglu' if geglu else 'plain'}: {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us per launch", flush=True)
