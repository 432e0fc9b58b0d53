"""A/B timing of one sampler step (CFG pair, 512x512, batch 4) inside ONE process: toggles given as name=values.

    python tools/ab_step.py zero                      # zero-context shortcut on / off
    python tools/ab_step.py dbg:rows_epi 1 0          # a tuning knob of udt_debug_set
    python tools/ab_step.py py:sgm.modules.hipnn.GN_FOLD_PROJ_IN 1 0     # a module-level Python switch (values through eval)
    python tools/ab_step.py multi "" no_epi=1 ...     # cost attribution: needs a MEASUREMENT build of the library
                                                      # (UDT_EXTRA_FLAGS=-DUDT_MEASURE python -m udifftext_amd.build --force)
"""
import os, sys, time
sys.path.insert(0, os.environ.get("UDT_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import udifftext_amd
from udifftext_amd import pipeline, synth
from sgm.modules.diffusionmodules.sampling import _Stepper
dev = torch.device("cuda", 0)
torch.set_grad_enabled(False)
B, size = int(os.environ.get("AB_B", "4")), int(os.environ.get("AB_SIZE", "512"))     # (AB_B=1: the reference-default batch)
model = pipeline.build_engine(dev)
sampler = pipeline.init_sampling(50, 5.0, dev)
b = synth.synthetic_batch(B, size, size, 9, seed=0)
b = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
batch, buc = pipeline.prepare_batch(b, dev)
c, uc = model.conditioner.get_unconditional_conditioning(batch, batch_uc=buc, force_uc_zero_embeddings=["label"])
st = _Stepper(model, c, uc, B, (size // 8, size // 8), 5.0)
sig = sampler._host_sigmas()
x = torch.randn((B, 4, size // 8, size // 8), device=dev) * 14.0

def run(n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        st.step(x, sig[5 + i], sig[6 + i])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

import udifftext_amd.ops as O
which = sys.argv[1] if len(sys.argv) > 1 else "zero"
if which == "zero":
    variants = {"zero_rows=B": lambda: setattr(st, "zero_ctx_rows", B), "zero_rows=0": lambda: setattr(st, "zero_ctx_rows", 0)}
elif which == "multi":
    from udifftext_amd import lib as L
    def setall(**kw):
        for k, v in kw.items():
            L.check(L.load().udt_debug_set(k.encode(), v), "dbg")
    base = dict(no_xchg=0, no_epi=0, no_store=0, no_res=0, no_bias=0, no_fast=0)
    variants = {}
    for spec in sys.argv[2:]:
        kv = dict(base)
        for item in spec.split(","):
            if item:
                k, v = item.split("="); kv[k] = int(v)
        variants[spec or "base"] = (lambda kv=kv: setall(**kv))
elif which.startswith("dbg:"):
    key = which[4:]
    from udifftext_amd import lib as L
    vals = [int(v) for v in (sys.argv[2:] or ["1", "0"])]
    variants = {f"{key}={v}": (lambda v=v: L.check(L.load().udt_debug_set(key.encode(), v), "dbg")) for v in vals}
elif which.startswith("py:"):
    import importlib
    modname, attr = which[3:].rsplit(".", 1)
    mod = importlib.import_module(modname)
    vals = [eval(v) for v in (sys.argv[2:] or ["True", "False"])]
    variants = {f"{attr}={v}": (lambda v=v: setattr(mod, attr, v)) for v in vals}
else:
    raise SystemExit(__doc__)
if os.environ.get("AB_EAGER"):
    # eager launches: host-bound (~25 us of Python / ctypes per launch) — only differences in LAUNCH COUNT show
    for name, fn in variants.items():
        fn(); run(3)
    for rnd in range(4):
        for name, fn in variants.items():
            fn()
            print(f"round {rnd} {name:14s} {run(10):.3f} ms/step (eager)", flush=True)
    raise SystemExit(0)
# default: every variant captures ten sampler steps into hipGraphs (as the sampler runs them); rounds replay them alternately
import sgm.modules.diffusionmodules.sampling as S
runners = {}
for name, fn in variants.items():
    fn()
    gs = S._GraphedSteps(model, c, uc, B, (size // 8, size // 8), 5.0, sig)
    gs.x.copy_(x)
    for i in range(5, 15):
        gs._capture(i)
    runners[name] = gs
def replay(gs):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(gs.capture_stream):
        gs.graphs[5].replay()
        e0.record()
        for i in range(5, 15):
            gs.graphs[i].replay()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10
for rnd in range(5):
    for name, gs in runners.items():
        print(f"round {rnd} {name:28s} {replay(gs):.3f} ms/step (graph replay)", flush=True)
