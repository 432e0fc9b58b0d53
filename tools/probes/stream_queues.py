"""Which torch streams share a hardware queue?  Two streams each run a chain of narrow convolutions (32 workgroups,
~100 us each); on different queues the chains overlap (time ~ 1x), on one queue they serialise (~ 2x)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from udifftext_amd import ops, packing

dev = torch.device("cuda", 0)
n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 12
x = torch.randn((8, 16, 16, 1280), device=dev).bfloat16()
w = packing.pack_conv(torch.randn((1280, 1280, 3, 3), device=dev) / math.sqrt(1280 * 9))
b = torch.zeros((1280,), device=dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
outs = [torch.empty((8, 16, 16, 1280), dtype=torch.bfloat16, device=dev) for _ in range(n_streams)]
wss = [ops.Workspace(dev) for _ in range(n_streams)]


def chain(i, n=30):
    with torch.cuda.stream(streams[i]), ops.launch_context(cu_share=8, workspace=wss[i]):
        for _ in range(n):
            ops.conv2d(x, w, b, out=outs[i])


def timed(idx):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for st in (streams[i] for i in idx):
        st.wait_stream(torch.cuda.current_stream())
    for i in idx:
        chain(i)
    for i in idx:
        torch.cuda.current_stream().wait_stream(streams[i])
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e)


for i in range(n_streams):
    chain(i, 3)
t1 = timed([0])
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')} one chain: {t1:.2f} ms")
print("pair (0, j):  " + "  ".join(f"{j}:{timed([0, j]) / t1:.2f}" for j in range(1, n_streams)))
print("pair (1, j):  " + "  ".join(f"{j}:{timed([1, j]) / t1:.2f}" for j in range(2, n_streams)))
for k in (2, 3, 4, 5, 6, 8):
    if k <= n_streams:
        print(f"first {k} streams together: {timed(list(range(k))) / t1:.2f}x one chain")


# the default stream as one more lane (it owns a hardware queue of its own)
def timed_with_default(idx, n=30):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for st in (streams[i] for i in idx):
        st.wait_stream(torch.cuda.current_stream())
    for i in idx:
        chain(i, n)
    with ops.launch_context(cu_share=8, workspace=wss[-1]):
        for _ in range(n):
            ops.conv2d(x, w, b, out=outs[-1])
    for i in idx:
        torch.cuda.current_stream().wait_stream(streams[i])
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e)


for k in (1, 2, 3, 4):
    print(f"default stream + first {k} streams together: {timed_with_default(list(range(k))) / t1:.2f}x one chain")
print("streams 0,1,2 + 6: %.2f   0,1,2 + 7: %.2f   0,1,3: %.2f   0,1,6: %.2f  2,3: %.2f  2,6: %.2f  2,7: %.2f  3,6: %.2f 3,7: %.2f" % (
    timed([0, 1, 2, 6]) / t1, timed([0, 1, 2, 7]) / t1, timed([0, 1, 3]) / t1, timed([0, 1, 6]) / t1, timed([2, 3]) / t1,
    timed([2, 6]) / t1, timed([2, 7]) / t1, timed([3, 6]) / t1, timed([3, 7]) / t1))
