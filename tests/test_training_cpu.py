"""Data-parallel gradient average of the training step (udifftext_amd.training.allreduce_gradients): ONE flat bucket per step, the mean
over the ranks lands in every rank's gradient tensors — world_size-2 gloo on the CPU.  (On the GPUs the same function takes the
reduce-scatter + all-gather form over RCCL.)  Reference: Lightning's DDP strategy, configs/train.yaml:19-22."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from udifftext_amd import training
    calls = []
    real = dist.all_reduce
    dist.all_reduce = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    names = ["b.weight", "a.weight", "a.bias"]
    g = torch.Generator().manual_seed(100 + rank)
    grads = {"a.weight": torch.randn((7, 5), generator=g), "a.bias": torch.randn((5,), generator=g), "b.weight": torch.randn((3, 3, 3), generator=g)}
    training.allreduce_gradients(grads, names, dist)
    out[rank] = ({k: v.clone() for k, v in grads.items()}, len(calls))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_average_over_two_ranks_is_one_collective():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    ref = {}
    for r in range(2):
        g = torch.Generator().manual_seed(100 + r)
        for k, shp in (("a.weight", (7, 5)), ("a.bias", (5,)), ("b.weight", (3, 3, 3))):
            ref[k] = ref.get(k, 0) + torch.randn(shp, generator=g) / 2
    for r in range(2):
        grads, n_calls = out[r]
        assert n_calls == 1
        for k in ref:
            assert torch.allclose(grads[k], ref[k], rtol=1e-6, atol=1e-7), (r, k)


def test_single_process_is_a_no_op():
    sys.path.insert(0, ROOT)
    from udifftext_amd import training
    g = {"w": torch.ones(4)}
    training.allreduce_gradients(g, ["w"], None)
    assert torch.equal(g["w"], torch.ones(4))
