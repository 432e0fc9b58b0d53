"""world_size-2 gloo test of the data-parallel layer (image sharding + the single all-gather of frames)."""
import os
import socket

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from udifftext_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_images, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = parallel.shard_range(n_images, rank, world)
    # "decode" image i into a frame whose content identifies it (seeded by the world-size-independent image seed)
    frames = torch.stack([torch.full((3, 4, 4), float(parallel.image_seed(7, i) % 1000)) for i in range(b, e)]) \
        if e > b else torch.zeros((0, 3, 4, 4))
    counts = [parallel.shard_range(n_images, r, world)[1] - parallel.shard_range(n_images, r, world)[0] for r in range(world)]
    if len(set(counts)) == 1:
        out = parallel.gather_frames(frames, dist)
    else:
        out = parallel.gather_ragged(frames, counts, dist)
    q.put((rank, out[:, 0, 0, 0].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def _run(n_images, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_images, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_shard_ranges_cover_everything():
    for n in (1, 7, 8, 64):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_all_gather_of_frames_world2_even():
    res = _run(8)
    expect = [float(parallel.image_seed(7, i) % 1000) for i in range(8)]
    assert res[0] == expect and res[1] == expect          # every rank holds every frame, in global image order


def test_all_gather_of_frames_world2_ragged():
    res = _run(5)
    expect = [float(parallel.image_seed(7, i) % 1000) for i in range(5)]
    assert res[0] == expect and res[1] == expect


# ---- the sharded entry point end to end (pipeline.predict_many + rng + one all-gather) with a stub engine ----------
class _StubConditioner:
    def get_unconditional_conditioning(self, batch, batch_uc=None, force_uc_zero_embeddings=None):
        from udifftext_amd import rng
        B = batch["image"].shape[0]
        feat = batch["image"].mean(dim=(1, 2, 3)).reshape(B, 1, 1, 1)
        c = {"concat": rng.randn((B, 4, 2, 2)) + feat}            # draw 1: posterior noise of c
        uc = {"concat": rng.randn((B, 4, 2, 2)) + feat}           # draw 2: posterior noise of uc
        return c, uc


class _StubModel:
    conditioner = _StubConditioner()

    def decode_first_stage(self, z):
        return z[:, :3].repeat_interleave(2, -1).repeat_interleave(2, -2) * 0.1


class _StubSampler:
    def get_init_noise(self, cfgs, model, cond, batch, uc=None):
        from udifftext_amd import rng
        assert cfgs.batch_size == cond["concat"].shape[0]
        return rng.randn((cfgs.batch_size, 4, 2, 2))              # draw 3: x0

    def sample_in_flight(self, model, xs, conds, ucs, init_step=0, deferred_checks=None, streams=None):
        return [x + 0.5 * c["concat"] - 0.25 * u["concat"] for x, c, u in zip(xs, conds, ucs)]


def _global_batch(n, seed):
    g = torch.Generator().manual_seed(seed)
    return {"image": torch.rand((n, 3, 8, 8), generator=g), "label": [f"img{i}" for i in range(n)],
            "txt": [f"t{i}" for i in range(n)], "name": [str(i) for i in range(n)]}


def _sharded(dist_mod, n_images, micro):
    from udifftext_amd import config as C
    cfgs = C.default_runtime_config(steps=2, batch_size=micro, noise_iters=0)
    gbs = [_global_batch(n_images, 1), _global_batch(n_images, 2)]
    return parallel.predict_sharded(cfgs, _StubModel(), _StubSampler(), gbs, [11, 12], dist=dist_mod, micro_batch=micro,
                                    in_flight=2, fuse=1, device=torch.device("cpu"))


def _worker_sharded(rank, world, port, n_images, micro, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []
    orig = dist.all_gather_into_tensor
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    frames = _sharded(dist, n_images, micro)
    q.put((rank, [f.numpy() for f in frames], len(calls)))
    dist.barrier()
    dist.destroy_process_group()


def _run_sharded(n_images, micro, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sharded, args=(r, world, port, n_images, micro, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, frames, ncoll = q.get(timeout=180)
        res[r] = (frames, ncoll)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_predict_sharded_world2_equals_world1():
    """config #3's shape in miniature: a global batch sharded over 2 ranks gives, on every rank, exactly the frames
    (values AND order) a single process computes — per-image seeds make the draws independent of world size and of
    how a shard is cut into micro-batches — with ONE all-gather per global batch"""
    import numpy as np
    for n_images, micro in ((8, 2), (5, 2), (3, 4)):
        single = _sharded(None, n_images, micro)
        other_micro = _sharded(None, n_images, 3)
        res = _run_sharded(n_images, micro)
        for gi in range(2):
            assert single[gi].shape == (n_images, 3, 4, 4)
            np.testing.assert_array_equal(single[gi].numpy(), other_micro[gi].numpy())   # batching-independent
            for r in (0, 1):
                np.testing.assert_array_equal(res[r][0][gi], single[gi].numpy())
        assert res[0][1] == 2 and res[1][1] == 2             # one collective per global batch
    assert not torch.equal(_sharded(None, 4, 2)[0], _sharded(None, 4, 2)[1])              # seeds / inputs differ


def test_per_image_noise_source():
    from udifftext_amd import rng
    with rng.per_image([5, 6, 7]):
        a = rng.randn((3, 4, 2, 2))
        b = rng.randn((3, 4, 2, 2))
    with rng.per_image([6]):
        a1 = rng.randn((1, 4, 2, 2))
        b1 = rng.randn((1, 4, 2, 2))
    assert torch.equal(a[1:2], a1) and torch.equal(b[1:2], b1) and not torch.equal(a, b)
    torch.manual_seed(3)
    ref = torch.randn((2, 4, 2, 2))
    torch.manual_seed(3)
    assert torch.equal(rng.randn((2, 4, 2, 2)), ref)            # outside a context: the reference's global CPU stream


def test_single_process_is_identity():
    x = torch.randn(2, 3, 4, 4)
    assert parallel.gather_frames(x, None) is x


# ---- lane assignment of pipeline.predict_many (host logic, stub engine) -------------------------------------------
@pytest.mark.parametrize("n_batches,in_flight,fuse,expect", [(4, 3, 1, [(0, 2), (1, 2), (0, 2), (1, 2)]), (5, 3, 1, [(0, 3), (1, 3), (2, 3), (0, 3), (1, 3)]),
                                                             (7, 3, 1, [(0, 3), (1, 3), (2, 3), (0, 3), (1, 3), (2, 3), (0, 3)]),
                                                             (2, 3, 1, [(0, 2), (1, 2)]), (3, 1, 1, [(0, 1), (0, 1), (0, 1)]),
                                                             (5, 2, 2, [(0, 2), (1, 2), (0, 2)])])
def test_predict_many_free_running_lanes(n_batches, in_flight, fuse, expect):
    """sampling batch u runs on lane u % lanes, every lane planned for 1 / lanes of the CUs; the lane count is the smallest that
    keeps the number of rounds (4 batches with 3 lanes allowed: 2 + 2 on two lanes, not 2 + 1 + 1); results keep the input order and
    the per-batch CPU draw order whatever the lane count"""
    from udifftext_amd import config as C, pipeline

    class Rec(_StubSampler):
        calls = []

        def sample_lane(self, model, x, cond, uc, slot, n_lanes, init_step=0, deferred_checks=None):
            Rec.calls.append((slot, n_lanes))
            return super().sample_in_flight(model, [x], [cond], [uc], init_step)[0]

    Rec.calls = []
    cfgs = C.default_runtime_config(steps=2, batch_size=2, noise_iters=0)
    batches = [_global_batch(2, 100 + i) for i in range(n_batches)]
    torch.manual_seed(7)
    outs = pipeline.predict_many(cfgs, _StubModel(), Rec(), batches, torch.device("cpu"), in_flight=in_flight, fuse=fuse)
    assert Rec.calls == expect and len(outs) == n_batches
    torch.manual_seed(7)
    seq = pipeline.predict_many(cfgs, _StubModel(), _StubSampler(), [_global_batch(2, 100 + i) for i in range(n_batches)],
                                torch.device("cpu"), in_flight=1, fuse=1)
    for (a, _), (b, _) in zip(outs, seq):
        assert torch.equal(a, b)
