"""world_size-2 gloo test of the data-parallel layer (image sharding + the single all-gather of frames)."""
import os
import socket

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


def test_single_process_is_identity():
    x = torch.randn(2, 3, 4, 4)
    assert parallel.gather_frames(x, None) is x
