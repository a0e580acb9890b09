"""world_size-2 gloo test of the clip sharding + feature all-gather (CPU; the per-clip extractor is a stand-in
function: the collective and the partitioning are what is under test)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

import relax_vqa_amd  # noqa: F401
from relax_vqa_amd import distributed as rd


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_clip_vector(i, F=37):
    g = torch.Generator().manual_seed(1000 + i)
    return torch.randn(F, generator=g)


def _worker(rank, world, port, n_clips, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import relax_vqa_amd  # noqa: F401
    from relax_vqa_amd import distributed as rd2
    r, w, _ = rd2.init_from_env(backend="gloo")
    out = rd2.extract_dataset(_fake_clip_vector, n_clips, r, w)
    q.put((rank, out.numpy().copy()))   # by value: a shared-memory tensor would die with this process before the parent reads it
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_clips", [2, 5, 8])
def test_two_rank_gather_equals_single_rank(n_clips):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = torch.stack([_fake_clip_vector(i) for i in range(n_clips)])
    for r in range(world):
        assert torch.equal(torch.from_numpy(results[r]), want), "sharding changed values or order"   # bit-for-bit


def test_shards_partition_all_clips():
    for n, w in [(1, 1), (7, 2), (1200, 8), (9, 4)]:
        seen = sum((rd.shard_clips(n, r, w) for r in range(w)), [])
        assert seen == list(range(n))
        sizes = [len(rd.shard_clips(n, r, w)) for r in range(w)]
        assert max(sizes) - min(sizes) <= 1
