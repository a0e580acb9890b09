"""The dataset driver (relax-vqa_amd/dataset.py: BASELINE config 4 as written) on CPU with a stand-in engine: sharding, batching,
per-frame files + resume, the NaN-row failure contract and the all-gather, at world size 1 and 2 (gloo).  The stand-in computes
per-frame rows from the clip's bytes, so any mix-up of clips, rows or ranks changes values."""
import os
import socket
import time

import numpy as np
import pytest
import scipy.io
import torch
import torch.multiprocessing as mp

import relax_vqa_amd  # noqa: F401
from relax_vqa_amd import dataset, sampling

F = 11


class FakeEngine:
    """clip_vectors(list of clips) -> [len, F] means of per-frame rows; raises on a 'poisoned' clip (first byte 255) the way a
    failing engine call would take a whole batch down."""
    device = torch.device("cpu")
    vit_dim = None

    def __init__(self):
        self.batches = []

    @staticmethod
    def _rows(clip):
        c = torch.as_tensor(np.asarray(clip)).to(torch.float64)
        base = c.reshape(c.shape[0], -1).mean(dim=1, keepdim=True)
        return (base * torch.arange(1, F + 1, dtype=torch.float64)).to(torch.float32)

    def rows_mean(self, rows):
        return rows.mean(dim=0)   # the reduction clip_vectors below applies to fresh rows

    def clip_vectors(self, clips, resnet=True, vit=True, per_frame=False):
        self.batches.append(len(clips))
        for c in clips:
            if int(np.asarray(c).reshape(-1)[0]) == 255:
                raise RuntimeError("relax_fragment_pairs failed (-2): injected fault")
        rows = [self._rows(c) for c in clips]
        out = torch.stack([r.mean(dim=0) for r in rows])
        return (out, rows) if per_frame else out


def _clip(i):
    if i == 3:
        raise OSError(f"cannot decode video_{i + 1}.mp4")
    g = np.random.default_rng(i)
    t = 1 + i % 4                                            # ragged: 1..4 pairs
    c = g.integers(0, 255, (t, 2, 16 + 16 * (i % 2), 32, 3), dtype=np.uint8)
    if i == 5:
        c.reshape(-1)[0] = 255                               # the engine itself fails on this one
    if i == 6:
        return c[:, :1]                                      # malformed: no `next` frame
    return c


def _want(n):
    want = np.full((n, F), np.nan, dtype=np.float32)
    for i in range(n):
        if i in (3, 5, 6):
            continue
        want[i] = FakeEngine._rows(_clip(i)).mean(dim=0).numpy()
    return want


def _patched(monkeypatch_dim=None):
    dataset.feature_dim = lambda engine, resnet=True, vit=True, full=False: F   # the stand-in's width


def test_single_rank_ragged_clips_failures_files_and_resume(tmp_path):
    _patched()
    eng = FakeEngine()
    n = 10
    out_dir = str(tmp_path / "feats")
    mat = str(tmp_path / "m" / "konvid.mat")
    matrix, errors = dataset.extract_dataset_clips(_clip, n, eng, clips_per_step=4, out_dir=out_dir, network_name="resnet50",
                                                   mat_path=mat, data_name="konvid_1k", rank=0, world=1)
    want = _want(n)
    assert np.array_equal(matrix.numpy(), want, equal_nan=True)
    assert [i for i, _ in errors] == [3, 5, 6]
    assert "cannot decode" in errors[0][1] and "injected fault" in errors[1][1] and "[T,2,H,W,3]" in errors[2][1]
    # the batch that held the poisoned clip was retried clip by clip: its healthy neighbours kept their rows
    assert not np.isnan(want[4]).any() and np.array_equal(matrix[4].numpy(), want[4])
    # per-frame files under the reference's names, means of the files == rows (extract_npy2mat.py:121-126)
    for i in range(n):
        path = os.path.join(out_dir, sampling.feature_file_name(i, "resnet50"))
        assert os.path.exists(path) == (i not in (3, 5, 6))
        if i not in (3, 5, 6):
            rows = np.load(path)
            assert rows.shape == (1 + i % 4, F) and np.allclose(rows.mean(axis=0), want[i])
    m = scipy.io.loadmat(mat)["konvid_1k"]
    assert m.shape == (n, F) and m.dtype == np.float64 and np.array_equal(m, want.astype(np.float64), equal_nan=True)
    # resume: nothing healthy is recomputed, the failures are tried again
    eng2 = FakeEngine()
    again, errors2 = dataset.extract_dataset_clips(_clip, n, eng2, clips_per_step=4, out_dir=out_dir, skip_existing=True,
                                                   rank=0, world=1)
    assert sum(eng2.batches) == 1 and [i for i, _ in errors2] == [3, 5, 6]      # only clip 5 reaches the engine (and fails)
    assert np.array_equal(again.numpy(), matrix.numpy(), equal_nan=True)   # resumed rows: the bits of the first run's rows


@pytest.mark.parametrize("prefetch", [0, 2])
def test_a_per_frame_file_that_cannot_be_written_costs_only_its_clip(tmp_path, monkeypatch, prefetch):
    """A failing save_clip_features (disk full, permissions) becomes that clip's error entry - inline and from the writer pool - and the
    pass still returns every row and reaches the collective (a raise here would leave the other ranks waiting in the all-gather)."""
    _patched()
    real = sampling.save_clip_features

    def flaky(out_dir, i, network_name, arr):
        if i == 2:
            raise OSError(28, "No space left on device")
        return real(out_dir, i, network_name, arr)

    monkeypatch.setattr(sampling, "save_clip_features", flaky)
    healthy = lambda i: _clip([0, 1, 2, 4, 7, 8][i])            # noqa: E731 - none of the poisoned clips
    matrix, errors = dataset.extract_dataset_clips(healthy, 6, FakeEngine(), clips_per_step=4, out_dir=str(tmp_path / "f"),
                                                   network_name="resnet50", rank=0, world=1, prefetch=prefetch)
    assert [i for i, _ in errors] == [2] and "per-frame file not written" in errors[0][1] and "No space left" in errors[0][1]
    assert not torch.isnan(matrix).any()                        # the row itself was computed
    assert not os.path.exists(os.path.join(str(tmp_path / "f"), sampling.feature_file_name(2, "resnet50")))
    assert os.path.exists(os.path.join(str(tmp_path / "f"), sampling.feature_file_name(3, "resnet50")))


def test_a_loader_with_an_alloc_parameter_decodes_into_the_staging_pool(tmp_path):
    """clips(i, alloc=f): the driver hands the loader its staging allocator, the loader fills f(shape) and returns it; nothing is copied
    on the host (staged_bytes stays 0) and the rows equal those of the plain loader.  sampling.load_clip_from_frames is such a loader."""
    _patched()
    asked = []

    def loader(i, alloc):
        c = np.asarray(_clip([0, 1, 2, 4, 7, 8][i]))
        out = alloc(c.shape)
        assert out.dtype == np.uint8 and out.shape == c.shape
        asked.append(i)
        np.copyto(out, c)
        return out

    t = {}
    got, errors = dataset.extract_dataset_clips(loader, 6, FakeEngine(), clips_per_step=4, rank=0, world=1, prefetch=2, workers=3, timings=t)
    want, _ = dataset.extract_dataset_clips(lambda i: _clip([0, 1, 2, 4, 7, 8][i]), 6, FakeEngine(), clips_per_step=4, rank=0, world=1)
    assert not errors and sorted(asked) == list(range(6)) and torch.equal(got, want) and t["staged_bytes"] == 0
    # frames on disk -> the same protocol through sampling.load_clip_from_frames
    from PIL import Image
    rng = np.random.default_rng(3)
    clip = rng.integers(0, 256, (2, 2, 32, 48, 3), dtype=np.uint8)
    for n in range(2):
        Image.fromarray(clip[n, 0][..., ::-1]).save(tmp_path / f"v_{n}.png")
        Image.fromarray(clip[n, 1][..., ::-1]).save(tmp_path / f"v_{n}_next.png")
    seen = []

    def alloc(shape):
        seen.append(tuple(shape))
        return np.empty(shape, dtype=np.uint8)

    out = sampling.load_clip_from_frames(str(tmp_path), "v", alloc=alloc)
    assert seen == [clip.shape] and np.array_equal(out, clip) and np.array_equal(sampling.load_clip_from_frames(str(tmp_path), "v"), clip)


def test_sequence_input_and_empty_shards():
    _patched()
    clips = [_clip(i) for i in (0, 1)]
    m, e = dataset.extract_dataset_clips(clips, 2, FakeEngine(), rank=0, world=1)
    assert m.shape == (2, F) and not e


@pytest.mark.parametrize("prefetch,workers", [(0, 1), (1, 1), (2, 3), (5, 8)])
def test_prefetch_depths_give_the_same_matrix_and_errors(prefetch, workers):
    """The overlapped pass (loader threads, `prefetch` batches ahead) and the inline pass (prefetch = 0) return the same rows and
    the same error list; a loader that raises (clip 3) or returns a malformed clip (6) costs its own row only, and nothing hangs."""
    _patched()
    eng = FakeEngine()
    matrix, errors = dataset.extract_dataset_clips(_clip, 10, eng, clips_per_step=3, rank=0, world=1, prefetch=prefetch, workers=workers)
    assert np.array_equal(matrix.numpy(), _want(10), equal_nan=True)
    assert [i for i, _ in errors] == [3, 5, 6]
    assert eng.batches[0] == 3                               # whole batches reach the engine


def test_slow_and_failing_loaders_do_not_deadlock_the_pass():
    import time
    _patched()

    def slow(i):
        time.sleep(0.02 * (i % 3))
        if i % 4 == 1:
            raise RuntimeError(f"decoder crashed on clip {i}")
        return _clip(8)

    timings = {}
    matrix, errors = dataset.extract_dataset_clips(slow, 13, FakeEngine(), clips_per_step=2, rank=0, world=1, prefetch=3, workers=2,
                                                   timings=timings)
    assert [i for i, _ in errors] == [1, 5, 9] and all("decoder crashed" in m for _, m in errors)
    ok = [i for i in range(13) if i % 4 != 1]
    assert np.isnan(matrix.numpy()[[1, 5, 9]]).all() and np.isfinite(matrix.numpy()[ok]).all()
    assert timings["loader_wait_s"] >= 0 and timings["h2d_bytes"] == 0


def test_the_pass_opens_with_short_batches_and_returns_the_same_rows():
    """ramp: B/8, 3B/8, B/2 (together one full batch), then full batches (the loading of the first batch has nothing to hide under); rows and error list as
    without it."""
    _patched()
    src = lambda i: _clip(8 + (i % 3) * 4)                      # noqa: E731  (healthy clips only)
    e1, e2 = FakeEngine(), FakeEngine()
    a, ea = dataset.extract_dataset_clips(src, 70, e1, clips_per_step=16, rank=0, world=1)
    b, eb = dataset.extract_dataset_clips(src, 70, e2, clips_per_step=16, rank=0, world=1, ramp=False)
    assert e1.batches == [2, 6, 8, 16, 16, 16, 6] and e2.batches == [16, 16, 16, 16, 6]     # the same tail with and without the ramp
    assert np.array_equal(a.numpy(), b.numpy()) and not ea and not eb


def test_clips_per_step_must_be_positive():
    _patched()
    for bad in (0, -3):
        with pytest.raises(ValueError, match="clips_per_step"):
            dataset.extract_dataset_clips(_clip, 4, FakeEngine(), clips_per_step=bad, rank=0, world=1)


def test_resume_recomputes_a_truncated_file_instead_of_failing_forever(tmp_path):
    """A run killed mid-write used to leave a truncated .npy that every resume turned into a NaN row (round-3 advice).  Files are
    now written under a temporary name and renamed; a file that still cannot be read, or has the wrong width, is recomputed and
    overwritten."""
    _patched()
    out_dir = str(tmp_path / "feats")
    first, _ = dataset.extract_dataset_clips(_clip, 3, FakeEngine(), clips_per_step=2, out_dir=out_dir, rank=0, world=1)
    assert not [f for f in os.listdir(out_dir) if f.endswith(".tmp")]
    p1 = os.path.join(out_dir, sampling.feature_file_name(1, "resnet50"))
    p2 = os.path.join(out_dir, sampling.feature_file_name(2, "resnet50"))
    with open(p1, "r+b") as f:
        f.truncate(70)                                       # header survives, data gone
    np.save(p2, np.zeros((2, F + 1), np.float32))            # a foreign file of the wrong width
    eng = FakeEngine()
    again, errors = dataset.extract_dataset_clips(_clip, 3, eng, clips_per_step=2, out_dir=out_dir, skip_existing=True, rank=0, world=1)
    assert not errors and sum(eng.batches) == 2              # clips 1 and 2 went through the engine again, clip 0 did not
    assert np.allclose(again.numpy(), first.numpy())
    assert np.load(p1).shape == (2, F) and np.load(p2).shape == (3, F)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import relax_vqa_amd  # noqa: F401
    from relax_vqa_amd import dataset as ds
    from relax_vqa_amd import distributed as rd
    rd.init_from_env(backend="gloo")
    ds.feature_dim = lambda engine, resnet=True, vit=True, full=False: F
    eng = FakeEngine()
    # rank / world from the process group; eight ranks: loader threads on, as a node's eight ranks would run (prefetch 2, 2 workers each)
    matrix, errors = ds.extract_dataset_clips(_clip, n, eng, clips_per_step=2 if world <= 2 else 3, **({} if world <= 2 else dict(prefetch=2, workers=2)))
    q.put((rank, matrix.numpy().copy(), errors, sum(eng.batches)))
    rd.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n", [1, 9])
def test_two_ranks_equal_one_rank_bit_for_bit(n):
    """n = 1: rank 1's shard is empty and the run still completes (a dataset smaller than the node must not hang the gather)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _want(n)
    for rank, matrix, errors, _ in results:
        assert np.array_equal(matrix, want, equal_nan=True), f"rank {rank}"
        assert [i for i, _ in errors] == [i for i in (3, 5, 6) if i < n]              # every rank holds the whole error list


def test_eight_ranks_finish_a_64_clip_pass_with_their_loaders_running():
    """The shape of a full node: eight processes (gloo), each with its own loader threads, pinned-pool bookkeeping and shard of a
    64-clip list; every rank returns the whole matrix and the whole error list, equal to the single-rank result bit for bit."""
    n, world = 64, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = _want(n)
    assert sorted(r[0] for r in results) == list(range(world))
    for rank, matrix, errors, _ in results:
        assert np.array_equal(matrix, want, equal_nan=True), f"rank {rank}"
        assert [i for i, _ in errors] == [3, 5, 6]


def test_pinned_budget_is_granted_in_clip_order_and_never_locks_up():
    """dataset._PinnedGate: 12 'loader threads' ask for 40 bytes each under a cap of 100, in scrambled arrival order, while a 'copy
    engine' hands bytes back in clip order: every clip is served, in order, never more than the cap is out - and a clip larger than
    the cap still goes through once nothing else is held.  Clips that need nothing (errors, resumed rows) give their turn up."""
    import random
    import threading
    gate = dataset._PinnedGate(100)
    gate.start_pass()
    order, lock = [], threading.Lock()
    sizes = {k: (250 if k == 7 else 40) for k in range(12)}
    skipped = {3, 9}

    def loader(k):
        time.sleep(random.Random(k).random() * 0.05)
        if k in skipped:
            gate.skip(k)
            return
        gate.acquire(k, sizes[k])
        with lock:
            order.append(k)
            assert gate.in_use <= max(gate.cap, sizes[k])

    threads = [threading.Thread(target=loader, args=(k,)) for k in random.Random(1).sample(range(12), 12)]
    for t in threads:
        t.start()
    released = 0
    deadline = time.time() + 30
    while released < 10 and time.time() < deadline:          # the copy engine: releases in the order the clips were served
        with lock:
            todo = order[released:]
        for k in todo:
            time.sleep(0.005)
            gate.release(sizes[k])
            released += 1
        time.sleep(0.002)
    for t in threads:
        t.join(timeout=5)
        assert not t.is_alive()
    assert order == [k for k in range(12) if k not in skipped] and gate.in_use == 0 and gate.peak <= 250


def _gated_engine(cap):
    """A FakeEngine whose stager applies the pinned budget although the 'device' is the CPU (what a GPU run does; the buffers are
    plain host memory here)."""
    eng = FakeEngine()
    st = dataset.ClipStager(eng.device, 1)
    st.gated = True
    st.pool_limit_bytes = cap
    st.gate.cap = cap
    eng._clip_stager = st
    return eng, st


def test_alloc_buffers_of_clips_that_fail_after_loading_go_back_to_the_budget():
    """A loader that decodes into `alloc` memory and then returns something the driver refuses (a malformed clip, a copy, two
    allocations) must not keep its bytes: after the pass nothing is in use, and a second pass over the same stager runs."""
    _patched()
    ids = [0, 1, 6, 2, 4, 6, 7, 8]                              # _clip(6) is malformed ([T,1,...]): _check_clip refuses it AFTER the decode

    def loader(i, alloc):
        c = np.asarray(_clip(ids[i]))
        out = alloc(c.shape)
        np.copyto(out, c)
        if i == 3:                                              # a second allocation by the same clip, and a COPY returned: both go back
            alloc(c.shape)
            return out.copy()
        return out

    n_bytes = max(int(np.prod(np.asarray(_clip(k)).shape)) for k in ids)
    eng, st = _gated_engine(9 * n_bytes)                        # (a CPU 'device' holds a batch's buffers until its compute is through - a GPU
    for _ in range(2):                                          # hands them back as each copy lands -: room for prefetch + 1 batches)
        got, errors = dataset.extract_dataset_clips(loader, len(ids), eng, clips_per_step=3, rank=0, world=1, prefetch=2, workers=3)
        assert [i for i, _ in errors] == [2, 5]
        assert st.gate.in_use == 0, "pinned bytes leaked"
        assert not torch.isnan(got[[0, 1, 3, 4, 6, 7]]).any()
    # a clip whose two allocations exceed the cap on their own does not wait for itself
    gate = dataset._PinnedGate(100)
    gate.start_pass()
    gate.acquire(0, 80)
    gate.acquire(None, 80, own=80)                              # (without `own` this would wait for in_use == 0 forever)
    assert gate.in_use == 160
    gate.close()
    with pytest.raises(RuntimeError):
        gate.acquire(1, 80)


def test_an_exception_in_the_driver_thread_does_not_hang_on_loaders_waiting_for_pinned_memory():
    """Something escapes the driver loop (here: the engine's rows_mean is missing when a resumed clip comes up ... a HIP error would
    do the same) while loader threads wait at the gate for memory only the driver could hand back: the pass must raise, not hang,
    and the next pass over the same stager starts with a clean budget."""
    _patched()

    class Boom(BaseException):
        pass

    class BadEngine(FakeEngine):
        def clip_vectors(self, clips, **kw):
            raise Boom("escapes `except Exception`")

    def loader(i, alloc):
        c = np.asarray(_clip(0))
        out = alloc(c.shape)
        np.copyto(out, c)
        return out

    n_bytes = int(np.prod(np.asarray(_clip(0)).shape))
    eng = BadEngine()
    st = dataset.ClipStager(eng.device, 1)
    st.gated = True
    st.pool_limit_bytes = st.gate.cap = 3 * n_bytes             # three clips' worth: the other loaders of the 12 wait at the gate
    eng._clip_stager = st
    done = []

    def run():
        try:
            dataset.extract_dataset_clips(loader, 12, eng, clips_per_step=2, rank=0, world=1, prefetch=3, workers=4, ramp=False)
        except Boom:
            done.append("raised")

    import threading
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(timeout=20)
    assert not th.is_alive(), "the pass hangs in pool.shutdown behind loaders waiting for pinned memory"
    assert done == ["raised"]
    good = FakeEngine()
    good._clip_stager = st
    st.pool_limit_bytes = st.gate.cap = 8 * n_bytes             # (a CPU 'device' keeps a batch's buffers until its compute is through)
    got, errors = dataset.extract_dataset_clips(loader, 4, good, clips_per_step=2, rank=0, world=1, prefetch=2, workers=2)
    assert not errors and st.gate.in_use == 0 and not torch.isnan(got).any()


def test_host_placement_helpers(tmp_path, monkeypatch):
    """hostnode.py against a stand-in /sys tree: the GPU's NUMA node, that node's CPUs, the per-rank share of the pinned budget; and
    'do nothing' when the platform does not tell."""
    from relax_vqa_amd import hostnode
    assert hostnode.parse_cpulist("0-3,8,10-11") == {0, 1, 2, 3, 8, 10, 11} and hostnode.parse_cpulist("") == set()
    dev = tmp_path / "bus" / "pci" / "devices" / "0000:c1:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    node = tmp_path / "devices" / "system" / "node" / "node1"
    node.mkdir(parents=True)
    allowed = sorted(os.sched_getaffinity(0))
    (node / "cpulist").write_text(f"{allowed[0]}-{allowed[-1]}\n")
    monkeypatch.setattr(hostnode, "pci_address", lambda i: "0000:c1:00.0")
    assert hostnode.gpu_numa_node(0, sysfs=str(tmp_path)) == 1
    assert hostnode.loader_cpus(0, sysfs=str(tmp_path)) == set(allowed)
    (dev / "numa_node").write_text("-1\n")
    assert hostnode.gpu_numa_node(0, sysfs=str(tmp_path)) is None and hostnode.loader_cpus(0, sysfs=str(tmp_path)) is None
    monkeypatch.setenv("RELAX_NUMA_BIND", "0")
    (dev / "numa_node").write_text("1\n")
    assert hostnode.loader_cpus(0, sysfs=str(tmp_path)) is None
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    monkeypatch.delenv("RELAX_PINNED_POOL_GB", raising=False)
    assert hostnode.pinned_pool_budget() == 8 << 30                    # 64 GiB per node over eight ranks
    monkeypatch.setenv("RELAX_PINNED_POOL_GB", "16")
    assert hostnode.pinned_pool_budget() == 2 << 30
    assert hostnode.bind_this_thread(set()) is False and hostnode.bind_this_thread(set(allowed)) is True
