"""Loader PROCESSES (relax-vqa_amd/loaderpool.py) under the dataset driver, on CPU with the stand-in engine of test_dataset_cpu.py: the
matrix equals the thread path bit for bit, failures cost their clip only, a dead worker does not hang the pass, shared memory is
bounded and gone after close()."""
import os

import numpy as np
import pytest
import torch

import relax_vqa_amd  # noqa: F401
from relax_vqa_amd import dataset
from relax_vqa_amd.loaderpool import LoaderProcessPool
from tests import loader_sources
from tests.test_dataset_cpu import F, FakeEngine, _patched


def _shm_names():
    return sorted(f for f in os.listdir("/dev/shm") if f.startswith("relaxldr_")) if os.path.isdir("/dev/shm") else []


def _thread_path(n):
    def src(i):
        if i == 3:
            raise OSError(f"cannot decode video_{i + 1}.mp4")
        c = loader_sources.clip_array(i)
        return c[:, :1] if i == 6 else c
    return dataset.extract_dataset_clips(src, n, FakeEngine(), clips_per_step=4, rank=0, world=1, prefetch=2, workers=3)


@pytest.mark.parametrize("src", [loader_sources.source, loader_sources.plain_source], ids=["alloc", "plain"])
def test_process_pool_gives_the_matrix_of_the_thread_path(src):
    _patched()
    n = 14
    before = _shm_names()
    with LoaderProcessPool(src, processes=3, segments_per_worker=2).start() as pool:
        t = {}
        got, errors = dataset.extract_dataset_clips(pool, n, FakeEngine(), clips_per_step=4, rank=0, world=1, prefetch=2, workers=6, timings=t)
        again, errors2 = dataset.extract_dataset_clips(pool, n, FakeEngine(), clips_per_step=4, rank=0, world=1, prefetch=1, workers=2)
        in_flight = _shm_names()
        assert 0 < len(in_flight) - len(before) <= 3 * 2, "shared memory is bounded by workers x segments"
    assert _shm_names() == before, "segments left in /dev/shm after close()"
    if src is loader_sources.source:
        want, want_errors = _thread_path(n)
        assert [i for i, _ in errors] == [3, 6] == [i for i, _ in want_errors] and "cannot decode" in errors[0][1]
        assert torch.equal(got[[i for i in range(n) if i not in (3, 6)]], want[[i for i in range(n) if i not in (3, 6)]])
    else:
        assert not errors
        want = torch.stack([FakeEngine._rows(loader_sources.clip_array(i)).mean(dim=0) for i in range(n)])
        assert torch.equal(got, want)
    assert torch.equal(torch.nan_to_num(got), torch.nan_to_num(again)) and [i for i, _ in errors] == [i for i, _ in errors2]
    assert t["staged_bytes"] == 0, "a clip of the pool was copied on the host"


def test_a_dead_worker_costs_its_clips_only(monkeypatch):
    _patched()
    monkeypatch.setenv("RELAX_TEST_KILL_WORKER", "1")          # (spawned workers inherit the environment)
    n = 16
    with LoaderProcessPool(loader_sources.source, processes=3).start() as pool:
        got, errors = dataset.extract_dataset_clips(pool, n, FakeEngine(), clips_per_step=4, rank=0, world=1, prefetch=2, workers=6)
    bad = [i for i, _ in errors]
    assert 11 in bad and 3 in bad and 6 in bad and any("died" in m for _, m in errors)
    healthy = [i for i in range(n) if i not in bad]
    assert len(healthy) >= n - 6 and not torch.isnan(got[healthy]).any()
    assert _shm_names() == [] or all("relaxldr" not in f for f in _shm_names())


def test_the_pool_refuses_to_start_after_the_gpu_was_touched(monkeypatch):
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "is_initialized", lambda: True)
    with pytest.raises(RuntimeError, match="before this process touches a GPU"):
        LoaderProcessPool(loader_sources.plain_source, processes=1).start()
