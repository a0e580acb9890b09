"""Error behaviour of the C-ABI (status codes + messages -> RuntimeError) and the RCCL all-gather path on one GPU."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch

from tests.gpu_common import engine, synth

pytestmark = pytest.mark.gpu


def test_errors_are_reported_not_swallowed():
    eng = engine()
    lib, h = eng.lib, eng.h
    assert lib.relax_fragment_pairs(h, None, None, 0, 1, 16, 16, 196, None, None, None, None, None, None) == -1
    assert b"NULL" in lib.relax_last_error(h)
    f = torch.zeros((1, 2, 32, 32, 3), dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError, match="top_n"):
        eng.fragment_pairs(f, top_n=500)
    with pytest.raises(ValueError):
        eng.fragment_pairs(torch.zeros((1, 3, 32, 32, 3), dtype=torch.uint8))
    with pytest.raises(RuntimeError, match="multiple of 16"):       # bf16x6 kernel (default): K in 16-deep chunks
        eng.op_gemm(torch.zeros(8, 40, device="cuda"), torch.zeros(64, 40, device="cuda"))
    eng.set_precision("fp32")
    with pytest.raises(RuntimeError, match="multiple of 32"):       # exact-fp32 kernel
        eng.op_gemm(torch.zeros(8, 40, device="cuda"), torch.zeros(64, 40, device="cuda"))
    eng.set_precision("bf16x6")
    with pytest.raises(RuntimeError, match="multiple of 64"):
        eng.op_gemm(torch.zeros(8, 64, device="cuda"), torch.zeros(48, 64, device="cuda"))
    with pytest.raises(RuntimeError, match="unknown option"):
        eng.set_option("no_such_option", 1)
    sd = synth.resnet50_state_dict()
    bad = dict(sd)
    del bad["layer3.4.conv2.weight"]
    from relax_vqa_amd.engine import RelaxEngine
    e2 = RelaxEngine(0)
    with pytest.raises(RuntimeError, match="missing key 'layer3.4.conv2.weight'"):
        e2.load_resnet50(bad)
    with pytest.raises(RuntimeError, match="relax_load_resnet50 first"):
        e2.resnet50_features(torch.zeros((1, 224, 224, 3), dtype=torch.uint8))
    bad = dict(sd)
    bad["conv1.weight"] = np.zeros((64, 3, 5, 5), np.float32)
    with pytest.raises(RuntimeError, match="conv1.weight"):
        e2.load_resnet50(bad)
    e2.close()
    assert lib.relax_create(99, C.byref(C.c_void_p())) != 0 and b"out of range" in lib.relax_last_error(None)


def test_rccl_all_gather_single_rank():
    """backend 'nccl' is RCCL on ROCm: run the feature all-gather through it (world size 1 is all one box offers)."""
    import torch.distributed as dist
    from relax_vqa_amd import distributed as rd
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        local = torch.randn(3, 19779, device="cuda")
        out = rd.gather_clip_vectors(local, 3, 0, 1)
        torch.cuda.synchronize()
        assert torch.equal(out, local)
    finally:
        dist.destroy_process_group()


def test_dataset_driver_under_rccl_single_rank():
    """The dataset driver with an initialised nccl (= RCCL) process group: the device all_gather_into_tensor, the error-list
    all_gather_object and the nccl barrier (device named) are the branches an 8-GPU run takes; world size 1 is all one box offers."""
    import torch.distributed as dist
    from relax_vqa_amd import dataset
    from relax_vqa_amd import distributed as rd
    from tests.gpu_common import rn50_weights, vit_weights
    rn50_weights(), vit_weights("vit_base")
    eng = engine()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        clips = [synth.synthetic_clip(2, 240, 320, clip_id=70 + i) for i in range(3)]
        src = lambda i: None if i == 1 else clips[i]      # noqa: E731
        matrix, errors = dataset.extract_dataset_clips(src, 3, eng, clips_per_step=2)      # rank / world from the process group
        rd.barrier()
        torch.cuda.synchronize()
        assert matrix.shape == (3, 19779) and [i for i, _ in errors] == [1]
        assert bool(torch.isnan(matrix[1]).all()) and bool(torch.isfinite(matrix[[0, 2]]).all())
    finally:
        dist.destroy_process_group()


def test_pinned_feeder_overlaps_and_preserves_data():
    from relax_vqa_amd.feeder import PinnedClipFeeder
    clips = [torch.from_numpy(synth.synthetic_clip(2, 64, 96, clip_id=50 + i)).pin_memory() for i in range(3)]
    feeder = PinnedClipFeeder(clips, 2, torch.device("cuda", 0))
    seen = []
    feeder.run(4, lambda batch: seen.append([b.clone() for b in batch]))
    torch.cuda.synchronize()
    for k, batch in enumerate(seen):
        for j, b in enumerate(batch):
            assert torch.equal(b.cpu(), clips[(k * 2 + j) % 3]), (k, j)
    # a second run on the same feeder, its first batch started ahead of time (bench.py's steady-state timing): same batches
    again = []
    feeder.prime()
    feeder.run(3, lambda batch: again.append([b.clone() for b in batch]))
    torch.cuda.synchronize()
    for k, batch in enumerate(again):
        for j, b in enumerate(batch):
            assert torch.equal(b.cpu(), clips[(k * 2 + j) % 3]), (k, j)


def test_bench_two_ranks_sharing_the_gpu():
    """The driver's N > 1 launch line on a 1-GPU box: two ranks under torch.distributed.run share the device (gloo, host-
    staged collective, since RCCL refuses duplicate devices).  Checks the multi-rank control flow of bench.py end to end:
    one JSON line from rank 0, n_gpus = world, every rank's clips in the gathered matrix."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from tests.gpu_common import run_ranks
    env = dict(os.environ, RELAX_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = run_ranks(lambda port: [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                                  "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1",
                                  "--warmup", "1", "--workload", "config2", "--clips-per-step", "1"], root, env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert "roofline" in out and "cpu_baseline" not in out          # the CPU baseline is an N = 1 leg
    assert out["ranks"] == 2 and out["backend"] == "gloo" and out["rccl_ranks"] == 0


def test_bench_gpus_2_run_plainly_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the shape of the driver's N = 1 command line with a larger N):
    bench.py starts torch.distributed.run itself, as a child, before touching the GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RELAX_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--workload", "config2",
           "--clips-per-step", "1"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["backend"] == "gloo" and out["rccl_ranks"] == 0 and out["value"] > 0


# ---- real RCCL: arms itself wherever the box has two GPUs (the pool's boxes have one: skipped there, with the reason printed) ----
_TWO_GPUS = torch.cuda.is_available() and torch.cuda.device_count() >= 2
_NEED_TWO = pytest.mark.skipif(not _TWO_GPUS, reason="real-RCCL test: needs >= 2 GPUs on the box (torch.cuda.device_count() < 2 here)")


def _bench_plain(args, timeout=1200):
    """`python bench.py ...` with no launcher and no backend override: N > 1 starts its own ranks under the default nccl (= RCCL)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "RELAX_DIST_BACKEND")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, cwd=root, env=env, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    return json.loads(lines[0])


@_NEED_TWO
def test_rccl_two_gpus_dataset_matrix_equals_one_rank_bit_for_bit(tmp_path):
    """Config 4 as written over two GPUs under RCCL: contiguous shards, one all_gather_into_tensor over xGMI; with the tail split off
    the [7, 19779] matrix is the one-rank matrix bit for bit (src/main_fragment_layerstack.py:269 has no cross-clip state)."""
    import numpy as np
    common = ["--workload", "config4", "--dataset-clips", "7", "--clips-per-step", "2", "--warmup", "1", "--gemm-split-k", "0",
              "--resident-clips", "3"]
    one = _bench_plain(["--gpus", "1"] + common + ["--dump-matrix", str(tmp_path / "one.npy")])
    two = _bench_plain(["--gpus", "2"] + common + ["--dump-matrix", str(tmp_path / "two.npy")])
    assert one["rccl_ranks"] == 1 and two["rccl_ranks"] == 2 and two["backend"] == "nccl" and two["ranks"] == 2
    a, b = np.load(tmp_path / "one.npy"), np.load(tmp_path / "two.npy")
    assert a.shape == (7, 19779) and np.isfinite(a).all() and np.array_equal(a, b), "sharding over two GPUs changed the matrix"


@_NEED_TWO
def test_rccl_two_gpus_weak_scaling_line():
    out = _bench_plain(["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "config3", "--clips-per-step", "4"])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["backend"] == "nccl" and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["clips_per_step_per_gpu"] == 4 and "roofline" in out
