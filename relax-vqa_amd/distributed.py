"""Clip-sharded data parallelism: one process per GPU, clips are independent (the reference's outer loop
`for i in range(len(videodata))`, src/main_fragment_layerstack.py:269, carries no state), weights replicated,
and ONE collective: an all-gather of the per-clip feature vectors so that every rank holds the [n_clips, F]
matrix the scaler/MLP consume (the matrix src/data_processing/extract_npy2mat.py:117-130 builds on disk).
Backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # RELAX_DIST_BACKEND=gloo: several ranks sharing ONE GPU (RCCL refuses duplicate devices) to rehearse the
            # multi-rank control flow on a single-GPU box; collectives are then staged through host memory
            backend = os.environ.get("RELAX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            local_rank = local_rank % max(torch.cuda.device_count(), 1) if backend == "gloo" else local_rank
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def barrier():
    """dist.barrier(); under nccl (RCCL) the device is named so that the barrier's collective runs on this rank's GPU and not
    on whichever device torch guesses (a warning today, a hang when the guess is another rank's GPU)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    if dist.get_backend() == "nccl" and torch.cuda.is_available():
        dist.barrier(device_ids=[torch.cuda.current_device()])
    else:
        dist.barrier()


def gather_objects(obj, world, group=None):
    """-> [obj of rank 0, ..., obj of rank world-1] on every rank (small python objects: the per-rank error lists)."""
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj, group=group)
    return out


def _host_staged(t):
    """gloo has no device collectives for every op: stage CUDA tensors through host memory under that backend."""
    return t.is_cuda and dist.get_backend() == "gloo"


def all_reduce_max(value, device):
    """max over ranks of a python float (the timing reduction of bench.py)."""
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_clips(n_clips, rank, world):
    """Contiguous block partition of clip indices; the first n_clips % world ranks get one extra."""
    base, extra = divmod(n_clips, world)
    lo = rank * base + min(rank, extra)
    return list(range(lo, lo + base + (1 if rank < extra else 0)))


def gather_clip_vectors(local, n_clips, rank, world, group=None):
    """local: [len(shard_clips(n_clips, rank, world)), F] -> [n_clips, F] on every rank, rows in clip order.
    Ragged shards are padded to ceil(n_clips / world) rows for a single fixed-size all-gather."""
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return local
    per = -(-n_clips // world)
    F = local.shape[1]
    padded = torch.zeros((per, F), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    if _host_staged(padded):
        out_h = torch.empty((world * per, F), dtype=local.dtype)
        dist.all_gather_into_tensor(out_h, padded.cpu(), group=group)
        out = out_h.to(local.device)
    else:
        out = torch.empty((world * per, F), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, padded, group=group)
    rows = []
    for r in range(world):
        n_r = len(shard_clips(n_clips, r, world))
        rows.append(out[r * per: r * per + n_r])
    return torch.cat(rows, dim=0)


def extract_dataset(extract_fn, n_clips, rank, world, group=None):
    """Run extract_fn(clip_index) -> [F] on this rank's shard and all-gather the per-clip vectors."""
    if n_clips < world:
        # decided from the arguments alone, so EVERY rank raises (an error on the empty ranks only would leave the others
        # waiting in the all-gather until the collective times out)
        raise ValueError(f"every rank needs at least one clip: n_clips={n_clips} < world={world}")
    mine = shard_clips(n_clips, rank, world)
    local = torch.stack([extract_fn(i) for i in mine])
    return gather_clip_vectors(local, n_clips, rank, world, group)
