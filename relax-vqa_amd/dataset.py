"""Dataset driver: BASELINE config 4 as written - a list of clips sharded over the ranks of one node, every rank pushing its
shard through the engine in batches, ONE all-gather of the per-clip vectors at the end.

The reference's counterpart is the per-video loop of its drivers (`for i in range(len(videodata))`,
src/main_fragment_layerstack.py:269-361: sample frames, fragments, features, `np.save` of the per-frame [T,F] array under
`video_{i+1}_{network}_feature_map_original.npy`) followed by src/data_processing/extract_npy2mat.py:117-130 (mean over the
frames of every file into one [n_videos, F] matrix, saved as a .mat keyed by the dataset name).  Here:

  rank r owns the contiguous block `shard_clips(n_clips, r, world)` of clip indices          (distributed.py)
  batches of `clips_per_step` clips go through RelaxEngine.clip_vectors / full_clip_vectors   (one pass of each backbone)
  the pass is HOST-FED and OVERLAPPED: `workers` loader threads call clips(i) for the next `prefetch` batches (decode / read /
    resume-from-file happen there) and land host clips in pinned staging buffers - a loader that takes an `alloc` argument
    decodes STRAIGHT into one (no pageable copy of the clip ever exists); a side HIP stream copies batch k+1 into the other of
    two device slots while the engine works on batch k; events order the two streams both ways               (ClipStager)
  the loader threads of a rank run on the CPUs of its GPU's NUMA node and allocate its pinned buffers there, and the pinned pool a
    rank keeps is its share of a per-node budget                                                               (hostnode.py)
  optional: the per-frame [T,F] rows of every clip are written under the reference's file name (resume: skip_existing)
  the [n_local, F] means are all-gathered into the [n_clips, F] matrix every rank returns     (RCCL over xGMI; gloo on CPU)
  optional: rank 0 writes the .mat the reference's regression scripts read

Failure contract: a clip that cannot be loaded or extracted does not take its batch - or the run - down.  Its
row in the matrix is NaN and (clip index, message) goes into the returned error list (gathered over the ranks, sorted);
the reference's imputer zeroes NaN / inf before the regressor (src/model_regression.py:123-126), relax_mlp_head imputes
them with the training means.  A loader thread that raises yields exactly that entry for its clip; nothing waits on it forever.
"""
import inspect
import os
import threading
import time
from concurrent.futures import ThreadPoolExecutor
from concurrent.futures import TimeoutError as FutureTimeout

import numpy as np
import torch

from . import distributed as rdist
from . import hostnode
from . import sampling
from .engine import LAYER_STACK_DIM, RN50_POOL_DIM

FULL_DIM = 35203


def feature_dim(engine, resnet=True, vit=True, full=False):
    if full:
        return FULL_DIM
    return (LAYER_STACK_DIM + RN50_POOL_DIM if resnet else 0) + (6 * engine.vit_dim if vit else 0)


def _check_clip(clip):
    """The argument errors the engine would raise for a whole batch, found per clip before the batch is assembled."""
    if not isinstance(clip, (torch.Tensor, np.ndarray)):
        raise TypeError(f"a clip must be a uint8 array [T,2,H,W,3], got {type(clip).__name__}")
    shape = tuple(clip.shape)
    if len(shape) != 5 or shape[1] != 2 or shape[4] != 3:
        raise ValueError(f"clip must be [T,2,H,W,3], got {shape}")
    if str(clip.dtype).replace("torch.", "") != "uint8":
        raise TypeError(f"clip must be uint8, got {clip.dtype}")
    if shape[0] == 0:
        raise ValueError("clip has no (frame, next) pair")
    if shape[2] < 16 or shape[3] < 16:
        raise ValueError(f"frames of {shape[3]}x{shape[2]} hold no 16x16 patch")


class _PinnedGate:
    """The byte budget of a rank's pinned staging memory, granted IN CLIP ORDER.  A loader thread asks for its clip's bytes before it
    decodes; it waits until (a) every earlier clip of the pass has been served or has declared that it needs nothing (`skip`) and
    (b) the bytes fit under the cap - or nothing at all is held (one clip larger than the cap still goes through).  Bytes come back
    when the clip's host -> device copy has landed.  Because the grant follows the clip order, the memory is always held by the
    earliest clips - the ones the driver copies next - and a pass cannot lock itself up however small the cap; loaders that run too
    far ahead of the copy engine simply wait (back-pressure instead of unbounded pinned memory)."""

    def __init__(self, cap):
        self.cap = int(cap)
        self.in_use = 0
        self.peak = 0
        self.closed = False
        self._next = 0
        self._served = set()
        self._cv = threading.Condition()

    def start_pass(self):
        """A new pass: no loader of an earlier pass is alive any more (the pass's `finally` shut its pool down and handed every buffer
        back), so whatever a failed pass left in the books is dropped here rather than shrinking the budget for good."""
        with self._cv:
            self._next, self._served = 0, set()
            self.in_use = 0
            self.closed = False
            self._cv.notify_all()

    def close(self):
        """The pass is over (or is being torn down after an error in the driver thread): nobody will hand bytes back any more, so a
        loader still waiting for its turn must not wait forever - acquire raises from now on."""
        with self._cv:
            self.closed = True
            self._cv.notify_all()

    def _advance(self, seq):
        if seq >= self._next:
            self._served.add(seq)
            while self._next in self._served:
                self._served.remove(self._next)
                self._next += 1
        self._cv.notify_all()

    def acquire(self, seq, n, own=0):
        """own: bytes the asking clip holds already (a loader that calls `alloc` more than once): they do not count against "nothing
        at all is held", or a clip larger than the cap would wait for itself."""
        with self._cv:
            while not ((seq is None or seq <= self._next) and (self.in_use - own <= 0 or self.in_use + n <= self.cap)):
                if self.closed:
                    raise RuntimeError("the pass was torn down while this clip waited for pinned staging memory")
                self._cv.wait(0.05)
            if self.closed:
                raise RuntimeError("the pass was torn down while this clip waited for pinned staging memory")
            self.in_use += n
            self.peak = max(self.peak, self.in_use)
            if seq is not None:
                self._advance(seq)

    def skip(self, seq):
        with self._cv:
            self._advance(seq)

    def release(self, n):
        with self._cv:
            self.in_use = max(self.in_use - n, 0)
            self._cv.notify_all()


class ClipStager:
    """Pinned host staging + two device slots + a side stream for one (process, device).  Kept on the engine object between
    passes so that the pinned and device buffers are allocated once (hipHostMalloc of GBs is slow).

    Loader threads call `to_pinned(host_tensor)`; the driver thread calls `upload(list of clips)` for batch k+1 right after it
    has enqueued the compute of batch k, then `done_with(token)` once that batch has been consumed.  Clips already on the device
    pass through untouched; on a CPU 'device' (tests with a stand-in engine) everything passes through."""

    def __init__(self, device, world=1):
        self.device = device
        self.on_gpu = device.type == "cuda"
        self.gated = self.on_gpu                     # the pinned budget applies (tests switch it on for a CPU stand-in)
        self._lock = threading.Lock()
        self._pinned_free = {}                       # nbytes -> [pinned flat uint8 tensors]
        self._pooled_bytes = 0
        self.pool_limit_bytes = hostnode.pinned_pool_budget(world)   # pinned staging kept for reuse: this rank's share of the node's budget
        self.bytes_copied = 0                        # host -> device
        self.bytes_staged = 0                        # pageable -> pinned (zero for loaders that decode into `alloc` buffers)
        self.pinned_live_bytes = 0                   # pinned bytes this stager has allocated and not dropped (pool + handed out)
        self.pinned_peak_bytes = 0
        self.gate = _PinnedGate(self.pool_limit_bytes)   # bytes handed out to clips in flight never exceed the same budget
        self._landing = []                           # (event of the clip's host -> device copy, its pinned buffer)
        if self.on_gpu:
            self.copy_stream = torch.cuda.Stream(device=device)
            self._slots = [None, None]               # flat uint8 device buffers, grown on demand
            self._free = [torch.cuda.Event(), torch.cuda.Event()]     # compute has finished with the slot
            for e in self._free:
                e.record(torch.cuda.current_stream(device))
            self._turn = 0

    # ---- loader-thread side ------------------------------------------------------------------------------
    def _take_pinned(self, n, seq=None, own=0):
        """A flat pinned uint8 buffer of n bytes, once the budget allows it (in clip order: _PinnedGate): from the pool, else a new
        allocation - made by the CALLING (loader) thread, which runs on the CPUs of the GPU's NUMA node, so the pages land there.
        Pooled (idle) and handed-out buffers count against ONE budget: before a new allocation, idle buffers of other sizes are
        dropped until pool + in flight fit this rank's share again."""
        if self.gated:                               # (a CPU 'device' uses the clips where they lie: nothing is pinned, nothing is budgeted)
            self.gate.acquire(seq, n, own)
        with self._lock:
            free = self._pinned_free.get(n)
            buf = free.pop() if free else None
            if buf is not None:
                self._pooled_bytes -= n
            else:
                for size in sorted(self._pinned_free, reverse=True):
                    lst = self._pinned_free[size]
                    while lst and self._pooled_bytes + self.gate.in_use > self.pool_limit_bytes:
                        lst.pop()
                        self._pooled_bytes -= size
                        self.pinned_live_bytes -= size
        if buf is None:
            buf = torch.empty(n, dtype=torch.uint8, pin_memory=self.on_gpu)
            with self._lock:
                self.pinned_live_bytes += n
                self.pinned_peak_bytes = max(self.pinned_peak_bytes, self.pinned_live_bytes)
        return buf

    def alloc_pinned(self, shape, seq=None, own=0):
        """-> (uint8 tensor of `shape` over pinned memory, buffer to hand back): what a loader decodes into (the `alloc` protocol)."""
        n = int(np.prod(shape))
        buf = self._take_pinned(n, seq, own)
        return buf.view(tuple(int(d) for d in shape)), buf

    def to_pinned(self, t, seq=None):
        """host uint8 tensor -> (pinned tensor of the same shape, buffer to hand back) - a copy unless `t` is pinned already."""
        if not self.on_gpu or t.is_pinned():
            if seq is not None:
                self.gate.skip(seq)
            return t, None
        buf = self._take_pinned(t.numel(), seq)
        buf.copy_(t.reshape(-1))                     # releases the GIL: the loader threads copy in parallel
        with self._lock:
            self.bytes_staged += t.numel()
        return buf.view(t.shape), buf

    def give_back(self, bufs):
        """Buffers whose copies have landed (or that were never used) go back to the pool - their bytes back to the budget -, up to this
        rank's share of the node's pinned budget; beyond it they are dropped (unpinned and freed)."""
        shm = [b for b in bufs if b is not None and hasattr(b, "release")]   # clips of a loader-process pool: their segment goes back to its worker
        for b in shm:
            b.release()
        bufs = [b for b in bufs if b is not None and not hasattr(b, "release")]
        if self.gated:
            for b in bufs:
                self.gate.release(b.numel())
        with self._lock:
            for b in bufs:
                if b is None:
                    continue
                if self._pooled_bytes + self.gate.in_use + b.numel() <= self.pool_limit_bytes:
                    self._pinned_free.setdefault(b.numel(), []).append(b)
                    self._pooled_bytes += b.numel()
                else:
                    self.pinned_live_bytes -= b.numel()

    # ---- driver-thread side --------------------------------------------------------------------------------
    def begin(self, n_clips=0):
        """Start a batch of (at most) n_clips clips: take this turn's device slot, make the side stream wait until the batch that used
        it two turns ago is through.  Then `add(clip)` per clip as its loader finishes (the copy starts at once, under the loading of
        the batch's other clips), `end()` -> token."""
        slot = self._turn
        self._turn ^= 1
        self._cur = {"slot": slot, "at": 0, "any": False, "left": max(int(n_clips), 1)}
        self.copy_stream.wait_event(self._free[slot])
        return self._cur

    def reap(self, wait=False):
        """Hand back the pinned buffers of the clips whose host -> device copies have landed (wait: of all clips copied so far)."""
        if not self.on_gpu or not self._landing:
            return
        keep, done = [], []
        for ev, buf in self._landing:
            if wait:
                ev.synchronize()
            (done if (wait or ev.query()) else keep).append((ev, buf))
        self._landing = keep
        self.give_back([b for _, b in done])

    def add(self, c, pinned_buf=None):
        """One clip of the batch begun: device tensors pass through; a (pinned) host tensor is copied on the side stream into the slot
        (a slot that turns out too small is replaced by a larger block: the views handed out keep the old one alive).  pinned_buf: the
        staging buffer behind `c`, handed back (reap) as soon as this copy has landed."""
        if not self.on_gpu or c.is_cuda:
            if pinned_buf is not None:
                self.give_back([pinned_buf])
            return c
        st = self._cur
        slot, sz = st["slot"], -(-c.numel() // 256) * 256
        cur = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self.copy_stream):
            buf = self._slots[slot]
            if buf is None or st["at"] + sz > buf.numel():
                # sized once per batch: room for the clips still to come at this clip's size (exact for a batch of equal clips; a
                # ragged batch that outgrows it gets one more block for ITS rest, and the clips copied so far stay in the old block,
                # which the views handed out keep alive until the batch is through).  A slot is never larger than one batch.
                self._slots[slot] = None                                                  # drop the old block before asking for the new one
                buf = torch.empty(st["left"] * sz, dtype=torch.uint8, device=self.device)   # block of the copy stream's pool ...
                buf.record_stream(cur)                                                    # ... that the compute stream reads
                st["at"] = 0
                self._slots[slot] = buf
            d = buf[st["at"]:st["at"] + c.numel()].view(c.shape)
            d.copy_(c, non_blocking=True)
            if pinned_buf is not None:
                ev = torch.cuda.Event()
                ev.record(self.copy_stream)
                self._landing.append((ev, pinned_buf))
        st["at"] += sz
        st["left"] = max(st["left"] - 1, 1)
        st["any"] = True
        self.bytes_copied += c.numel()
        return d

    def end(self):
        st, self._cur = self._cur, None
        if not st["any"]:
            self._turn ^= 1                                   # nothing was copied: the slot was not used, give the turn back
            return None
        ready = torch.cuda.Event()
        ready.record(self.copy_stream)
        return (st["slot"], ready)

    def upload(self, clips):
        """clips: device tensors and / or (pinned) host tensors -> (device tensors, token): begin / add / end in one call."""
        if not self.on_gpu or all(c.is_cuda for c in clips):
            return list(clips), None
        self.begin()
        out = [self.add(c) for c in clips]
        return out, self.end()

    def wait(self, token):
        if token is not None:
            torch.cuda.current_stream(self.device).wait_event(token[1])

    def done_with(self, token):
        """Called after the batch's compute has been enqueued on the current stream."""
        if token is not None:
            self._free[token[0]].record(torch.cuda.current_stream(self.device))

    def copies_landed(self, token):
        if token is not None:
            token[1].synchronize()


def _stager(engine, world=1):
    st = getattr(engine, "_clip_stager", None)
    if st is None or st.device != engine.device:
        st = ClipStager(engine.device, world)
        try:
            engine._clip_stager = st
        except AttributeError:
            pass
    return st


def extract_dataset_clips(clips, n_clips, engine, *, clips_per_step=16, resnet=True, vit=True, full=False, flow=True,
                          out_dir=None, network_name="resnet50", skip_existing=False, mat_path=None, data_name=None,
                          rank=None, world=None, group=None, timings=None, prefetch=2, workers=4, batch_invariant=True, ramp=True):
    """clips: callable i -> uint8 [T,2,H,W,3] (device tensor, host tensor or ndarray; T and the resolution may differ from
    clip to clip), or a sequence indexed the same way.  Only this rank's shard is ever requested; the callable runs in loader
    threads (`workers` of them, up to `prefetch` batches ahead of the engine), so it may block on I/O or decode.
    -> (matrix fp32 [n_clips, F] on the engine's device, identical on every rank; errors [(clip index, message), ...]).

    prefetch = 0: no loader threads, no side stream - clips(i) is called in the driver thread as the batch is assembled (the
             round-3 behaviour; same rows bit for bit).
    ramp (default, with prefetch > 0 and clips_per_step >= 16): the pass opens with batches of B/8, 3B/8, B/2 clips (together one
             full batch) before the full ones, so that the loading of the first batch - which nothing hides - is short.
    batch_invariant (default): the pass runs with the engine's tail split-K off, so a clip's row does not depend on which clips
             share its batch - i.e. not on the number of ranks (engine.clip_vectors); the option is restored afterwards.
    out_dir: write each clip's per-frame rows [T, F] as `video_{i+1}_{network_name}_feature_map_original.npy`
             (sampling.feature_file_name; written to a temporary name and renamed, so a killed run never leaves a truncated file
             under the final name); with skip_existing a clip whose file exists is not recomputed: its row is the mean of the
             stored rows (the resume the reference lacks) - a file that cannot be read or has the wrong shape is recomputed and
             overwritten.  Not available with full=True (whole-frame and fragment features have different frame counts there).
    mat_path / data_name: rank 0 saves the matrix as {data_name: float64 [n_clips, F]} (extract_npy2mat.py:79-84).
    timings: optional dict, receives 'extract_s', 'all_gather_s' (device-synchronised wall times of the two phases),
             'loader_wait_s' (time the driver thread spent waiting for loader threads), 'h2d_bytes', 'staged_bytes' (pageable ->
             pinned copies: zero for `alloc` loaders), 'pinned_peak_bytes' / 'pinned_live_bytes' / 'pinned_pool_limit_bytes' and
             'loader_cpus' (CPUs the loader threads were bound to; 0 = not bound).
    A callable with an `alloc` parameter - clips(i, alloc=f) - gets f(shape) -> a uint8 ndarray of that shape in this rank's PINNED
    staging pool and returns the clip decoded into it: the clip is then copied host -> device straight from there."""
    if rank is None or world is None:
        import torch.distributed as dist
        on = dist.is_available() and dist.is_initialized()
        rank = dist.get_rank(group) if on else 0
        world = dist.get_world_size(group) if on else 1
    if out_dir is not None and full:
        raise ValueError("per-frame files are written for the fragment features only (full=False)")
    B = int(clips_per_step)
    if B < 1:
        raise ValueError(f"clips_per_step must be >= 1, got {clips_per_step!r}")
    get = clips if callable(clips) else clips.__getitem__
    mine = rdist.shard_clips(n_clips, rank, world)
    F = feature_dim(engine, resnet, vit, full)
    dev = engine.device
    on_gpu = dev.type == "cuda"
    local = torch.full((len(mine), F), float("nan"), dtype=torch.float32, device=dev)
    errors = []
    stager = _stager(engine, world) if prefetch > 0 else None
    # loader threads: (1) start on device 0 like every new thread: name this rank's device before they pin memory or touch tensors;
    # (2) run on the CPUs of the GPU's NUMA node (hostnode.py): decode, the pinned buffers they allocate and the PCIe root of the GPU
    # then sit on one socket
    cpus = hostnode.loader_cpus(dev.index if on_gpu else None)

    def _loader_init():
        if on_gpu:
            torch.cuda.set_device(dev)
        hostnode.bind_this_thread(cpus)

    pool = ThreadPoolExecutor(max_workers=max(int(workers), 1), thread_name_prefix="relax-loader",
                              initializer=_loader_init) if prefetch > 0 else None
    h2d_bytes0 = stager.bytes_copied if stager is not None else 0
    staged_bytes0 = stager.bytes_staged if stager is not None else 0
    # the `alloc` protocol: a loader `clips(i, alloc=f)` asks f(shape) for the uint8 array it decodes into - pinned staging memory of
    # this rank's pool - and returns it (or a view of it): no pageable copy of the clip exists and nothing is copied on the host
    takes_alloc = False
    if callable(clips) and stager is not None and not hasattr(clips, "fetch"):      # (a LoaderProcessPool brings its own staging memory)
        try:
            takes_alloc = "alloc" in inspect.signature(clips).parameters
        except (TypeError, ValueError):
            takes_alloc = False
    writes = []

    restore_split = None
    if batch_invariant and hasattr(engine, "get_option") and hasattr(engine, "set_option"):
        restore_split = engine.get_option("gemm_split_k")
        engine.set_option("gemm_split_k", 0)

    def run(batch):
        if full:
            return engine.full_clip_vectors(batch, flow=flow), None
        if out_dir is not None:
            return engine.clip_vectors(batch, resnet=resnet, vit=vit, per_frame=True)
        return engine.clip_vectors(batch, resnet=resnet, vit=vit), None

    def store(slots, idxs, vecs, frames):
        if slots == list(range(slots[0], slots[0] + len(slots))):
            local[slots[0]:slots[0] + len(slots)] = vecs          # device-to-device on the current stream: no host sync
        else:
            local[torch.as_tensor(slots, device=dev)] = vecs
        if frames is not None:
            # a per-frame file that cannot be written (disk full, permissions) costs its own clip an entry in the error list, like
            # any other per-clip failure: the row stays (it was computed), the run goes on and every rank still reaches the all-gather
            for i, rows in zip(idxs, frames):
                arr = rows.cpu().numpy()
                if pool is not None:
                    writes.append((i, pool.submit(sampling.save_clip_features, out_dir, i, network_name, arr)))
                else:
                    try:
                        sampling.save_clip_features(out_dir, i, network_name, arr)
                    except Exception as e:              # noqa: BLE001
                        errors.append((i, f"per-frame file not written: {type(e).__name__}: {e}"))

    seq_of = {i: k for k, i in enumerate(mine)}        # a clip's place in this rank's pass: the order pinned memory is granted in
    if stager is not None:
        stager.gate.start_pass()

    def load(i):
        """One clip, in a loader thread (or inline when prefetch = 0) -> ("rows", the stored per-frame rows of a finished clip) | ("clip", tensor, pinned buffer) |
        ("err", message).  Never raises.  Whatever happens, the clip's turn at the pinned budget is taken or given up (the clips
        behind it wait for that)."""
        seq, took = seq_of[i], [False]
        handed, keep, shm = [], [None], [None]          # pinned buffers this call's `alloc` gave out; the one the clip is returned in; a pool's clip

        def my_turn():
            first, took[0] = not took[0], True
            return seq if first else None

        try:
            if out_dir is not None and skip_existing:
                path = os.path.join(out_dir, sampling.feature_file_name(i, network_name))
                if os.path.exists(path):
                    try:
                        rows = np.load(path)
                        if rows.ndim == 2 and rows.shape[1] == F and rows.shape[0] > 0 and rows.dtype == np.float32:
                            return ("rows", np.ascontiguousarray(rows))
                    except Exception:                   # noqa: BLE001 - truncated / foreign file: recompute and overwrite it
                        pass
            if takes_alloc:
                def alloc(shape):
                    view, buf = stager.alloc_pinned(shape, my_turn(), own=sum(b.numel() for b in handed))
                    handed.append(buf)
                    return view.numpy()                 # (shares the pinned memory)
                clip = get(i, alloc=alloc)
            else:
                clip = get(i)
            if hasattr(clip, "release") and hasattr(clip, "tensor"):     # loaderpool.ShmClip: decoded by a loader PROCESS into shared memory the
                shm[0] = clip                                            # parent has page-locked - copied from where it lies, handed back after
                _check_clip(clip.tensor)
                if stager is None or not stager.on_gpu:                  # (a CPU 'device' uses clips where they lie, until its compute is through:
                    return ("clip", clip.tensor.clone(), None)           # the segment goes back to its worker now - `finally` below - not then)
                keep[0] = clip
                return ("clip", clip.tensor, clip)
            _check_clip(clip)
            t = clip if isinstance(clip, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(clip))
            if t.is_cuda or stager is None:
                return ("clip", t, None)
            if handed and t.is_contiguous() and any(t.data_ptr() == b.data_ptr() for b in handed):
                keep[0] = next(b for b in handed if b.data_ptr() == t.data_ptr())
                return ("clip", t, keep[0])             # decoded in place: pinned already, nothing to stage
            t, buf = stager.to_pinned(t.contiguous(), my_turn())
            return ("clip", t, buf)
        except Exception as e:                          # noqa: BLE001 - the contract: the clip fails, the run goes on
            return ("err", f"{type(e).__name__}: {e}")
        finally:
            # every exit - a loader that raised, a clip that fails _check_clip, a loader that returned a device tensor or a copy: the
            # buffers `alloc` handed out go back to the budget, except the one the returned clip lives in
            if handed:
                stager.give_back([b for b in handed if b is not keep[0]])
            if shm[0] is not None and shm[0] is not keep[0]:
                shm[0].release()
            if stager is not None and not took[0]:
                stager.gate.skip(seq)

    # ramp: the first batch has nothing to hide its loading under, so the pass opens with short batches (B/8, 3B/8, B/2) - the
    # engine starts after an eighth of a batch has been loaded and every next batch loads under the previous one's compute
    sizes = []
    if ramp and prefetch > 0 and B >= 16 and len(mine) >= 2 * B:
        sizes = [B // 8, B - B // 8 - B // 2, B // 2]       # together one full batch: the rest of the pass splits as it would without them
    starts, at = [], 0
    while at < len(mine):
        starts.append(at)
        at += sizes.pop(0) if sizes else B
    batches = [mine[lo:hi] for lo, hi in zip(starts, starts[1:] + [len(mine)])]
    futures = {}
    loader_wait = 0.0

    def submit(b):
        if pool is not None and 0 <= b < len(batches) and b not in futures:
            futures[b] = [pool.submit(load, i) for i in batches[b]]

    def stage(b):
        """Collect batch b from the loaders and start its host-to-device copies -> (slots, idxs, device clips, token, pinned)."""
        nonlocal loader_wait
        futs = futures.pop(b) if pool is not None else None
        slots, idxs, devs, pinned = [], [], [], []
        staging = stager is not None and stager.on_gpu
        if staging:
            stager.begin(len(batches[b]))
        for k, (slot, i) in enumerate(zip(range(starts[b], starts[b] + len(batches[b])), batches[b])):
            t_w = time.perf_counter()
            if futs is None:
                r = load(i)
            else:
                while True:                             # while the loader works (or waits for pinned memory): hand back what has landed
                    try:
                        r = futs[k].result(timeout=0.02 if staging else None)
                        break
                    except FutureTimeout:
                        stager.reap()
            loader_wait += time.perf_counter() - t_w
            if r[0] == "err":
                errors.append((i, r[1]))
            elif r[0] == "rows":
                # resume: the stored per-frame rows go through the SAME reduction as freshly computed ones (relax_segment_mean on
                # the device), so a resumed row is the bits of the row the first run produced (extract_npy2mat.py:121-126 is the
                # reference's host-side mean of the same files)
                local[slot] = engine.rows_mean(torch.from_numpy(r[1]).to(dev))
            else:
                slots.append(slot)
                idxs.append(i)
                if staging:
                    devs.append(stager.add(r[1], r[2]))     # the copy of this clip starts while the others still load; its pinned
                    stager.reap()                           # buffer goes back to the budget as soon as the copy has landed
                else:
                    devs.append(r[1])
                    pinned.append(r[2])
        token = stager.end() if staging else None
        return slots, idxs, devs, token, pinned

    try:
        torch.cuda.synchronize(dev) if on_gpu else None
        t0 = time.perf_counter()
        for b in range(min(prefetch + 1, len(batches))):
            submit(b)
        staged = stage(0) if batches else None
        for b in range(len(batches)):
            slots, idxs, batch, token, pinned = staged
            failed = None
            if batch:
                if stager is not None:
                    stager.wait(token)
                try:
                    vecs, frames = run(batch)
                except Exception as e0:                     # noqa: BLE001 - find the clip(s) that broke the batch, keep the others
                    failed = e0
            # batch b's compute is enqueued: get batch b+1 from the loaders and start its copies under it, refill the loader queue
            if b + 1 < len(batches):
                submit(b + 1 + prefetch)
                staged = stage(b + 1)
            if batch:
                if failed is None:
                    store(slots, idxs, vecs, frames)
                elif len(batch) == 1:
                    errors.append((idxs[0], f"{type(failed).__name__}: {failed}"))
                else:
                    for slot, i, clip in zip(slots, idxs, batch):
                        try:
                            vecs, frames = run([clip])
                            store([slot], [i], vecs, frames)
                        except Exception as e:              # noqa: BLE001
                            errors.append((i, f"{type(e).__name__}: {e}"))
                if stager is not None:
                    stager.done_with(token)
                    stager.copies_landed(token)
                    stager.reap()
                    stager.give_back(pinned)            # (a CPU 'device': the clips were used where they lay)
        for i, w in writes:
            try:
                w.result()
            except Exception as e:                          # noqa: BLE001
                errors.append((i, f"per-frame file not written: {type(e).__name__}: {e}"))
        torch.cuda.synchronize(dev) if on_gpu else None
        t1 = time.perf_counter()
        if stager is not None:
            stager.reap(wait=True)
    finally:
        # teardown, also when something escaped the driver loop (a HIP error, a stand-in engine, KeyboardInterrupt): loaders that wait
        # for pinned memory nobody will hand back must not keep pool.shutdown - and so the exception, and so the other ranks at the
        # all-gather - waiting: close the gate first (their acquire raises -> an "err" result), then collect what is still held
        if stager is not None:
            stager.gate.close()
        if pool is not None:
            pool.shutdown(wait=True, cancel_futures=True)
        if stager is not None:
            left = []
            for futs in futures.values():               # loaded but never consumed: their pinned buffers
                for f in futs:
                    if f.done() and not f.cancelled() and f.exception() is None:
                        r = f.result()
                        if r[0] == "clip" and r[2] is not None:
                            left.append(r[2])
            futures.clear()
            try:
                stager.reap(wait=True)
            except Exception:                           # noqa: BLE001 - a dead device: drop the records, the buffers are garbage-collected
                stager._landing = []
            stager.give_back(left)
        if restore_split is not None:
            engine.set_option("gemm_split_k", restore_split)
    matrix = rdist.gather_clip_vectors(local, n_clips, rank, world, group)
    all_errors = sorted(e for part in rdist.gather_objects(errors, world, group) for e in part)
    torch.cuda.synchronize(dev) if on_gpu else None
    t2 = time.perf_counter()
    if timings is not None:
        timings["extract_s"] = t1 - t0
        timings["all_gather_s"] = t2 - t1
        timings["loader_wait_s"] = loader_wait
        timings["h2d_bytes"] = stager.bytes_copied - h2d_bytes0 if stager is not None else 0
        timings["staged_bytes"] = stager.bytes_staged - staged_bytes0 if stager is not None else 0     # pageable -> pinned copies
        timings["pinned_live_bytes"] = stager.pinned_live_bytes if stager is not None else 0
        timings["pinned_peak_bytes"] = stager.pinned_peak_bytes if stager is not None else 0
        timings["pinned_in_flight_peak_bytes"] = stager.gate.peak if stager is not None else 0
        timings["pinned_pool_limit_bytes"] = stager.pool_limit_bytes if stager is not None else 0
        timings["loader_cpus"] = len(cpus) if cpus else 0
    if mat_path is not None and rank == 0:
        sampling.save_mat(mat_path, data_name or "features", matrix.cpu().numpy().astype(np.float64))
    return matrix, all_errors
