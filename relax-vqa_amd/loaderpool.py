"""Loader PROCESSES for the dataset pass: the decode of a from-files run (sampling.load_clip_from_frames: Pillow PNG decode, np copies)
leaves the driver process.  P worker processes write every clip straight into shared memory which the parent has page-locked
(hipHostRegister), so the clip still goes host -> device from where it was decoded, without a copy on the host.  (Measured,
profiles/r06_from_files_config4.json: a 960 x 540 PNG costs 13.7 ms of a core; on the 16-CPU quota of a GPU box 16 processes reach
1137 decodes/s against 964 of 16 loader threads - the wall there is CPU time, which round 5 had read as the GIL; a host with the
cores - one MI355X consumes 7300 decodes/s at config 4 - needs processes to use them.)

    worker p:  clip = source(i, alloc)      alloc(shape) -> a numpy view of one of the worker's shared-memory segments
               -> ("ok", segment name, shape)                 (or ("err", message): the clip fails, the pass goes on)
    parent:    attaches the segment by name (once: cached), registers it with the GPU runtime (once), wraps it in a tensor;
               dataset.ClipStager copies it on the side stream like any pinned clip and hands the segment back to its worker
               (`ShmClip.release`) when the copy has landed

The reference's counterpart is the frame reading of its per-video loop (src/main_fragment_layerstack.py:283-296: cv2.imread of every
sampled frame, serially, in the driver process).

Rules of this pool (and of the GPU boxes):
  * worker processes are SPAWNED (a fresh interpreter each), and `start()` must run BEFORE the parent process touches a GPU: a process
    that has initialised the GPU must not fork + exec another program.  bench.py starts the pool first thing; `start()` refuses when
    torch reports an initialised CUDA context.
  * `source` must be picklable (a module-level function or a functools.partial of one); it runs in the workers and never sees a GPU.
  * a worker keeps at most `segments_per_worker` clips in flight; with none free it waits for the parent to hand one back
    (back-pressure: shared memory is bounded by workers x segments x the largest clip).
  * a worker that dies fails the clips it held (an error entry each); the others go on.  `close()` unlinks every segment.
"""
import inspect
import itertools
import multiprocessing as mp
import os
import queue
import threading
from multiprocessing import shared_memory

import numpy as np

_PREFIX = "relaxldr"


class LoaderError(RuntimeError):
    pass


class ShmClip:
    """A decoded clip in a worker's shared-memory segment: `.tensor` (uint8 [T,2,H,W,3], page-locked when a GPU is in use) and `.release()`
    (hands the segment back to the worker; idempotent).  dataset.extract_dataset_clips accepts it wherever a clip is expected."""

    def __init__(self, tensor, release):
        self.tensor = tensor
        self._release = release
        self._done = False

    def numel(self):
        return self.tensor.numel()

    def release(self):
        if not self._done:
            self._done = True
            self._release()


def _worker_main(source, wid, task_q, result_q, free_q, max_segments, tag):
    takes_alloc = False
    try:
        takes_alloc = "alloc" in inspect.signature(source).parameters
    except (TypeError, ValueError):
        pass
    segs, free = {}, []                     # name -> SharedMemory; names free for reuse
    counter = itertools.count()

    def take(nbytes):
        """A segment of at least nbytes that is not in flight: a free one, a new one while fewer than max_segments exist, a free one that
        is too small replaced by a larger one (the parent is told to drop its mapping), else wait for the parent to hand one back."""
        while True:
            try:
                while True:
                    free.append(free_q.get_nowait())
            except queue.Empty:
                pass
            fit = [n for n in free if segs[n].size >= nbytes]
            if fit:
                name = min(fit, key=lambda n: segs[n].size)
                free.remove(name)
                return segs[name]
            if len(segs) >= max_segments and free:
                name = min(free, key=lambda n: segs[n].size)
                free.remove(name)
                old = segs.pop(name)
                result_q.put((-1, -1, "drop", name, None, wid))
                old.close()
                old.unlink()
            if len(segs) < max_segments:
                size = -(-max(int(nbytes), 1) // (1 << 20)) * (1 << 20)        # whole MiB: clips of nearly equal size share segments
                seg = shared_memory.SharedMemory(create=True, size=size, name=f"{_PREFIX}_{tag}_{wid}_{next(counter)}")
                segs[seg.name] = seg
                return seg
            free.append(free_q.get())       # every segment is in flight: wait for the parent (back-pressure)

    try:
        while True:
            task = task_q.get()
            if task is None:
                break
            seq, i = task
            held = []
            try:
                def alloc(shape):
                    n = int(np.prod(shape))
                    seg = take(n)
                    held.append(seg)
                    return np.ndarray(tuple(int(d) for d in shape), dtype=np.uint8, buffer=seg.buf)

                arr = source(i, alloc=alloc) if takes_alloc else source(i)
                arr = np.asarray(arr)
                if arr.dtype != np.uint8:
                    raise TypeError(f"clip must be uint8, got {arr.dtype}")
                keep = None
                for seg in held:            # decoded in place?
                    base = np.ndarray((seg.size,), dtype=np.uint8, buffer=seg.buf)
                    if arr.size and np.shares_memory(arr, base) and arr.flags["C_CONTIGUOUS"] and \
                            arr.__array_interface__["data"][0] == base.__array_interface__["data"][0]:
                        keep = seg
                        break
                if keep is None:            # a copy / a loader without `alloc`: what it was handed goes back first, then one copy into a segment
                    for seg in held:
                        free.append(seg.name)
                    held.clear()
                    keep = take(arr.size)
                    np.ndarray(arr.shape, dtype=np.uint8, buffer=keep.buf)[...] = arr
                for seg in held:
                    if seg is not keep:
                        free.append(seg.name)
                held.clear()
                result_q.put((seq, i, "ok", keep.name, tuple(arr.shape), wid))
                del arr
            except BaseException as e:      # noqa: BLE001 - the clip fails, the worker goes on
                for seg in held:
                    free.append(seg.name)
                result_q.put((seq, i, "err", f"{type(e).__name__}: {e}", None, wid))
                if not isinstance(e, Exception):
                    raise
    finally:
        for seg in segs.values():
            try:
                seg.close()
                seg.unlink()
            except Exception:               # noqa: BLE001
                pass


class LoaderProcessPool:
    """pool = LoaderProcessPool(source, processes).start()   # before any GPU call of this process
    matrix, errors = dataset.extract_dataset_clips(pool, n, engine, workers=2 * processes, ...)
    pool.close()"""

    def __init__(self, source, processes=8, segments_per_worker=3):
        self.source = source
        self.processes = max(int(processes), 1)
        self.segments_per_worker = max(int(segments_per_worker), 1)
        self._ctx = mp.get_context("spawn")
        self._procs, self._task_qs, self._free_qs = [], [], []
        self._result_q = None
        self._futures = {}                  # seq -> [event, result, worker]
        self._lock = threading.Lock()
        self._seq = itertools.count()
        self._rr = itertools.count()
        self._attached = {}                 # segment name -> (SharedMemory, torch tensor over all of it, registered?)
        self._closed = False
        self._tag = f"{os.getpid()}_{os.urandom(2).hex()}"
        self.bytes_decoded = 0

    # the dataset driver treats the pool as its `clips` callable
    def __call__(self, i):
        return self.fetch(i)

    def start(self):
        try:
            import torch
            if torch.cuda.is_available() and torch.cuda.is_initialized():
                raise RuntimeError("LoaderProcessPool.start() must run before this process touches a GPU (a process that has "
                                   "initialised the GPU must not start other programs on these boxes): start the pool first")
        except ImportError:
            pass
        self._result_q = self._ctx.Queue()
        for w in range(self.processes):
            tq, fq = self._ctx.Queue(), self._ctx.Queue()
            p = self._ctx.Process(target=_worker_main, name=f"relax-loader-{w}", daemon=True,
                                  args=(self.source, w, tq, self._result_q, fq, self.segments_per_worker, self._tag))
            p.start()
            self._procs.append(p)
            self._task_qs.append(tq)
            self._free_qs.append(fq)
        self._dispatcher = threading.Thread(target=self._dispatch, name="relax-loader-dispatch", daemon=True)
        self._dispatcher.start()
        return self

    def _dispatch(self):
        while not self._closed:
            try:
                seq, i, status, payload, shape, wid = self._result_q.get(timeout=0.25)
            except queue.Empty:
                self._fail_dead_workers()
                continue
            except (EOFError, OSError):
                return
            if status == "drop":            # a worker replaced a segment: forget the mapping
                self._drop(payload)
                continue
            with self._lock:
                fut = self._futures.get(seq)
            if fut is not None:
                fut[1] = (status, payload, shape, wid)
                fut[0].set()

    def _drop(self, name):
        with self._lock:
            att = self._attached.pop(name, None)
        if att is None:
            return
        seg, flat, registered = att
        try:
            if registered:
                import torch
                torch.cuda.cudart().cudaHostUnregister(flat.data_ptr())
        except Exception:                   # noqa: BLE001
            pass
        del flat, att
        try:
            seg.close()
        except Exception:                   # noqa: BLE001
            pass

    def _fail_dead_workers(self):
        for w, p in enumerate(self._procs):
            if p is not None and not p.is_alive() and not self._closed:
                with self._lock:
                    pending = [f for f in self._futures.values() if f[2] == w and not f[0].is_set()]
                for f in pending:
                    f[1] = ("err", f"LoaderError: loader process {w} died (exit code {p.exitcode})", None, w)
                    f[0].set()

    def _alive(self):
        return [w for w, p in enumerate(self._procs) if p.is_alive()]

    def fetch(self, i):
        """Blocks until clip i has been decoded by a worker -> ShmClip; raises LoaderError with the worker's message if it failed."""
        import torch
        if self._closed or not self._procs:
            raise LoaderError("the loader pool is not running")
        alive = self._alive()
        if not alive:
            raise LoaderError("every loader process has died")
        w = alive[next(self._rr) % len(alive)]
        seq = next(self._seq)
        fut = [threading.Event(), None, w]
        with self._lock:
            self._futures[seq] = fut
        self._task_qs[w].put((seq, int(i)))
        fut[0].wait()
        with self._lock:
            self._futures.pop(seq, None)
        status, payload, shape, wid = fut[1]
        if status != "ok":
            raise LoaderError(payload)
        name = payload
        with self._lock:
            att = self._attached.get(name)
            if att is None:
                seg = shared_memory.SharedMemory(name=name)
                # (Python 3.10 registers attached segments with the resource tracker too; parent and spawned workers share ONE tracker
                # whose cache is a set, so the owner's unlink clears the entry)
                flat = torch.frombuffer(seg.buf, dtype=torch.uint8)
                registered = False
                if torch.cuda.is_available() and torch.cuda.is_initialized():
                    rc = torch.cuda.cudart().cudaHostRegister(flat.data_ptr(), flat.numel(), 0)
                    registered = int(rc) == 0
                att = self._attached[name] = (seg, flat, registered)
        n = int(np.prod(shape))
        self.bytes_decoded += n
        t = att[1][:n].view(tuple(shape))
        fq = self._free_qs[wid]
        return ShmClip(t, lambda: fq.put(name))

    def close(self):
        if self._closed:
            return
        self._closed = True
        for tq in self._task_qs:
            try:
                tq.put(None)
            except Exception:               # noqa: BLE001
                pass
        for p in self._procs:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
                p.join(timeout=2)
        import torch
        for name in list(self._attached):
            seg = self._attached[name][0]
            self._drop(name)
            try:
                seg.unlink()                # (a terminated worker could not unlink its own)
            except Exception:               # noqa: BLE001
                pass
        for q_ in self._task_qs + self._free_qs + ([self._result_q] if self._result_q is not None else []):
            try:
                q_.close()
            except Exception:               # noqa: BLE001
                pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def frames_source(i, alloc, sampled_frame_path=None, names=None):
    """The module-level (picklable) source of a from-files pass: clip i = the sampled frames of video names[i % len(names)] under
    sampled_frame_path (use functools.partial to bind the two)."""
    from . import sampling
    return sampling.load_clip_from_frames(sampled_frame_path, names[i % len(names)], alloc=alloc)
