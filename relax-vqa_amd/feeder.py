"""Host -> device clip feeding overlapped with compute (SURVEY §7 step 6, §8(d) metric (ii)).

The reference reads PNGs from disk one at a time (src/main_fragment_layerstack.py:295-296).  Here decoded clips sit
in pinned host memory; a side HIP stream copies batch k+1 into the other half of a double buffer while the engine works
on batch k, and events order the two streams (copy -> compute, compute -> reuse of the buffer).  398 MB per 1080p clip
at PCIe Gen5 rates is ~8 ms against ~24 ms of compute, so the copies hide completely.
"""
import torch


class PinnedClipFeeder:
    """Iterates device-resident batches (lists of uint8 [T,2,H,W,3] tensors) from pinned host clips.

    clips_host: list of pinned uint8 CPU tensors (one per clip).  batch: clips per step.  Yields lists of device
    tensors valid until the next-but-one iteration (double buffer)."""

    def __init__(self, clips_host, batch, device):
        self.clips = clips_host
        self.batch = batch
        self.device = device
        self.copy_stream = torch.cuda.Stream(device=device)
        self.buffers = [[torch.empty_like(c, device=device) for c in clips_host[:batch]] for _ in range(2)]
        self.ready = [torch.cuda.Event() for _ in range(2)]      # copy finished
        self.free = [torch.cuda.Event() for _ in range(2)]       # compute finished with the buffer
        for e in self.free:
            e.record(torch.cuda.current_stream(device))

    def _issue(self, step):
        slot = step & 1
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(self.free[slot])
            for j in range(self.batch):
                src = self.clips[(step * self.batch + j) % len(self.clips)]
                self.buffers[slot][j].copy_(src, non_blocking=True)
            self.ready[slot].record(self.copy_stream)

    def prime(self):
        """Starts the copy of the first batch (a pipeline that keeps running pays for it once: bench.py times the steady state)."""
        self._issue(0)
        self._primed = True

    def run(self, n_steps, fn):
        """Calls fn(list_of_device_clips) n_steps times with the copies of step k+1 overlapping fn of step k."""
        out = None
        if not getattr(self, "_primed", False):
            self._issue(0)
        self._primed = False
        cur = torch.cuda.current_stream(self.device)
        for k in range(n_steps):
            if k + 1 < n_steps:
                self._issue(k + 1)
            slot = k & 1
            cur.wait_event(self.ready[slot])
            out = fn(self.buffers[slot])
            self.free[slot].record(cur)
        return out
