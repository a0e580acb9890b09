"""Picklable clip sources for the loader-process tests (numpy only: a spawned worker imports this module, not torch or pytest)."""
import os

import numpy as np


def clip_array(i):
    g = np.random.default_rng(1000 + i)
    t = 1 + i % 3                                            # ragged: 1..3 pairs
    return g.integers(0, 255, (t, 2, 16 + 16 * (i % 2), 32, 3), dtype=np.uint8)


def source(i, alloc):
    """Decodes 'video i' into the pool's memory; clip 3 cannot be read, clip 6 is malformed, clip 9 is returned as a COPY (not decoded in
    place), clip 11 takes the worker down."""
    if i == 3:
        raise OSError(f"cannot decode video_{i + 1}.mp4")
    c = clip_array(i)
    if i == 6:
        c = c[:, :1]
    if i == 11 and os.environ.get("RELAX_TEST_KILL_WORKER") == "1":
        os._exit(17)
    out = alloc(c.shape)
    np.copyto(out, c)
    return out.copy() if i == 9 else out


def plain_source(i):
    return clip_array(i)
