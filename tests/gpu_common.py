"""Shared helpers for the -m gpu parity tests (HIP path vs oracle through the C-ABI)."""
import numpy as np
import torch

import relax_vqa_amd  # noqa: F401
from relax_vqa_amd import synth
from relax_vqa_amd.engine import RelaxEngine

# north_star tolerance: features within 1e-3 relative fp32.  The engine's contractions run in one of its fp32-grade arithmetics
# (`each_precision` in conftest.py runs a test under each: split operands on the 16-bit matrix cores with fp32 accumulation - the
# default -, or the exact fp32 MFMA); the element-wise check below uses rtol 1e-3 with an absolute floor of 1e-4 x the block's mean
# magnitude (features that are exactly ~0 after ReLU have no meaningful relative error).
RTOL = 1e-3
ATOL_FRAC = 1e-4

_engine = None
_weights = {}


def engine():
    global _engine
    if _engine is None:
        _engine = RelaxEngine(0)
    return _engine


def rn50_weights(adversarial=False):
    """The synthetic ResNet-50 weights (or the adversarial second set), loaded into the shared engine."""
    key = "rn" + (f":{'adv' if adversarial is True else adversarial}" if adversarial else "")   # (False | True | "outliers")
    if key not in _weights:
        _weights[key] = synth.resnet50_state_dict(adversarial=adversarial)
    if _weights.get("rn_loaded") != key:
        engine().load_resnet50(_weights[key])
        _weights["rn_loaded"] = key
    return _weights[key]


def vit_weights(name, adversarial=False):
    key = "vit:" + name + (f":{'adv' if adversarial is True else adversarial}" if adversarial else "")
    if key not in _weights:
        _weights[key] = synth.vit_state_dict(name, adversarial=adversarial)
    if _weights.get("vit_loaded") != key:
        engine().load_vit(_weights[key], name)
        _weights["vit_loaded"] = key
    return _weights[key]


WEIGHT_SETS = [False, True, "outliers"]          # regular, adversarial, real-checkpoint pathology (synth.py)
WEIGHT_SET_IDS = ["regular", "adversarial", "outliers"]


def golden_tag(adversarial):
    return {False: "", True: "_adv", "outliers": "_out"}[adversarial]


def assert_close(got, want, what, rtol=RTOL, atol_frac=ATOL_FRAC, channel_axis=None):
    """channel_axis: the absolute floor of a channel is atol_frac x max(the tensor's mean magnitude, THAT CHANNEL's mean magnitude).  A
    channel that runs a hundred times above the others (the outlier weight set) carries a hundred times their absolute rounding error,
    in EVERY fp32 arithmetic - torch's own fp32 forward is 5e-4 away from an fp64 run there, where the floor of the tensor mean is 1e-4 -
    and where such a channel passes through zero a relative bound has nothing to hold on to; the floor never tightens below the tensor's."""
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    assert np.isfinite(got).all(), f"{what}: non-finite values"
    scale = float(np.abs(want).mean()) + 1e-30
    err = np.abs(got - want)
    floor = atol_frac * scale
    if channel_axis is not None:
        axes = tuple(a for a in range(want.ndim) if a != channel_axis)
        floor = np.maximum(floor, atol_frac * np.abs(want).mean(axis=axes, keepdims=True))
    bound = rtol * np.abs(want) + floor
    worst = float((err / bound).max())
    assert worst <= 1.0, (f"{what}: max err/bound {worst:.3g}; max abs err {err.max():.3e}, "
                          f"mean |want| {scale:.3e}, norm-rel {np.linalg.norm(got - want) / np.linalg.norm(want):.3e}")
    return float(np.linalg.norm(got - want) / (np.linalg.norm(want) + 1e-30))


def run_ranks(make_cmd, cwd, env, timeout=600, attempts=2):
    """Runs a multi-rank launch line (make_cmd(port) -> argv) as a child process.  The rendezvous port is probed free a moment
    before the launcher binds it, so another process can take it in between.  A launch is repeated (once, on a fresh port) ONLY
    when it provably died before the program started: the launcher itself could not bind its port (address in use) AND nothing
    of the program ran - no stdout, no Python traceback of the program in stderr.  Peer ranks print 'Connection refused' /
    'DistNetworkError' too when another rank crashes mid-run and the store goes away, so those strings alone never trigger a
    repeat.  The first attempt is never discarded silently: a repeat raises a warning with its stderr tail and the returned
    result carries it as `first_attempt`."""
    import socket
    import subprocess
    import warnings
    res, first = None, None
    for k in range(attempts):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        res = subprocess.run(make_cmd(port), cwd=cwd, env=env, capture_output=True, text=True, timeout=timeout)
        if res.returncode == 0:
            break
        bind_failed = any(k_ in res.stderr for k_ in ("EADDRINUSE", "Address already in use", "address already in use"))
        # the program's own frames in a traceback, or anything on stdout, mean it had started (the launcher's own traceback on a
        # bind failure names torch/distributed files only)
        program_ran = bool(res.stdout.strip()) or 'bench.py", line' in res.stderr
        if not bind_failed or program_ran or k + 1 == attempts:
            break
        first = res
        warnings.warn(f"run_ranks: the launcher could not bind port {port}; repeating once on a fresh port.  stderr tail of the "
                      f"first attempt:\n{res.stderr[-1500:]}")
    res.first_attempt = first
    return res
