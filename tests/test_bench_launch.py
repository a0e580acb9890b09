"""bench.py's launcher: `python bench.py --gpus N` run plainly starts N ranks itself; a WORLD_SIZE that
disagrees with --gpus is refused.  CPU only: --launch-check rehearses rendezvous + all-gather + timing reduction over gloo."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=600)


def test_gpus_2_without_a_launcher_spawns_two_ranks():
    p = _run(["--gpus", "2", "--launch-check"], {"RELAX_DIST_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout          # ONE JSON line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks"] == 2 and rec["backend"] == "gloo" and rec["rccl_ranks"] == 0   # gloo is not RCCL
    assert "launching 2 ranks" in p.stderr


def test_world_size_mismatch_is_an_error_not_a_warning():
    p = _run(["--gpus", "8", "--launch-check"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert p.returncode != 0 and "refusing" in p.stderr and not p.stdout.strip()


def test_single_rank_needs_no_launcher():
    p = _run(["--gpus", "1", "--launch-check"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 1
    assert "launching" not in p.stderr
