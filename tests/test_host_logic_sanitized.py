"""CPU sanitizer pass on the host half of the library (round-2 review, next #9): csrc/host_logic.cpp - state-dict key matching,
BatchNorm folding, OIHW -> packed weight rows, the quality head's fold, the tail split-K cost model - is built ALONE with
-fsanitize=address,undefined and driven by tests/host_logic_driver.py in a subprocess that preloads libasan (the sanitizer
runtime must come first in a process whose python is not instrumented).  Never on the GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "relax-vqa_amd", "csrc")


def _gcc_file(name):
    out = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def test_host_logic_under_asan_and_ubsan():
    asan, ubsan = _gcc_file("libasan.so"), _gcc_file("libubsan.so")
    if not asan:
        pytest.skip("no libasan in this toolchain")
    subprocess.run(["make", "-C", CSRC, "sanitize"], check=True, capture_output=True)
    env = dict(os.environ, LD_PRELOAD=":".join(p for p in (asan, ubsan) if p),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               PYTHONDONTWRITEBYTECODE="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "host_logic_driver.py")], env=env, capture_output=True, text=True,
                         timeout=600)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-4000:])
    assert "HOST_LOGIC_SANITIZED_OK" in res.stdout
    assert "ERROR: AddressSanitizer" not in res.stderr and "runtime error" not in res.stderr, res.stderr[-4000:]


def test_the_sanitizer_build_does_catch_an_overrun():
    """The pass above means something only if the harness can fail: hand the packer an output buffer that is one row short."""
    asan = _gcc_file("libasan.so")
    if not asan:
        pytest.skip("no libasan in this toolchain")
    subprocess.run(["make", "-C", CSRC, "sanitize"], check=True, capture_output=True)
    code = (
        "import ctypes as C, numpy as np\n"
        f"lib = C.CDLL({os.path.join(CSRC, 'librelax_host_san.so')!r})\n"
        "w = np.ones((64, 64, 1, 1), np.float32)\n"
        "ptrs = (C.c_void_p * 1)(w.ctypes.data); names = (C.c_char_p * 1)(b'c.weight'); numels = (C.c_int64 * 1)(w.size)\n"
        "out = np.zeros((63, 64), np.float32)\n"
        "err = C.create_string_buffer(64)\n"
        "lib.relax_host_pack_conv(ptrs, names, numels, 1, b'c', b'', 64, 64, 64, 1, out.ctypes.data_as(C.c_void_p), None, err, 64)\n"
        "print('survived')\n")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1")
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and "heap-buffer-overflow" in res.stderr, (res.returncode, res.stderr[-1500:])
