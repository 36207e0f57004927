"""CPU sanitizer pass over the native host helpers ("race detection / sanitizers": the reference has
none; this repo's native code that parses USER FILES on the host -- svx_candidate_table -- and fills caller buffers --
svx_format_alignments, svx_draw_indices, svx_mt19937_choice -- is built from its one source file with
-fsanitize=address,undefined and driven through its C ABI, on good inputs against known answers and on hostile ones
(empty files, one-token lines, CR-only line ends, megabyte-long tokens, NUL bytes, binary noise, undersized buffers).
No GPU: GPU AddressSanitizer is not available on the pool, and these functions never touch the device."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "speech-vecalign_amd", "csrc", "svx_host.hip")


def test_native_host_helpers_under_asan_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no host C++ compiler")
    asan = subprocess.run([gxx, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no AddressSanitizer runtime for g++")
    lib = str(tmp_path / "libsvx_host_san.so")
    subprocess.check_call([gxx, "-x", "c++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-fno-omit-frame-pointer", "-shared", "-fPIC", SRC, "-o", lib])
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    work = tmp_path / "work"
    work.mkdir()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "native", "host_san_driver.py"), lib, str(work)],
                         env=env, capture_output=True, text=True, timeout=600)
    tail = (out.stdout[-3000:] + "\n" + out.stderr[-3000:])
    assert out.returncode == 0, tail
    assert "HOST_SAN_OK" in out.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
