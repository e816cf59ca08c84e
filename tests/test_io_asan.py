"""CPU: the C++ readers against malformed PCD / PLY files, built with AddressSanitizer (make asan): every file
is rejected with `false` -- no overrun of the record decoder's buffer, no escaping exception, no allocation the
file cannot back (SIZE 64 used to smash the stack).  Runs without a GPU: nothing reaches the device."""
import os
import subprocess

from conftest import ROOT


def test_malformed_files_are_rejected_under_asan(tmp_path):
    from cupoch_amd import _lib
    _lib.build()
    cpp = os.path.join(ROOT, "cupoch_amd", "cpp")
    subprocess.check_call(["make", "-s", "-C", cpp, "asan"])
    exe = os.path.join(ROOT, "cupoch_amd", "lib", "test_io_malformed_asan")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0")
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    assert out.stdout.strip().splitlines()[-1] == "ok"
    assert "AddressSanitizer" not in out.stderr
