"""the host-side native code (pose2mesh_release_amd/csrc/p2m_host.cpp: HEM matching and the
tree ordering, the C++ replacements of lib/coarsening.py:153-258) under AddressSanitizer + UndefinedBehaviorSanitizer,
on a well-formed graph and on malformed inputs (tests/host_sanitize_driver.cpp)."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_host_library_is_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "host_sanitize")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", "-o", exe, os.path.join(HERE, "host_sanitize_driver.cpp"),
           os.path.join(ROOT, "pose2mesh_release_amd", "csrc", "p2m_host.cpp")]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "host sanitizer driver ok" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr
