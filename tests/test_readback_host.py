"""CPU: the read-back helpers of csrc/rato_common.h (arm a pinned record, watch its words arrive) as a host-only program:
the waiter returns exactly when every word is there, whatever the values (NaN, zero, a neighbouring NaN payload)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_readback_wait_returns_when_every_word_has_arrived(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "readback_test")
    cmd = [hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "riskaversetrajopt_amd", "csrc"),
           os.path.join(ROOT, "tests", "host", "readback_test.hip"), "-o", exe, "-lpthread"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    env = dict(os.environ, RATO_CUT_POLL="1", LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=env)
    assert run.returncode == 0 and "readback ok" in run.stdout, (run.returncode, run.stdout, run.stderr[-2000:])
