"""The C ABI used from C++ directly - no Python, no torch in the process: the program in
tests/c_abi/ is compiled against include/pixtrack_hip.h, linked to libpixtrack_hip.so and run."""
import os
import shutil
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_cpp_caller_recovers_the_pose(tmp_path, device):
    from pixtrack_amd import _build

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    lib = _build.build(verbose=False)
    exe = tmp_path / "lm_roundtrip"
    subprocess.check_call([hipcc, "-O2", "-I", str(ROOT / "include"), str(ROOT / "tests/c_abi/lm_roundtrip.cpp"),
                           "-L", str(lib.parent), "-lpixtrack_hip", "-o", str(exe)])
    env = dict(os.environ, LD_LIBRARY_PATH=f"{lib.parent}:{os.environ.get('LD_LIBRARY_PATH', '')}")
    out = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "C-ABI ROUNDTRIP OK" in out.stdout
