"""Builds libpixtrack_hip.so (the C-ABI library, include/pixtrack_hip.h) in-tree with
hipcc for gfx950.  No torch headers are involved: the library is plain HIP + extern "C".

    python -m pixtrack_amd._build            # incremental
    python -m pixtrack_amd._build --force
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
LIB = ROOT / "libpixtrack_hip.so"
OBJ = ROOT / "csrc" / "_obj"

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-fno-fast-math",     # parity with the fp32 CPU oracle: no reassociation / approx div
    "-ffp-contract=on",
    "-Wno-comment",
    "-Wno-unused-result",
]


# per-file extra flags: the NeRF march must round exactly like the numpy oracle; MFMA results of the
# NeRF MLPs go straight to VGPRs (hipcc's default puts them in AGPRs and copied 214 registers back per
# 64 samples with v_accvgpr_read: a quarter of the shade kernel's VALU instructions)
EXTRA = {"pxt_ngp": ["-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def sources():
    return sorted(CSRC.glob("*.hip"))


def _stale(out: Path, deps) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    OBJ.mkdir(exist_ok=True)
    headers = list(CSRC.glob("*.h")) + [ROOT.parent / "include" / "pixtrack_hip.h"]
    objs = []
    for src in sources():
        obj = OBJ / (src.stem + ".o")
        objs.append(obj)
        if force or _stale(obj, [src, *headers]):
            cmd = [HIPCC, *FLAGS, *EXTRA.get(src.stem, []), "-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(LIB)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
