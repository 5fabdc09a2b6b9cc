"""Run-time variants of the hot path must be bit-identical: one / two / three / four pipes per NeRF render (slices of the ray
list, all in the same three launches), other workgroup counts of the persistent render kernel and of the ray generation
(a ray's result depends neither on the grid nor on which rays share its wave), the two UNet passes on two streams vs one
batched pass, the bit-plane vs byte-plane mask kernel, the two coarse UNet heads in one launch vs two, the first UNet
layer fused into the second layer's staging vs its own launch.  Each variant is a knob read once per process, so every
run is a subprocess of scripts/variant_checksum.py; the digests of its outputs are compared."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _digests(env_extra):
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, str(ROOT / "scripts" / "variant_checksum.py"), "320", "240"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = dict(line.split()[1:3] for line in out.stdout.splitlines() if line.startswith("DIGEST"))
    assert len(d) == 9, out.stdout
    return d


def test_runtime_variants_are_bit_identical():
    base = _digests({})
    for knobs in ({"PXT_NGP_PIPES": "2"}, {"PXT_NGP_PIPES": "3"}, {"PXT_UNET_STREAMS": "1"},
                  {"PXT_MASK_BYTES": "1"}, {"PXT_UNET_FUSE_FIRST": "0"}, {"PXT_UNET_MERGE_HEADS": "0"},
                  {"PXT_NGP_GRID": "96", "PXT_NGP_GRID_RAYGEN": "64"},
                  {"PXT_NGP_GRID": "1024", "PXT_NGP_GRID_DIV": "1"}, {"PXT_NGP_COOP": "0"},
                  {"PXT_NGP_GRID": "16384", "PXT_NGP_GRID_DIV": "4096", "PXT_NGP_PIPES": "4", "PXT_NGP_GRID_RAYGEN": "8192"}):
        assert _digests(knobs) == base, knobs
