"""Build libnsdecode.so (HIP, gfx950) in-tree with hipcc.

-ffp-contract=off is part of the numerical contract: the decoder's fp32
log-softmax and fp64 score arithmetic must round operation by operation so
that results are reproducible on the host.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
SO = PKG / "libnsdecode.so"
SOURCES = ["nsd_step.hip", "nsd_aux.hip", "nsd_prior.hip", "nsd_attn.hip"]
HEADERS = [CSRC / "nsd_device.h", ROOT / "include" / "nsdecode.h"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found: libnsdecode.so cannot be built")
    return exe


def stale() -> bool:
    if not SO.exists():
        return True
    t = SO.stat().st_mtime
    return any(p.stat().st_mtime > t for p in [CSRC / s for s in SOURCES] + HEADERS)


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not stale():
        return SO
    cmd = [hipcc(), *FLAGS, f"-I{ROOT / 'include'}", f"-I{CSRC}", *[str(CSRC / s) for s in SOURCES], "-o", str(SO)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=str(CSRC))
    return SO


if __name__ == "__main__":
    build(force=True, verbose=True)
    print(SO, os.path.getsize(SO), "bytes")
