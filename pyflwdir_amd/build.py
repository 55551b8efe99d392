"""Build libpfd_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m pyflwdir_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpfd_hip.so")


def build(force: bool = False, verbose: bool = False) -> str:
    cmd = ["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))]
    if force:
        cmd.append("-B")
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(cmd, stdout=out)
    if not os.path.exists(LIB):
        raise RuntimeError("hipcc build finished without producing " + LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
