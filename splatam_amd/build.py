"""Builds libsplat_hip.so for gfx950 in-tree (splatam_amd/lib/).

hipcc cross-compiles without a GPU, so this runs in the CPU-only container too.
"""
from __future__ import annotations

import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(force: bool = False, verbose: bool = False) -> str:
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc, "-j4"] + (["-B"] if force else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        sys.stderr.write(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libsplat_hip.so failed")
    out = os.path.join(_HERE, "lib", "libsplat_hip.so")
    assert os.path.exists(out), out
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
