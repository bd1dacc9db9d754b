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


def build_capi_smoke(force: bool = False):
    """Compiles the plain-C binding tests/capi_smoke.c (gcc, include/splat_hip.h, the HIP runtime API) against the built library and the
    oracle's C build: proof that the header is C and every entry point it declares links.  Returns the binary's path, or None -- with a
    message on stderr -- when the toolchain for it is not there (no gcc, no ROCm headers, no oracle build): a missing C toolchain is
    not a failed build of the library."""
    import shutil
    root = os.path.dirname(_HERE)
    exe = os.path.join(root, "tests", "_build", "capi_smoke")
    src = os.path.join(root, "tests", "capi_smoke.c")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    oracle_so = os.path.join(root, "oracle", "_build", "libraster_ref.so")
    deps = [src, os.path.join(root, "include", "splat_hip.h"), os.path.join(_HERE, "lib", "libsplat_hip.so"), oracle_so]
    missing = [d for d in deps if not os.path.exists(d)]
    if shutil.which("gcc") is None:
        missing.append("gcc")
    if not os.path.exists(os.path.join(rocm, "include", "hip", "hip_runtime_api.h")):
        missing.append(os.path.join(rocm, "include", "hip", "hip_runtime_api.h"))
    if missing:
        sys.stderr.write(f"capi_smoke not built (skipped): missing {', '.join(missing)}\n")
        return None
    if not force and os.path.exists(exe) and all(os.path.getmtime(exe) >= os.path.getmtime(d) for d in deps):
        return exe
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Werror=implicit-function-declaration", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(root, "include"),
           "-I", os.path.join(rocm, "include"), src, "-o", exe, "-L", os.path.join(_HERE, "lib"), "-lsplat_hip",
           "-L", os.path.join(root, "oracle", "_build"), "-lraster_ref", "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-lm",
           "-Wl,-rpath,$ORIGIN/../../splatam_amd/lib", "-Wl,-rpath,$ORIGIN/../../oracle/_build", "-Wl,-rpath," + os.path.join(rocm, "lib")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building tests/capi_smoke.c failed:\n" + res.stderr[-4000:])
    return exe


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
