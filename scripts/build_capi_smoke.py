#!/usr/bin/env python3
"""Build recipe of the plain-C binding test tests/capi_smoke.c (test infrastructure: it links the oracle's C build as its checker, so it
lives outside the product package).  Called by __graft_entry__.build() and tests/test_gpu_capi_c.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_capi_smoke(force: bool = False):
    """Compiles the plain-C binding tests/capi_smoke.c (gcc, include/splat_hip.h, the HIP runtime API) against the built library and the
    oracle's C build: proof that the header is C and every entry point it declares links.  Returns the binary's path, or None -- with a
    message on stderr -- when the toolchain for it is not there (no gcc, no ROCm headers, no oracle build): a missing C toolchain is
    not a failed build of the library."""
    import shutil
    root = ROOT
    exe = os.path.join(root, "tests", "_build", "capi_smoke")
    src = os.path.join(root, "tests", "capi_smoke.c")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    oracle_so = os.path.join(root, "oracle", "_build", "libraster_ref.so")
    deps = [src, os.path.join(root, "include", "splat_hip.h"), os.path.join(ROOT, "splatam_amd", "lib", "libsplat_hip.so"), oracle_so]
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
           "-I", os.path.join(rocm, "include"), src, "-o", exe, "-L", os.path.join(ROOT, "splatam_amd", "lib"), "-lsplat_hip",
           "-L", os.path.join(root, "oracle", "_build"), "-lraster_ref", "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-lm",
           "-Wl,-rpath,$ORIGIN/../../splatam_amd/lib", "-Wl,-rpath,$ORIGIN/../../oracle/_build", "-Wl,-rpath," + os.path.join(rocm, "lib")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building tests/capi_smoke.c failed:\n" + res.stderr[-4000:])
    return exe



if __name__ == "__main__":
    print(build_capi_smoke(force="--force" in sys.argv))
