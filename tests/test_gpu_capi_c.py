"""A binding in PLAIN C against include/splat_hip.h (tests/capi_smoke.c: gcc, the HIP runtime API, no Python, no torch): scratch sized
and wired by splat_state_layout / splat_state_bind, forward + backward on 1 000 Gaussians, every output compared with the CPU oracle
through the oracle's own C entry points.  The CPU half (it compiles and links against every symbol it uses) runs in the container;
the binary runs on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "_build", "capi_smoke")


def build_capi_smoke(force=False):
    src = os.path.join(ROOT, "tests", "capi_smoke.c")
    deps = [src, os.path.join(ROOT, "include", "splat_hip.h"), os.path.join(ROOT, "splatam_amd", "lib", "libsplat_hip.so"),
            os.path.join(ROOT, "oracle", "_build", "libraster_ref.so")]
    from oracle import c_ref
    c_ref.build()
    if not force and os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(d) for d in deps):
        return EXE
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Werror=implicit-function-declaration", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(rocm, "include"), src, "-o", EXE, "-L", os.path.join(ROOT, "splatam_amd", "lib"), "-lsplat_hip",
           "-L", os.path.join(ROOT, "oracle", "_build"), "-lraster_ref", "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-lm",
           "-Wl,-rpath,$ORIGIN/../../splatam_amd/lib", "-Wl,-rpath,$ORIGIN/../../oracle/_build", "-Wl,-rpath," + os.path.join(rocm, "lib")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-4000:]
    return EXE


def test_plain_c_binding_compiles_against_the_header():
    """The header is C (not C++), and every entry point the C binding uses is exported by the library."""
    exe = build_capi_smoke(force=True)
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_plain_c_binding_matches_the_oracle():
    exe = build_capi_smoke()
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(res.stdout)
    assert res.returncode == 0 and "capi_smoke ok" in res.stdout, res.stdout[-3000:] + res.stderr[-3000:]
