"""A binding in PLAIN C against include/splat_hip.h (tests/capi_smoke.c: gcc, the HIP runtime API, no Python, no torch): scratch sized
and wired by splat_state_layout / splat_state_bind, forward + backward on 1 000 Gaussians, every output compared with the CPU oracle
through the oracle's own C entry points.  The CPU half (it compiles and links against every symbol it uses) runs in the container;
the binary runs on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_capi_smoke(force=False):
    """(the recipe: scripts/build_capi_smoke.py -- __graft_entry__.build() runs it without importing tests)"""
    from oracle import c_ref
    from scripts.build_capi_smoke import build_capi_smoke as build
    c_ref.build()
    exe = build(force)
    if exe is None:
        pytest.skip("no C toolchain for tests/capi_smoke.c on this machine")
    return exe


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
