"""splatam_amd.dist.InStreamRccl, the parts that run without a GPU: the RCCL binding loads (the librccl.so torch ships), a unique id is
128 raw bytes that survive the trip through the struct (a c_char field would truncate at the first NUL), and the switch: the in-stream
communicator is only ever tried when SPLAT_INSTREAM_RCCL=1 AND the process group runs over the nccl backend -- over gloo (these tests,
two ranks on the CPU) the all-reduce helpers keep using torch.distributed."""
import ctypes as C
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from splatam_amd import dist as sdist


def test_rccl_binding_loads_and_unique_id_is_128_raw_bytes():
    try:
        L = sdist._load_rccl()
    except OSError:
        pytest.skip("no librccl.so on this machine")
    assert C.sizeof(sdist._NcclUniqueId) == 128
    ids = []
    for _ in range(2):
        u = sdist._NcclUniqueId()
        assert L.ncclGetUniqueId(C.byref(u)) == 0
        raw = C.string_at(C.byref(u), 128)
        assert len(raw) == 128
        back = sdist._NcclUniqueId()
        C.memmove(C.byref(back), raw, 128)
        assert bytes(back.internal) == raw                  # NUL bytes inside the id included
        ids.append(raw)
    assert ids[0] != ids[1]
    assert L.ncclGetErrorString(0).decode().lower().startswith("no error")


def test_instream_is_off_without_a_process_group_or_the_switch(monkeypatch):
    monkeypatch.setattr(sdist, "_instream", None)
    monkeypatch.setattr(sdist, "_instream_by_group", {})
    monkeypatch.delenv("SPLAT_INSTREAM_RCCL", raising=False)
    assert sdist.instream() is None
    monkeypatch.setenv("SPLAT_INSTREAM_RCCL", "1")
    assert sdist.instream() is None                         # no process group
    t = torch.arange(4.0)
    sdist.all_reduce_sum_flat(t)                            # single process: untouched
    sdist.all_reduce_mean_flat(t)
    assert torch.equal(t, torch.arange(4.0))


def _worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SPLAT_INSTREAM_RCCL="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert sdist.instream() is None                     # gloo: the switch does nothing
        t = torch.full((5,), float(rank + 1))
        sdist.all_reduce_sum_flat(t)
        assert torch.equal(t, torch.full((5,), 3.0))
        t = torch.full((5,), float(rank + 1))
        sdist.all_reduce_mean_flat(t)
        assert torch.equal(t, torch.full((5,), 1.5))
        assert sdist._instream is None and not sdist._instream_by_group       # (over gloo nothing is even cached)
    finally:
        dist.destroy_process_group()


def test_switch_is_ignored_over_gloo_two_ranks():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port), nprocs=2, join=True)
