"""View-sharded mapping across the GPUs of one node.

The reference is single-GPU.  Mapping optimises the SAME replicated Gaussian map
against independent keyframe views, so the views shard across ranks with one
exchange step per iteration: a sum all-reduce of the Gaussian gradients
(one flat bucket: 12 floats per isotropic Gaussian on the drop-in path, 8 on the fused path, which knows that the
rotation gradient of an isotropic map is exactly zero), followed by the identical
Adam step on every rank.  One process per GPU; backend "nccl" is RCCL over xGMI
on ROCm, "gloo" is used by the CPU tests.  Tracking shards over tile rows with one 16 KB
all-reduce of its partial sums per iteration (FusedEngine.tracking_iteration(shard=...)).
The two per-iteration all-reduces can be issued on the iteration's own stream (InStreamRccl, opt-in).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Iterable, Optional

import torch
import torch.distributed as dist

GAUSSIAN_KEYS = ('means3D', 'rgb_colors', 'unnorm_rotations', 'logit_opacities', 'log_scales')


def init_from_env(backend: Optional[str] = None):
    """Initialise torch.distributed from torchrun's environment; returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # SPLAT_DIST_BACKEND=gloo lets several ranks share one GPU (development boxes); RCCL needs one GPU per rank
            backend = os.environ.get("SPLAT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


class GradBucket:
    """One flat buffer holding the gradients of the replicated Gaussian parameters,
    so that the exchange is a single large collective (xGMI rings are per-link
    bound: fewer, larger messages)."""

    def __init__(self, params: Dict[str, torch.Tensor], keys: Iterable[str] = GAUSSIAN_KEYS):
        self.keys = [k for k in keys if k in params]
        self.sizes = [params[k].numel() for k in self.keys]
        dev, dt = params[self.keys[0]].device, params[self.keys[0]].dtype
        self.flat = torch.zeros(sum(self.sizes), device=dev, dtype=dt)
        self.views = []
        o = 0
        for k, n in zip(self.keys, self.sizes):
            self.views.append(self.flat[o:o + n].view_as(params[k]))
            o += n

    def all_reduce_mean(self, params: Dict[str, torch.Tensor], group=None) -> None:
        """grad <- mean over ranks of grad, for every replicated Gaussian parameter."""
        if not dist.is_initialized() or dist.get_world_size(group) == 1:
            return
        world = dist.get_world_size(group)
        with torch.no_grad():
            for k, v in zip(self.keys, self.views):
                g = params[k].grad
                if g is None:
                    v.zero_()
                else:
                    v.copy_(g)
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(world)
            for k, v in zip(self.keys, self.views):
                if params[k].grad is None:
                    params[k].grad = v.clone()
                else:
                    params[k].grad.copy_(v)


class _NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_ubyte * 128)]           # NCCL_UNIQUE_ID_BYTES (rccl.h); (c_char would truncate at the first NUL)


def _load_rccl():
    """The RCCL library torch itself has loaded (torch/lib/librccl.so: the same instance), else ROCm's."""
    cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so", "librccl.so"]
    last = None
    for path in cands:
        try:
            L = C.CDLL(path)
            break
        except OSError as e:
            last = e
    else:
        raise OSError(f"librccl.so not found ({last})")
    L.ncclGetUniqueId.argtypes = [C.POINTER(_NcclUniqueId)]
    L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _NcclUniqueId, C.c_int]
    L.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.ncclCommDestroy.argtypes = [C.c_void_p]
    L.ncclGetErrorString.argtypes = [C.c_int]
    L.ncclGetErrorString.restype = C.c_char_p
    for f in (L.ncclGetUniqueId, L.ncclCommInitRank, L.ncclAllReduce, L.ncclCommDestroy):
        f.restype = C.c_int
    return L


class InStreamRccl:
    """RCCL collectives issued ON the caller's HIP stream (ncclAllReduce through ctypes on the librccl.so torch has loaded, with
    ``torch.cuda.current_stream()``), on a communicator of its own.  ``torch.distributed``'s NCCL backend runs every collective on
    the process group's internal stream: each call is two event hand-offs (caller's stream -> comm stream -> caller's stream) around a
    message that, for the two per-iteration exchanges of this path (the 16 KB of tracking partial sums; the gradient bucket), takes
    less time than the hand-offs.  In stream, the all-reduce is one more kernel in the iteration's queue: K7 -> F6 -> all-reduce ->
    Adam, no host or event synchronisation at all.  (``torch.cuda.nccl.init_rank`` would be the torch-native way to get such a
    communicator; under Python 3.10 it raises "PY_SSIZE_T_CLEAN macro must be defined" in torch 2.10.)

    Opt-in (``SPLAT_INSTREAM_RCCL=1``, ``bench.py --instream-rccl``): it has run on hardware with ONE rank only (the development
    boxes have one GPU; tests/test_gpu_dist_pipeline.py::test_instream_rccl_single_rank) -- the default stays torch.distributed."""

    SUM, PROD, MAX, MIN, AVG = 0, 1, 2, 3, 4        # ncclRedOp_t
    _DTYPES = {torch.int32: 2, torch.int64: 4, torch.float32: 7, torch.float64: 8}       # ncclDataType_t

    def __init__(self, rank: int, world: int, uid: Optional[bytes] = None, group=None):
        self.L, self.rank, self.world = _load_rccl(), rank, world
        if uid is None:
            # rank 0 draws the id; it travels over the existing process group (any backend)
            box = [None]
            if rank == 0:
                u = _NcclUniqueId()
                self._check(self.L.ncclGetUniqueId(C.byref(u)), "ncclGetUniqueId")
                box[0] = C.string_at(C.byref(u), 128)
            if world > 1:
                dist.broadcast_object_list(box, src=0, group=group)
            uid = box[0]
        u = _NcclUniqueId()
        if len(uid) != 128:
            raise ValueError("ncclUniqueId is 128 bytes")
        C.memmove(C.byref(u), uid, 128)
        comm = C.c_void_p()
        self._check(self.L.ncclCommInitRank(C.byref(comm), world, u, rank), "ncclCommInitRank")
        self.comm = comm

    def _check(self, rc: int, what: str) -> None:
        if rc != 0:
            raise RuntimeError(f"{what} failed: {self.L.ncclGetErrorString(rc).decode()} ({rc})")

    def all_reduce(self, t: torch.Tensor, op: int = 0) -> None:
        """In place, on ``torch.cuda.current_stream(t.device)``."""
        if not t.is_contiguous():
            raise ValueError("in-stream all-reduce needs a contiguous tensor")
        if t.dtype not in self._DTYPES:
            raise ValueError(f"in-stream all-reduce: unsupported dtype {t.dtype}")
        with torch.cuda.device(t.device):
            stream = torch.cuda.current_stream(t.device).cuda_stream
            self._check(self.L.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), self._DTYPES[t.dtype], op, self.comm, stream), "ncclAllReduce")

    def close(self) -> None:
        if self.comm:
            self.L.ncclCommDestroy(self.comm)
            self.comm = None


_instream: Optional[InStreamRccl] = None         # the communicator of the DEFAULT group (bench.py reports it)
_instream_by_group: Dict[object, Optional[InStreamRccl]] = {}


def instream(group=None) -> Optional[InStreamRccl]:
    """The in-stream communicator of ``group`` when it is switched on (SPLAT_INSTREAM_RCCL=1) and the job runs over RCCL, else None.
    Every step that could differ between the ranks -- the environment switch, loading librccl, drawing the id, creating the
    communicator, the self-check -- is followed by ONE all-reduce(MIN) of its outcome over the process group before anything
    depends on it, so that all ranks issue the same sequence of collectives and take the same path (a rank whose library fails to
    load cannot leave the others waiting in a broadcast)."""
    global _instream
    key = group
    if key in _instream_by_group:
        return _instream_by_group[key]
    if not dist.is_initialized() or dist.get_backend(group) != "nccl":
        return None
    dev = torch.device("cuda", torch.cuda.current_device())

    def all_agree(ok: bool) -> bool:
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return int(t[0]) == 1

    comm, err = None, None
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    # 1. switched on everywhere, and the library loads everywhere
    L = None
    want = os.environ.get("SPLAT_INSTREAM_RCCL", "0") == "1"
    if want:
        try:
            L = _load_rccl()
        except Exception as e:                             # noqa: BLE001
            err = e
    if not all_agree(want and L is not None):
        _instream_by_group[key] = None
        if want:
            import warnings
            warnings.warn(f"SPLAT_INSTREAM_RCCL=1 but not every rank can use the in-stream communicator ({err!r}); using torch.distributed")
        return None
    # 2. rank 0 draws the id (a local call); whether it could is agreed BEFORE the broadcast that ships it
    uid = None
    if rank == 0:
        try:
            u = _NcclUniqueId()
            rc = L.ncclGetUniqueId(C.byref(u))
            uid = C.string_at(C.byref(u), 128) if rc == 0 else None
        except Exception as e:                             # noqa: BLE001
            err = e
    ok = all_agree(rank != 0 or uid is not None)
    if ok:
        box = [uid]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        # 3. communicator + self-check (sum of rank + 1), agreed again
        good = False
        try:
            comm = InStreamRccl(rank, world, uid=box[0], group=group)
            probe = torch.full((8,), float(rank + 1), device=dev)
            comm.all_reduce(probe, InStreamRccl.SUM)
            good = bool((probe == world * (world + 1) / 2).all())
        except Exception as e:                             # noqa: BLE001  (fall back to the process group, loudly)
            err = e
        ok = all_agree(good)
    if not ok:
        comm = None
        import warnings
        warnings.warn(f"SPLAT_INSTREAM_RCCL=1 but the in-stream communicator is unavailable on some rank ({err!r}); using torch.distributed")
    _instream_by_group[key] = comm
    if group is None:
        _instream = comm
    return comm


def all_reduce_mean_flat(flat: torch.Tensor, group=None) -> None:
    """In-place mean over ranks of ONE flat gradient buffer -- the exchange step of the fused mapping iteration
    (splatam_amd.fused.FusedEngine.grad_flat is already laid out as the bucket: no packing)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    comm = instream(group)
    if comm is not None:
        comm.all_reduce(flat, InStreamRccl.AVG)
    elif dist.get_backend(group) == "nccl":          # RCCL averages inside the collective: no separate scaling kernel
        dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=group)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(dist.get_world_size(group))


def all_reduce_sum_flat(flat: torch.Tensor, group=None) -> None:
    """In-place sum over ranks of one flat gradient buffer (the caller divides by the number of views of all ranks)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        comm = instream(group)
        if comm is not None:
            comm.all_reduce(flat, InStreamRccl.SUM)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)


def any_rank(flag: bool, device=None, group=None) -> bool:
    """True on every rank when ``flag`` is true on at least one (one tiny all-reduce; a no-op for a single process)."""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return bool(int(t[0]))


def max_int(value: int, device=None, group=None) -> int:
    """The largest ``value`` over the ranks (a no-op for a single process)."""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t[0])


def tile_row_band(tile_rows: int, rank: int, world: int):
    """Rows [begin, end) of a ``tile_rows``-row tile grid that rank ``rank`` composites in tile-row-sharded tracking: contiguous,
    disjoint, covering, sizes differing by at most one."""
    if not 0 <= rank < world or world > tile_rows:
        raise ValueError(f"cannot give each of {world} ranks a band of a {tile_rows}-row tile grid (rank {rank})")
    return (rank * tile_rows) // world, ((rank + 1) * tile_rows) // world


def shard_views(num_views: int, rank: int, world: int):
    """Indices of the keyframe views rank ``rank`` renders (round-robin)."""
    return list(range(rank, num_views, world))


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_initialized() else 1


def get_rank(group=None) -> int:
    return dist.get_rank(group) if dist.is_initialized() else 0


def broadcast_pose(params, time_idx: int, src: int = 0, group=None) -> None:
    """Every rank tracks the same frame (replicas; the backward composite sums with float atomics, so the tracked poses
    differ in the last bits from rank to rank).  The map edits that follow tracking (add_new_gaussians thresholds the render at
    the tracked pose) must see ONE pose: rank ``src``'s 7 floats are broadcast into params['cam_unnorm_rots' / 'cam_trans'][..., t]."""
    if world_size(group) == 1:
        return
    with torch.no_grad():
        pose = torch.cat((params['cam_unnorm_rots'].detach()[0, :, time_idx], params['cam_trans'].detach()[0, :, time_idx])).contiguous()
        dist.broadcast(pose, src=src, group=group)
        params['cam_unnorm_rots'][0, :, time_idx] = pose[0:4]
        params['cam_trans'][0, :, time_idx] = pose[4:7]


def assert_replicated_count(n: int, what: str, device, group=None) -> None:
    """The replicas of the map must hold the same number of rows after every edit (the gradient all-reduce of the next
    mapping step would otherwise mix buffers of different length): all-reduce of (min, max) of the row count."""
    if world_size(group) == 1:
        return
    t = torch.tensor([n, -n], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    lo, hi = int(t[0]), -int(t[1])
    if lo != hi:
        raise RuntimeError(f"map replicas diverged after {what}: row counts {lo} .. {hi} across ranks")
