"""The exchange of a sharded call as ONE direct RCCL all-gather (optional: GPYREG_AMD_EXCHANGE=rccl).

`torch.distributed.all_gather_into_tensor` costs ~35 us of host time to ISSUE, whatever it moves -- most of the 55 us a
sharded call pays for its one exchange (profiles/r05_exchange_probe.txt).  This module talks to the RCCL library that ships
with PyTorch through ctypes: one communicator per process group (unique id made by the group's first rank and handed round
through the group itself), and per exchange: pinned host image -> device block (hipMemcpyAsync), `ncclAllGather` over xGMI,
device result -> pinned host image, all on one stream, one `hipStreamSynchronize`.  The buffers stay torch tensors.

EXPERIMENTAL and off by default: only a one-rank communicator can be exercised on the one-GPU boxes this repository is
developed on (`tests/test_bench_launch.py::test_direct_rccl_exchange_one_rank`); the default exchange goes through
torch.distributed.

Whether a process group uses the direct path is decided ONCE per group and by ALL its ranks together (`comm_for`): every
rank reports whether it wants the path and whether each stage of the set-up worked for it, the reports are combined with
an all-reduce (MIN) over the torch group, and the group takes the direct path on every rank or on none -- a rank that
fell back by itself would leave its peers waiting in `ncclAllGather` on a communicator it never entered (VERDICT r5 /
ADVICE r5).  A half-made communicator is destroyed.  `tests/test_sharding_gloo.py::test_direct_exchange_falls_back_on_every_rank_or_none`
forces a failure on one of two ranks and checks that both take the torch path."""

from __future__ import annotations

import ctypes as C
import os
import warnings

_H2D, _D2H = 1, 2
_NCCL_FLOAT64 = 8


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_byte * 128)]


# comms: process group -> Comm, or None once the group has DECIDED against the direct path (keyed by the group object
# itself, not its id(): an id can be reused after a group is garbage-collected; 0 stands for the default group)
_state = {"libs": None, "comms": {}, "failed": False}


def wanted() -> bool:
    """This rank's wish (the environment); what a group does is the agreement of all its ranks, see comm_for."""
    return os.environ.get("GPYREG_AMD_EXCHANGE", "torch").lower() == "rccl" and not _state["failed"]


def _libs():
    if _state["libs"] is None:
        import torch

        d = os.path.join(os.path.dirname(torch.__file__), "lib")
        rccl = C.CDLL(os.path.join(d, "librccl.so"))
        hip = C.CDLL(os.path.join(d, "libamdhip64.so"))
        rccl.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        rccl.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        rccl.ncclGetErrorString.restype = C.c_char_p
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        _state["libs"] = (rccl, hip)
    return _state["libs"]


class Comm:
    """One RCCL communicator for a torch.distributed process group (nccl backend), with a stream of its own.  Made in two
    stages so that the ranks can agree between them (comm_for): `__init__` is local (libraries, the unique id on the
    group's first rank), `connect` is collective (the id is handed round through the group, ncclCommInitRank)."""

    def __init__(self, group):
        import torch.distributed as dist

        rccl, hip = _libs()
        self.rccl, self.hip = rccl, hip
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.comm = None
        self.stream = None
        self.uid = _UniqueId()
        if self.rank == 0:
            rc = rccl.ncclGetUniqueId(C.byref(self.uid))
            if rc:
                raise RuntimeError("ncclGetUniqueId: " + rccl.ncclGetErrorString(rc).decode())

    def connect(self, group):
        import torch
        import torch.distributed as dist

        box = [bytes(self.uid.internal)]
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(box, src=src, group=group)
        C.memmove(self.uid.internal, box[0], 128)
        self.stream = torch.cuda.Stream()
        comm = C.c_void_p()
        rc = self.rccl.ncclCommInitRank(C.byref(comm), self.world, self.uid, self.rank)
        if rc:
            raise RuntimeError("ncclCommInitRank: " + self.rccl.ncclGetErrorString(rc).decode())
        self.comm = comm

    def destroy(self):
        if self.comm is not None:
            self.rccl.ncclCommDestroy(self.comm)
            self.comm = None

    def issue(self, hin, buf, out, hout):
        """pinned hin -> device buf -> all-gather into device out -> pinned hout, all on this communicator's stream"""
        import torch

        # buf / out come from torch's caching allocator on the CURRENT stream: whatever that stream still has queued on
        # them (an earlier exchange's copies, their allocation itself) must be over before this stream touches them, and
        # the allocator must know this stream uses them
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        buf.record_stream(self.stream)
        out.record_stream(self.stream)
        s = self.stream.cuda_stream
        nb = buf.numel() * 8
        if self.hip.hipMemcpyAsync(buf.data_ptr(), hin.data_ptr(), nb, _H2D, s):
            raise RuntimeError("hipMemcpyAsync (upload) failed")
        rc = self.rccl.ncclAllGather(buf.data_ptr(), out.data_ptr(), buf.numel(), _NCCL_FLOAT64, self.comm, s)
        if rc:
            raise RuntimeError("ncclAllGather: " + self.rccl.ncclGetErrorString(rc).decode())
        if self.hip.hipMemcpyAsync(hout.data_ptr(), out.data_ptr(), nb * self.world, _D2H, s):
            raise RuntimeError("hipMemcpyAsync (download) failed")

    def wait(self):
        if self.hip.hipStreamSynchronize(self.stream.cuda_stream):
            raise RuntimeError("hipStreamSynchronize failed")


def _all_ranks(ok: bool, group) -> bool:
    """True when `ok` holds on EVERY rank of the group (an all-reduce MIN over the torch group itself)."""
    import torch
    import torch.distributed as dist

    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


def comm_for(group):
    """The communicator of `group`, or None when the group uses the torch.distributed exchange.  COLLECTIVE on first use:
    every rank of a device group calls it from its first exchange (sharding._Gather), whatever its own environment says,
    and the answer is the same on all of them -- the direct path is taken when every rank wants it AND every stage of the
    set-up worked on every rank; otherwise on none (a half-made communicator is destroyed, one warning per process)."""
    key = group if group is not None else 0
    comms = _state["comms"]
    if key in comms:
        return comms[key]
    comm, err = None, None
    stage = "not requested on every rank (GPYREG_AMD_EXCHANGE)"
    ok = _all_ranks(wanted(), group)
    if ok:
        stage = "loading RCCL / making the unique id"
        try:
            comm = Comm(group)
        except Exception as e:  # noqa: BLE001 - reported to the peers, then the whole group falls back
            err = e
        ok = _all_ranks(err is None, group)
    if ok:
        stage = "ncclCommInitRank"
        try:
            comm.connect(group)
        except Exception as e:  # noqa: BLE001
            err = e
        ok = _all_ranks(err is None, group)
    if not ok:
        if comm is not None:
            try:
                comm.destroy()
            except Exception:  # noqa: BLE001 - nothing more to do with it
                pass
        if wanted():
            _state["failed"] = True
            warnings.warn("gpyreg_amd: the direct RCCL exchange is not used by this process group (" + stage
                          + (f": {err}" if err is not None else ": another rank reported a failure or did not ask for it")
                          + "); every rank uses torch.distributed")
        comm = None
    comms[key] = comm
    return comm
