"""The exchange of a sharded call as ONE direct RCCL all-gather (optional: GPYREG_AMD_EXCHANGE=rccl).

`torch.distributed.all_gather_into_tensor` costs ~35 us of host time to ISSUE, whatever it moves -- most of the 55 us a
sharded call pays for its one exchange (profiles/r05_exchange_probe.txt).  This module talks to the RCCL library that ships
with PyTorch through ctypes: one communicator per process group (unique id made by the group's first rank and handed round
through the group itself), and per exchange: pinned host image -> device block (hipMemcpyAsync), `ncclAllGather` over xGMI,
device result -> pinned host image, all on one stream, one `hipStreamSynchronize`.  The buffers stay torch tensors.

Off by default: only a one-rank communicator can be exercised on the one-GPU boxes this repository is developed on
(`tests/test_gpu_sharding.py::test_direct_rccl_exchange_one_rank`); the default exchange goes through torch.distributed.
Any failure while setting it up falls back to that path (once, with a warning)."""

from __future__ import annotations

import ctypes as C
import os
import warnings

_H2D, _D2H = 1, 2
_NCCL_FLOAT64 = 8


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_byte * 128)]


_state = {"libs": None, "comms": {}, "failed": False}


def wanted() -> bool:
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
        hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        _state["libs"] = (rccl, hip)
    return _state["libs"]


class Comm:
    """One RCCL communicator for a torch.distributed process group (nccl backend), with a stream of its own."""

    def __init__(self, group):
        import torch
        import torch.distributed as dist

        rccl, hip = _libs()
        self.rccl, self.hip = rccl, hip
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        uid = _UniqueId()
        if self.rank == 0:
            rc = rccl.ncclGetUniqueId(C.byref(uid))
            if rc:
                raise RuntimeError("ncclGetUniqueId: " + rccl.ncclGetErrorString(rc).decode())
        box = [bytes(uid.internal)]
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(box, src=src, group=group)
        C.memmove(uid.internal, box[0], 128)
        self.stream = torch.cuda.Stream()
        self.comm = C.c_void_p()
        rc = rccl.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank)
        if rc:
            raise RuntimeError("ncclCommInitRank: " + rccl.ncclGetErrorString(rc).decode())

    def issue(self, hin, buf, out, hout):
        """pinned hin -> device buf -> all-gather into device out -> pinned hout, all on this communicator's stream"""
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


def comm_for(group):
    """The communicator of `group` (made on first use by ALL ranks of the group together), or None when the direct path is
    not wanted or could not be set up (the torch.distributed exchange is used then)."""
    if not wanted():
        return None
    key = id(group) if group is not None else 0
    if key not in _state["comms"]:
        try:
            _state["comms"][key] = Comm(group)
        except Exception as e:  # noqa: BLE001 - the exchange falls back to torch.distributed
            _state["failed"] = True
            warnings.warn(f"gpyreg_amd: the direct RCCL exchange could not be set up ({e}); using torch.distributed")
            return None
    return _state["comms"][key]
