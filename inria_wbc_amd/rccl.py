"""A raw `ncclComm_t` for `wbcqp_allgather_tau` (include/wbcqp.h) -- SURVEY 8(e) / kernel K6: the one optional exchange
step of the path, an all-gather of the joint torques when a caller wants the whole batch on every rank.

The C ABI takes a caller-supplied communicator (it never creates one: the caller owns the job's topology).  PyTorch does not
hand out the communicator behind `torch.distributed`, so the Python host side makes its own on the same librccl the process
already has loaded (PyTorch's), through ctypes: rank 0 draws the unique id, the job's existing process group carries it to the
other ranks, every rank calls ncclCommInitRank.  One process per GPU, as everywhere in this repo.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_lib = None


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_ubyte * 128)]  # NCCL_UNIQUE_ID_BYTES (c_ubyte: a c_char array field would be cut at the first NUL)


def lib():
    """librccl as loaded by (or beside) PyTorch; raises OSError when there is none."""
    global _lib
    if _lib is not None:
        return _lib
    names = []
    try:
        import torch
        names.append(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
    except Exception:  # noqa: BLE001
        pass
    names += ["librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"]
    err: Optional[Exception] = None
    for n in names:
        try:
            _lib = C.CDLL(n, mode=C.RTLD_GLOBAL)
            break
        except OSError as e:
            err = e
    if _lib is None:
        raise OSError("librccl not loadable: %s" % err)
    _lib.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    _lib.ncclGetUniqueId.restype = C.c_int
    _lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    _lib.ncclCommInitRank.restype = C.c_int
    _lib.ncclCommDestroy.argtypes = [C.c_void_p]
    _lib.ncclCommDestroy.restype = C.c_int
    _lib.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    _lib.ncclCommCount.restype = C.c_int
    _lib.ncclGetErrorString.argtypes = [C.c_int]
    _lib.ncclGetErrorString.restype = C.c_char_p
    return _lib


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, (lib().ncclGetErrorString(rc) or b"?").decode()))


def _bootstrap_env():
    """One node, one process per GPU: the bootstrap (unique-id exchange) runs over the loopback interface unless the caller
    chose one -- GPU boxes without an outward-facing interface otherwise fail in ncclCommInitRank with "remote process exited
    or there was a network error".  The data path is xGMI / P2P either way."""
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def unique_id() -> bytes:
    _bootstrap_env()
    uid = UniqueId()
    _check(lib().ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
    raw = bytes(uid.internal)
    assert len(raw) == 128
    return raw


def comm_init_rank(uid: bytes, nranks: int, rank: int) -> int:
    """ncclCommInitRank on the current HIP device; returns the ncclComm_t as an integer"""
    _bootstrap_env()
    assert len(uid) == 128, "a unique id is 128 bytes"
    u = UniqueId()
    C.memmove(C.byref(u), uid, 128)
    comm = C.c_void_p()
    _check(lib().ncclCommInitRank(C.byref(comm), int(nranks), u, int(rank)), "ncclCommInitRank")
    return int(comm.value)


def comm_count(comm: int) -> int:
    """ncclCommCount: how many ranks RCCL itself says the communicator spans (bench.py puts it into config.rccl_nranks)"""
    n = C.c_int(-1)
    _check(lib().ncclCommCount(C.c_void_p(comm), C.byref(n)), "ncclCommCount")
    return int(n.value)


def comm_destroy(comm: int):
    if comm:
        lib().ncclCommDestroy(C.c_void_p(comm))


def comm_from_torch(dist, rank: int, world: int, device) -> int:
    """A communicator over the ranks of an initialised torch.distributed job (the unique id travels over that job)."""
    import torch
    if world == 1:
        return comm_init_rank(unique_id(), 1, 0)
    buf = torch.zeros(128, dtype=torch.uint8, device=device)
    if rank == 0:
        buf.copy_(torch.frombuffer(bytearray(unique_id()), dtype=torch.uint8))
    dist.broadcast(buf, src=0)
    torch.cuda.synchronize()
    return comm_init_rank(bytes(buf.cpu().numpy().tobytes()), world, rank)
