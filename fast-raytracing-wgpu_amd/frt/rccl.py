"""RCCL called directly on the renderer's HIP streams (ctypes over librccl.so: ncclSend / ncclRecv in a group).

Why not torch.distributed's point-to-point calls for the halo rows: ProcessGroupNCCL runs every transfer on a stream of its own and, PER OPERATION of a batch,
records an event on the caller's stream and makes its stream wait for it — eight markers (~7 us each on the device) between T-merge and the next launch of a
strip that exchanges two kinds of rows with two neighbours, then a hop to its stream and a hop back: ~95 us of a 1/8 strip's 340 us frame
(profiles/r4_experiments/rccl_strips.md). Here the transfer is ONE grouped RCCL launch IN the stream that consumes the rows (the strip renderer's edge stream):
no events, no other stream. torch.distributed stays the plumbing: it carries the communicator's unique id to the ranks (and everything that is not per frame).

    comm = Comm.create(rank, world, device)          # collective over torch.distributed's default group: every rank calls it
    comm.exchange([(send_ptr, nbytes, peer), ...], [(recv_ptr, nbytes, peer), ...], stream_handle)
"""
import ctypes as C
import os

NCCL_UNIQUE_ID_BYTES = 128      # rccl.h:40
NCCL_UINT8 = 1                  # rccl.h: ncclUint8


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * NCCL_UNIQUE_ID_BYTES)]


_lib = None


def lib():
    """librccl.so as torch loaded it (the same handle: one RCCL in the process)."""
    global _lib
    if _lib is None:
        import torch
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        L = C.CDLL(path if os.path.exists(path) else "librccl.so")
        L.ncclGetErrorString.restype = C.c_char_p
        L.ncclGetErrorString.argtypes = [C.c_int]
        L.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]      # (the id travels BY VALUE)
        L.ncclCommDestroy.argtypes = [C.c_void_p]
        L.ncclSend.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclRecv.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        for f in (L.ncclGetUniqueId, L.ncclCommInitRank, L.ncclCommDestroy, L.ncclSend, L.ncclRecv, L.ncclGroupStart, L.ncclGroupEnd):
            f.restype = C.c_int
        _lib = L
    return _lib


class RcclError(RuntimeError):
    pass


def _check(rc, what):
    if rc != 0:
        raise RcclError(f"{what}: {lib().ncclGetErrorString(rc).decode()} ({rc})")


def _uid_struct():
    uid = _UniqueId()
    _check(lib().ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
    return uid


def unique_id():
    """128 opaque bytes (ncclGetUniqueId); the rank that draws them sends them to the others."""
    return C.string_at(C.byref(_uid_struct()), NCCL_UNIQUE_ID_BYTES)


class Comm:
    """One RCCL communicator over all ranks, used for grouped point-to-point transfers on caller-chosen HIP streams."""

    def __init__(self, rank, world, uid_bytes):
        assert len(uid_bytes) == NCCL_UNIQUE_ID_BYTES
        uid = _UniqueId()
        C.memmove(C.byref(uid), uid_bytes, NCCL_UNIQUE_ID_BYTES)
        self.rank, self.world = rank, world
        self._h = C.c_void_p()
        _check(lib().ncclCommInitRank(C.byref(self._h), world, uid, rank), "ncclCommInitRank")

    @classmethod
    def create(cls, rank, world, device):
        """Collective: rank 0 draws the unique id, torch.distributed's default group carries it. The HIP device must be current (hipSetDevice / torch.cuda.set_device)."""
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(device)
        box = [unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        return cls(rank, world, box[0])

    def exchange(self, sends, recvs, stream):
        """ONE grouped launch on HIP stream `stream` (a handle; 0 / None = the legacy default stream): sends / recvs = [(device pointer, bytes, peer rank)]."""
        if not sends and not recvs:
            return
        L = lib()
        s = C.c_void_p(stream or None)
        _check(L.ncclGroupStart(), "ncclGroupStart")
        try:
            for ptr, n, peer in sends:
                _check(L.ncclSend(C.c_void_p(ptr), n, NCCL_UINT8, peer, self._h, s), "ncclSend")
            for ptr, n, peer in recvs:
                _check(L.ncclRecv(C.c_void_p(ptr), n, NCCL_UINT8, peer, self._h, s), "ncclRecv")
        finally:
            _check(L.ncclGroupEnd(), "ncclGroupEnd")

    def destroy(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().ncclCommDestroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
