"""Launch-side plumbing of the strip-partitioned engine: one process per GPU, `torch.distributed`
only for the rendezvous (who am I, broadcast of the transport token) and for bench barriers.

The data path never goes through Python: halo rows, Krylov scalars and the coarse-grid residual move
inside libhdg_mi355x.so (RCCL on the engine's stream, or the shared-memory transport).
"""
import ctypes
import os
import uuid

__all__ = ["strip_rows", "make_comm_token", "comm_kwargs"]


def strip_rows(ny, rank, nranks):
    """Cell rows [j0, j1) owned by `rank` in the non-overlapping strip partition."""
    if ny % nranks != 0:
        raise ValueError(f"ny = {ny} is not divisible by the number of ranks {nranks}")
    n = ny // nranks
    return rank * n, (rank + 1) * n


def halo_message_bytes(nx, degree):
    """Bytes per neighbour and direction: (velocity rows, trace rows) -- SURVEY.md Appendix B."""
    nu = (degree + 2) * (degree + 3) // 2
    pitch = (nx + 1 + 15) // 16 * 16
    return 8 * 2 * nu * 2 * nx, 8 * 3 * (degree + 1) * pitch


def make_comm_token(backend, rank, broadcast):
    """Create the transport token on rank 0 and hand it to every rank through `broadcast(obj) -> obj`
    (e.g. a torch.distributed.broadcast_object_list wrapper)."""
    token = None
    if rank == 0:
        if backend == "rccl":
            from . import _lib

            buf = ctypes.create_string_buffer(128)
            rc = _lib.load_library().hdg_rccl_unique_id(buf)
            if rc != 0:
                raise RuntimeError("hdg_rccl_unique_id failed")
            token = buf.raw
        elif backend == "shm":
            token = f"/hdg_{os.getpid()}_{uuid.uuid4().hex[:10]}"
        else:
            raise ValueError(backend)
    return broadcast(token)


def comm_kwargs(backend, rank, nranks, token):
    """Keyword arguments for the timestepper / Engine constructors."""
    if nranks == 1:
        return {}
    return dict(rank=rank, nranks=nranks, comm_backend=backend, comm_token=token)
