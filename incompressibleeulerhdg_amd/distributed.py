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


def _probe_main():
    """Child process of `probe_transport`: one tiny strip-partitioned HDG-IMEX step over `backend`."""
    import sys

    import torch.distributed as dist

    backend = sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")

    def bcast(obj):
        lst = [obj]
        dist.broadcast_object_list(lst, src=0)
        return lst[0]

    import numpy as np

    from .mesh import UnitSquareMesh
    from .model_problems import TaylorGreen
    from .timesteppers import IncompressibleEulerHDGIMEXSSP2_332

    import torch

    nx = 8 * world
    token = make_comm_token(backend, rank, bcast)
    device = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    ts = IncompressibleEulerHDGIMEXSSP2_332(UnitSquareMesh(nx, nx), 1, 0.25 / nx, device=device,
                                            **comm_kwargs(backend, rank, world, token))
    mp = TaylorGreen(ts._V_Q, ts._V_p)
    Q, p = ts.solve(*mp.initial_condition(), None, mp.f_rhs(), 0.25 / nx, fused=True)
    ok = bool(np.all(np.isfinite(Q.dat.data)) and np.all(np.isfinite(p.dat.data)))
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 3)


def probe_transport(backend, timeout=180):
    """Run one tiny distributed step over `backend` in a CHILD process of every rank (same RANK /
    WORLD_SIZE / LOCAL_RANK, rendezvous port MASTER_PORT + 17) and report whether it finished in time.
    A transport that cannot initialise - or hangs - on this machine then costs a timeout, not the run."""
    import subprocess
    import sys

    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + 17)
    env.pop("TORCHELASTIC_RUN_ID", None)
    env["TORCHELASTIC_USE_AGENT_STORE"] = "False"  # rank 0 of the child group hosts its own store on the new port
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    try:
        r = subprocess.run([sys.executable, "-m", "incompressibleeulerhdg_amd.distributed", "--probe", backend], env=env,
                           timeout=timeout, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        if r.returncode != 0:
            sys.stderr.write(r.stderr.decode(errors="replace")[-1500:])
        return r.returncode == 0
    except subprocess.TimeoutExpired:
        return False


if __name__ == "__main__":
    import sys

    if len(sys.argv) >= 3 and sys.argv[1] == "--probe":
        _probe_main()
