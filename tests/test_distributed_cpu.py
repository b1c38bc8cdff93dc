"""CPU tests of the N > 1 launch path (no GPU): strip partition arithmetic and the token rendezvous
over torch.distributed with the gloo backend, world_size 2."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from incompressibleeulerhdg_amd.distributed import make_comm_token, comm_kwargs, strip_rows
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
def bcast(obj):
    lst = [obj]
    dist.broadcast_object_list(lst, src=0)
    return lst[0]
tok = make_comm_token("shm", rank, bcast)
assert isinstance(tok, str) and tok.startswith("/hdg_")
gathered = [None] * world
dist.all_gather_object(gathered, tok)
assert all(t == gathered[0] for t in gathered)           # every rank received rank 0's token
rid = make_comm_token("rccl", rank, bcast)               # 128-byte ncclUniqueId created on rank 0 (no GPU needed)
assert isinstance(rid, bytes) and len(rid) == 128
dist.all_gather_object(gathered, rid)
assert all(t == gathered[0] for t in gathered)
kw = comm_kwargs("shm", rank, world, tok)
assert kw == dict(rank=rank, nranks=world, comm_backend="shm", comm_token=tok)
j0, j1 = strip_rows(16, rank, world)
rows = [None] * world
dist.all_gather_object(rows, (j0, j1))
assert rows[0][0] == 0 and rows[-1][1] == 16 and all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))
dist.barrier()
dist.destroy_process_group()
print("OK", rank)
'''


def test_token_rendezvous_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0, out.decode()[-2000:]
        assert f"OK {r}" in out.decode()


def test_strip_partition_arithmetic():
    from incompressibleeulerhdg_amd.distributed import halo_message_bytes, strip_rows

    assert strip_rows(1024, 3, 8) == (384, 512)
    with pytest.raises(ValueError):
        strip_rows(10, 0, 4)
    # velocity facet data 64 KiB per cut per direction would be the minimum;
    # the engine sends whole cell rows (both shapes, all modes) -> 2 * 20 * 2 * 1024 * 8 B
    v, t = halo_message_bytes(1024, 2)
    assert v == 8 * 2 * 10 * 2 * 1024 and t == 8 * 3 * 3 * 1040
