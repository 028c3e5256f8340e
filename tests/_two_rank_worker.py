"""Worker of the world-size-2 CPU test (launched by torch.distributed.run, sockets only)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sopht-mpi_amd")]

from oracle.decomp import ZSlabs  # noqa: E402
from sopht_mpi_amd.utils import MPIConstruct3D, MPILagrangianFieldCommunicator3D  # noqa: E402
from sopht_mpi_amd.world import get_world  # noqa: E402

world = get_world()
rank, size = world.rank, world.size
assert size == 2
# collectives
assert world.allreduce(rank + 1.0, "sum") == 3.0
assert world.allreduce(rank + 1.0, "min") == 1.0
assert world.allreduce(rank + 1.0, "max") == 2.0
uid = world.bcast_bytes(bytes(range(128)) if rank == 0 else bytes(128), root=0)
assert uid == bytes(range(128))
# the halo pattern of the host-staged transport: each rank sends g planes to its neighbour
g, plane = 2, 5 * 7
mine = np.full(g * plane, float(rank + 1), dtype=np.float32)
got = np.empty_like(mine)
peer = 1 - rank
world.sendrecv({peer: mine}, {peer: got})
assert np.all(got == float(peer + 1))
# topology + Lagrangian ownership, scatter and gather
n, dx = (16, 8, 8), np.float32(1.0 / 8)
c = MPIConstruct3D(*n, real_t=np.float32, world=world)
assert list(c.local_grid_size) == [8, 8, 8] and c.grid.coords[0] == rank
comm = MPILagrangianFieldCommunicator3D(dx, np.float32(dx / 2), c)
rng = np.random.default_rng(5)
pos = rng.uniform(0.1, 0.9, (3, 31)) * np.array([1.0, 1.0, 2.0]).reshape(3, 1)
comm.map_lagrangian_nodes_based_on_position(pos if rank == 0 else None)
want = ZSlabs(n, 2, 2).lagrangian_owner(pos, dx, np.float32(dx / 2))
assert np.array_equal(comm.rank_address, want)
local = np.zeros((3, comm.local_num_lag_nodes))
comm.scatter_global_field(local, pos if rank == 0 else None)
assert np.array_equal(local, pos[:, want == rank])
back = np.zeros_like(pos)
comm.gather_local_field(back if rank == 0 else None, 2.0 * local)
if rank == 0:
    assert np.array_equal(back, 2.0 * pos)
# checkpoint: both ranks write their slab of one file at the same time, then read it back (mpi_io.py:271-554)
import tempfile  # noqa: E402

from sopht_mpi_amd.utils import MPIIO, h5lite  # noqa: E402

tmp = world.bcast_object(tempfile.mkdtemp(prefix="smi_io_") if rank == 0 else None, root=0)
path = os.path.join(tmp, "two_ranks.h5")
grid_size, gs = np.array(n), 2
glob = np.random.default_rng(9).random((3,) + n).astype(np.float32)
padded = tuple(int(k) + 2 * gs for k in c.local_grid_size)
inner = (slice(None),) + (slice(gs, -gs),) * 3
mine = np.zeros((3,) + padded, dtype=np.float32)
mine[inner] = glob[:, rank * 8:(rank + 1) * 8]
lag = pos.astype(np.float32) if rank == 1 else np.zeros((3, 31), dtype=np.float32)   # rank 1 holds the body


def make_io(vector, grid):
    io = MPIIO(mpi_construct=c, real_dtype=np.float32)
    io.define_eulerian_grid(origin=np.full(3, dx / 2), dx=np.full(3, dx), grid_size=grid_size, ghost_size=gs)
    io.add_as_eulerian_fields_for_io(vorticity=vector)
    io.add_as_lagrangian_fields_for_io(lagrangian_grid=grid, lagrangian_grid_master_rank=1,
                                       lagrangian_grid_name="body", force=grid)
    return io


make_io(mine, lag).save(path, time=0.75)
whole = h5lite.Reader(path)
for k in range(3):
    assert np.array_equal(whole[f"Eulerian/Vector/vorticity_{k}"][...], glob[k][None])
assert np.array_equal(whole["Lagrangian/body/Grid"][...], pos.astype(np.float32).T)
back_v, back_l = np.zeros_like(mine), np.zeros_like(lag)
assert make_io(back_v, back_l).load(path) == 0.75
assert np.array_equal(back_v[inner], mine[inner])
assert np.array_equal(back_l, lag)
world.barrier()
if rank == 0:
    import shutil

    shutil.rmtree(tmp)
print(f"rank {rank} ok")
