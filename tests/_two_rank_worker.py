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
world.barrier()
print(f"rank {rank} ok")
