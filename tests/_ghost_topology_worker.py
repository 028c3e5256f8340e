"""Worker of the general-topology test: the ghost, field and Lagrangian communicators on every rank distribution the
reference's own tests use (tests/test_utils/test_mpi_utils_3d.py:14-125, 795-960, test_mpi_utils_2d.py:14-101), x
divided and periodic domains included.  Unlike the reference's test, whose ranks all hold the same random block, every
rank holds its own block of one seeded global array, so a message that reaches the wrong neighbour shows.

usage: _ghost_topology_worker.py <dim> [quick]"""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sopht-mpi_amd"), os.path.join(ROOT, "tests")]

from sopht_mpi_amd import utils  # noqa: E402
from sopht_mpi_amd.world import get_world  # noqa: E402

world = get_world()
rank, size = world.rank, world.size
dim = int(sys.argv[1])
quick = "quick" in sys.argv[2:]
SENTINEL = -7.0

if dim == 3:
    Construct, Ghost, Field, Lag = (utils.MPIConstruct3D, utils.MPIGhostCommunicator3D, utils.MPIFieldCommunicator3D,
                                    utils.MPILagrangianFieldCommunicator3D)
    distributions = [(0, 1, 1), (1, 0, 1), (1, 1, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)]
    aspects = [(1, 1, 1), (1, 1, 2), (2, 2, 1)] if not quick else [(1, 2, 1)]
else:
    Construct, Ghost, Field, Lag = (utils.MPIConstruct2D, utils.MPIGhostCommunicator2D, utils.MPIFieldCommunicator2D,
                                    utils.MPILagrangianFieldCommunicator2D)
    distributions = [(1, 0), (0, 1), (2, 2)] if size == 4 else [(1, 0), (0, 1)]
    aspects = [(1, 1), (1, 2), (2, 1)] if not quick else [(2, 1)]
if quick:
    distributions = distributions[2:4] if dim == 3 else distributions[-2:]

cases = 0
constructs = {}
for dist, aspect, periodic, real_t, ghost, full in itertools.product(
        distributions, aspects, (True, False), (np.float32, np.float64), (1, 2), (True, False)):
    if quick and (ghost == 1 or real_t is np.float64):
        continue
    n = tuple(8 * a for a in aspect)
    key = (dist, aspect, periodic, real_t)
    if key not in constructs:
        try:
            constructs[key] = Construct(*n, periodic_domain=periodic, real_t=real_t, rank_distribution=dist)
        except RuntimeError:
            constructs[key] = None   # this rank count does not divide this grid that way
    construct = constructs[key]
    if construct is None:
        continue
    topo = tuple(int(t) for t in construct.grid_topology)
    # 3D: the operators' own contexts serve a divided x axis; periodic domains (and, in 2D, divided columns) go
    # through the general block exchange
    general = periodic or (dim == 2 and topo[-1] > 1)
    assert construct.general_topology == general
    local = tuple(int(v) for v in construct.local_grid_size)
    coords = tuple(int(v) for v in construct.grid.coords)
    if min(local) < (ghost if general else 2 * ghost):
        continue   # a block sends `ghost` interior layers; the operators' contexts need two per ghost layer
    ghost_comm = Ghost(ghost_size=ghost, mpi_construct=construct, full_exchange=full)
    ctx = construct.device_context(ghost, exchange_only=True)
    field_comm = Field(ghost_size=ghost, mpi_construct=construct, master_rank=1 % size)

    rng = np.random.default_rng(100 * dim + ghost)
    glob = rng.random((dim,) + n).astype(real_t)       # the same on every rank
    inner = (slice(ghost, -ghost),) * dim
    padded = tuple(v + 2 * ghost for v in local)
    if dim == 2:
        scalar, vector = ctx.zeros(padded), ctx.zeros((2,) + padded)
    else:
        scalar, vector = ctx.field(), ctx.field(3)
    scalar[...] = np.full(padded, SENTINEL, dtype=real_t)
    vector[...] = np.full((dim,) + padded, SENTINEL, dtype=real_t)
    # scatter from the master rank (blocks along every divided axis), gather back
    field_comm.scatter_global_scalar_field(scalar, glob[0] if rank == field_comm.master_rank else None)
    field_comm.scatter_global_vector_field(vector, glob if rank == field_comm.master_rank else None)
    block = tuple(slice(c * v, (c + 1) * v) for c, v in zip(coords, local))
    assert np.array_equal(scalar.get()[inner], glob[0][block]), (dist, "scatter")
    back = np.zeros_like(glob) if rank == field_comm.master_rank else None
    field_comm.gather_local_vector_field(back, vector)
    if rank == field_comm.master_rank:
        assert np.array_equal(back, glob), (dist, "gather")

    ghost_comm.exchange_scalar_field_init(scalar)
    ghost_comm.exchange_vector_field_init(vector)
    ghost_comm.exchange_finalise()

    # what the padded block must hold: the global array continued periodically, or the untouched sentinel outside
    def expected(g_arr):
        if periodic:
            ext = np.pad(g_arr, ghost, mode="wrap")
        else:
            ext = np.pad(g_arr, ghost, mode="constant", constant_values=SENTINEL)
        return ext[tuple(slice(c * v, (c + 1) * v + 2 * ghost) for c, v in zip(coords, local))]

    got = [scalar.get()] + list(vector.get())
    want = [expected(glob[0])] + [expected(glob[k]) for k in range(dim)]
    for a, b in zip(got, want):
        # faces: ghost layers along one axis, interior along the others
        for axis in range(dim):
            for side in (slice(0, ghost), slice(-ghost, None)):
                idx = tuple(side if d == axis else slice(ghost, -ghost) for d in range(dim))
                assert np.array_equal(a[idx], b[idx]), (dist, aspect, ghost, periodic, full, "face", axis)
        if full and general:
            # edges and vertices too: the whole padded block
            assert np.array_equal(a, b), (dist, aspect, ghost, periodic, full, "edges / vertices")
    if not general:
        # a plain decomposition still runs the operators' own exchange, and the context property hands it out
        assert ghost_comm.ctx is ctx
    else:
        try:
            ghost_comm.ctx
            raise AssertionError("operators must not get a context on a general decomposition")
        except NotImplementedError:
            pass

    # Lagrangian nodes: every rank owns the nodes inside its block (utils/mpi_utils_3d.py:1352-1384)
    dx = 1.0 / n[-1]
    lag = Lag(eul_grid_dx=dx, eul_grid_coord_shift=dx / 2, mpi_construct=construct, master_rank=0, real_t=np.float64)
    m = 40
    extent = np.array([v * dx for v in n[::-1]])               # x, y(, z) extents
    pos = dx / 2 + rng.random((dim, m)) * (extent.reshape(dim, 1) - dx)
    lag.map_lagrangian_nodes_based_on_position(pos if rank == 0 else None)
    lo = np.array([c * v * dx + dx / 2 for c, v in zip(coords, local)])[::-1].reshape(dim, 1)
    hi = np.array([(c + 1) * v * dx + dx / 2 for c, v in zip(coords, local)])[::-1].reshape(dim, 1)
    mine = np.all((pos >= lo) & (pos < hi), axis=0)
    assert np.array_equal(np.where(lag.rank_address == rank)[0], np.where(mine)[0]), (dist, "owner map")
    local_pos = np.zeros((dim, lag.local_num_lag_nodes))
    lag.scatter_global_field(local_pos, pos if rank == 0 else None)
    assert np.array_equal(local_pos, pos[:, mine])
    gathered = np.zeros((dim, m)) if rank == 0 else None
    lag.gather_local_field(gathered, local_pos)
    if rank == 0:
        assert np.array_equal(gathered, pos)
    cases += 1

world.barrier()
print(f"rank {rank} ok, {cases} cases, transport {ctx.transport}", flush=True)
