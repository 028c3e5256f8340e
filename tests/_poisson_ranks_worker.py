"""Worker of the multi-rank Poisson test: UnboundedPoissonSolverMPI3D on z-slabs (or on pencils, through the slab
route) against the oracle's solve of the gathered array, scalar and vector solves, for every back end that takes the
grid (the reference's own check: tests/.../test_unbounded_poisson_solver_mpi_3d.py solves on the ranks and compares
with the serial solver).  argv: grid "ZxYxX", then any of: double, pencil."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sopht-mpi_amd"), os.path.join(ROOT, "tests")]

from helpers import assert_close  # noqa: E402
from oracle.poisson import UnboundedPoissonSolver3D  # noqa: E402
from sopht_mpi_amd.numeric.eulerian_grid_ops import UnboundedPoissonSolverMPI3D  # noqa: E402
from sopht_mpi_amd.utils import MPIConstruct3D, MPIFieldCommunicator3D  # noqa: E402
from sopht_mpi_amd.world import get_world  # noqa: E402

world = get_world()
rank, size = world.rank, world.size
n = tuple(int(v) for v in sys.argv[1].split("x"))
real_t = np.float64 if "double" in sys.argv[2:] else np.float32
distribution = (0, 0, 1) if "pencil" in sys.argv[2:] else (0, 1, 1)
g, x_range = 2, 1.3
nz, ny, nx = n
construct = MPIConstruct3D(nz, ny, nx, real_t=real_t, rank_distribution=distribution)
comm = MPIFieldCommunicator3D(g, construct)
rng = np.random.default_rng(11)          # the same global arrays on every rank
glob = rng.standard_normal((3,) + n).astype(real_t)
oracle = UnboundedPoissonSolver3D(nz, ny, nx, x_range=x_range, real_t=real_t)
rhs_pad = np.zeros((3,) + tuple(m + 2 * g for m in n), dtype=real_t)
rhs_pad[(slice(None),) + (slice(g, -g),) * 3] = glob
want = np.zeros_like(rhs_pad)
oracle.vector_field_solve(want, rhs_pad, g)
want = want[(slice(None),) + (slice(g, -g),) * 3]

ran = []
for backend in ("pruned", "rocfft"):
    if not UnboundedPoissonSolverMPI3D.backend_supports(backend, nz, ny, nx, nranks=size):
        continue
    solver = UnboundedPoissonSolverMPI3D(nz, ny, nx, construct, g, x_range=x_range, real_t=real_t, backend=backend)
    ctx = solver.pencil_ctx or solver.ctx
    rhs, sol = ctx.field(3), ctx.field(3)
    comm.scatter_global_vector_field(rhs, glob)
    solver.vector_field_solve(solution_vector_field=sol, rhs_vector_field=rhs)
    got = np.zeros((3,) + n, dtype=real_t) if rank == 0 else None
    comm.gather_local_vector_field(got, sol)
    # the scalar solve, on the second component
    rhs1, sol1 = ctx.field(), ctx.field()
    comm.scatter_global_scalar_field(rhs1, glob[1])
    solver.solve(solution_field=sol1, rhs_field=rhs1)
    got1 = np.zeros(n, dtype=real_t) if rank == 0 else None
    comm.gather_local_scalar_field(got1, sol1)
    if rank == 0:
        # white noise: the pruned single-precision solve is held to the bound of the single-rank test
        factor = (1 + 0.025 * max(n)) if (backend == "pruned" and real_t is np.float32) else 4.0
        assert_close(got, want, real_t, factor=factor, what=f"{backend} vector solve on {size} ranks")
        assert_close(got1, want[1], real_t, factor=factor, what=f"{backend} scalar solve on {size} ranks")
    ran.append(backend)
    del solver
assert "rocfft" in ran
world.barrier()
print(f"rank {rank} ok, back ends {ran}, transport {construct.device_context(g).transport}", flush=True)
