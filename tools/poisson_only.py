"""Profiling helper: a few vector Poisson solves on a grid^3 field (pruned back end), with the
in-library per-kernel event timings printed at the end."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sopht-mpi_amd")]
from sopht_mpi_amd import _lib
from sopht_mpi_amd.numeric.eulerian_grid_ops import UnboundedPoissonSolverMPI3D
from sopht_mpi_amd.utils import MPIConstruct3D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
c = MPIConstruct3D(n, n, n, real_t=np.float32)
p = UnboundedPoissonSolverMPI3D(n, n, n, c, 2, x_range=1.0, real_t=np.float32, backend="pruned")
ctx = p.ctx
rhs = ctx.field(3); sol = ctx.field(3)
rhs[...] = np.random.default_rng(0).random((3,) + ctx.padded_shape, dtype=np.float32)
p.vector_field_solve(sol, rhs)
ctx.sync()
import time
t0 = time.perf_counter()
for _ in range(reps):
    p.vector_field_solve(sol, rhs)
ctx.sync()
print(f"  solve wall ms (no event timing): {(time.perf_counter() - t0) / reps * 1e3:.3f}")
_lib.call("smi_prof_enable", ctx.handle, 1)
for _ in range(reps):
    p.vector_field_solve(sol, rhs)
ctx.sync()
for k, v in sorted(_lib.prof_collect(ctx).items(), key=lambda kv: -kv[1]["ms"]):
    print(f"  {k:18s} n={v['launches']:3d} avg_ms={v['ms'] / v['launches']:.4f}")
print("done", float(np.abs(sol.get()).max()))
