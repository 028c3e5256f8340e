"""Timing helper: one vector Poisson solve against three scalar solves on the same fields (the per-component launches of
the multi-rank pipeline, on one GPU), with the in-library per-kernel event timings.  python tools/poisson_per_component.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sopht-mpi_amd")]
from sopht_mpi_amd import _lib
from sopht_mpi_amd.numeric.eulerian_grid_ops import UnboundedPoissonSolverMPI3D
from sopht_mpi_amd.utils import MPIConstruct3D
n = 512
c = MPIConstruct3D(n, n, n, real_t=np.float32)
p = UnboundedPoissonSolverMPI3D(n, n, n, c, 2, x_range=1.0, real_t=np.float32, backend="pruned")
ctx = p.ctx
rhs = ctx.field(3); sol = ctx.field(3)
rhs[...] = np.random.default_rng(0).random((3,) + ctx.padded_shape, dtype=np.float32)
for mode in ("vector", "scalar x3"):
    for rep in range(2):
        _lib.call("smi_prof_enable", ctx.handle, 1); _lib.call("smi_prof_reset", ctx.handle)
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(3):
            if mode == "vector":
                p.vector_field_solve(sol, rhs)
            else:
                for k in range(3):
                    p.solve(sol[k], rhs[k])
        ctx.sync(); el = (time.perf_counter() - t0) / 3 * 1e3
    prof = _lib.prof_collect(ctx)
    print(mode, f"wall {el:.3f} ms", {k: round(v["ms"] / 3, 3) for k, v in prof.items()})
