"""Worker of the multi-rank tests of the stand-alone FFT and doubled-domain types: FFTMPI3D on z-slabs against
scipy.fft.rfftn of the global array (the reference's own check, tests/.../test_fft_mpi_3d.py:105-115), and the
reference's unbounded Poisson solve assembled from FFTMPI3D + MPIDomainDoublingCommunicator3D
(UnboundedPoissonSolverMPI3D.py:133-167) against the oracle.  argv: grid "ZxYxX", precision single|double."""
import os
import sys

import numpy as np
import scipy.fft as spfft

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sopht-mpi_amd"), os.path.join(ROOT, "tests")]

from helpers import assert_close  # noqa: E402
from oracle.poisson import UnboundedPoissonSolver3D  # noqa: E402
from sopht_mpi_amd.numeric.eulerian_grid_ops import FFTMPI3D, MPIDomainDoublingCommunicator3D  # noqa: E402
from sopht_mpi_amd.utils import MPIConstruct3D  # noqa: E402
from sopht_mpi_amd.world import get_world  # noqa: E402

world = get_world()
rank, size = world.rank, world.size
n = tuple(int(v) for v in sys.argv[1].split("x"))
real_t = np.float64 if "double" in sys.argv[2:] else np.float32
g = 2
nz, ny, nx = n
construct = MPIConstruct3D(nz, ny, nx, real_t=real_t, rank_distribution=(0, 1, 1))
nzl, nyl = nz // size, ny // size
zs, ys = slice(rank * nzl, (rank + 1) * nzl), slice(rank * nyl, (rank + 1) * nyl)
rng = np.random.default_rng(5)          # the same global arrays on every rank
field = rng.standard_normal(n).astype(real_t)

# ---- the transform by itself
fft = FFTMPI3D(nz, ny, nx, construct, real_t=real_t)
assert fft.field_buffer.shape == (nzl, ny, nx) and fft.fourier_field_buffer.shape == (nz, nyl, nx // 2 + 1)
fft.field_buffer[...] = field[zs]
fft.forward(fft.field_buffer, fft.fourier_field_buffer)
want = spfft.rfftn(field.astype(np.float64))[:, ys, :]
got = fft.fourier_field_buffer.get()
scale = float(np.abs(want).max())
assert_close(got.real, want.real, real_t, scale=scale, factor=4.0, what="forward, real part")
assert_close(got.imag, want.imag, real_t, scale=scale, factor=4.0, what="forward, imaginary part")
back = fft.ctx.zeros((nzl, ny, nx), real_t)
fft.backward(fft.fourier_field_buffer, back)
assert_close(back.get(), field[zs], real_t, factor=4.0, what="backward(forward(field))")
assert np.array_equal(fft.fourier_field_buffer.get(), got), "backward must leave its input intact"

# ---- the reference's solve sequence on the doubled domain
x_range = 1.3
oracle = UnboundedPoissonSolver3D(nz, ny, nx, x_range=x_range, real_t=real_t)
rhs = np.zeros(tuple(m + 2 * g for m in n), dtype=real_t)
rhs[g:-g, g:-g, g:-g] = field
want_sol = np.zeros_like(rhs)
oracle.solve(want_sol, rhs, g)
dfft = FFTMPI3D(2 * nz, 2 * ny, 2 * nx, construct, real_t=real_t)
doubling = MPIDomainDoublingCommunicator3D(g, construct)
ctx = doubling.ctx
local_rhs = ctx.from_numpy(np.ascontiguousarray(rhs[rank * nzl : (rank + 1) * nzl + 2 * g]))
local_sol = ctx.field()
doubled = dfft.field_buffer
assert doubled.shape == (2 * nzl, 2 * ny, 2 * nx)
doubled[...] = 0
doubling.copy_to_doubled_domain(local_rhs, doubled)
# what the doubled slab must hold: planes [2 nzl rank, 2 nzl (rank + 1)) of the zero-padded global array
padded = np.zeros((2 * nz, 2 * ny, 2 * nx), dtype=real_t)
padded[:nz, :ny, :nx] = field
assert np.array_equal(doubled.get(), padded[2 * nzl * rank : 2 * nzl * (rank + 1)]), "copy_to_doubled_domain"
dfft.forward(doubled, dfft.fourier_field_buffer)
dys = slice(rank * 2 * nyl, (rank + 1) * 2 * nyl)
green_hat = oracle.fourier_greens_function_times_dx_cubed[:, dys, :]
conv = dfft.fourier_field_buffer.get() * green_hat
dfft.fourier_field_buffer[...] = conv.astype(dfft.complex_dtype)
dfft.backward(dfft.fourier_field_buffer, doubled)
doubling.copy_from_doubled_domain(doubled, local_sol)
got_sol = local_sol.get()[g:-g, g:-g, g:-g]
want_local = want_sol[g:-g, g:-g, g:-g][zs]
assert_close(got_sol, want_local, real_t, scale=float(np.abs(want_sol).max()), factor=1.5 + max(n) / 16.0,
             what="solve assembled from FFTMPI3D and the doubled-domain copies")
world.barrier()
print(f"rank {rank} ok, transport {ctx.transport}", flush=True)
