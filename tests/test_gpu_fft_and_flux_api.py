"""GPU parity of the helper operators around the Poisson solve and of the flux-only advection
operator:

* FFTMPI3D / FFTMPI2D against scipy.fft.rfftn / irfftn, the known answer the reference's own
  test uses (tests/test_numeric/test_eulerian_grid_ops/test_poisson_solver_3d/test_fft_mpi_3d.py:105-115);
* MPIDomainDoublingCommunicator3D / 2D copies;
* the reference's solve sequence assembled from those pieces (UnboundedPoissonSolverMPI3D.py:133-167)
  against the library's own solver and the oracle;
* gen_advection_flux_conservative_eno3_pyst_mpi_kernel_{2,3}d against the oracle, including the
  cells the operator must leave alone.
"""
import numpy as np
import pytest
import scipy.fft

from helpers import assert_close, rand_field
from oracle import stencils_2d as o2
from oracle import stencils_3d as o3
from oracle.poisson import UnboundedPoissonSolver3D

pytestmark = pytest.mark.gpu

PRECISIONS = [np.float32, np.float64]


def _construct3(shape, real_t):
    from sopht_mpi_amd.utils import MPIConstruct3D

    return MPIConstruct3D(*shape, real_t=real_t)


def _construct2(shape, real_t):
    from sopht_mpi_amd.utils import MPIConstruct2D

    return MPIConstruct2D(*shape, real_t=real_t)


@pytest.mark.parametrize("real_t", PRECISIONS)
@pytest.mark.parametrize("shape", [(8, 8, 8), (8, 12, 16), (16, 10, 6)])
def test_fft_mpi_3d_matches_scipy(smi, shape, real_t):
    from sopht_mpi_amd.numeric.eulerian_grid_ops import FFTMPI3D

    fft = FFTMPI3D(*shape, mpi_construct=_construct3(shape, real_t), real_t=real_t)
    assert fft.field_buffer.shape == shape and fft.field_buffer.dtype == np.dtype(real_t)
    assert fft.fourier_field_buffer.shape == shape[:2] + (shape[2] // 2 + 1,)
    field = np.random.default_rng(3).random(shape).astype(real_t)
    fft.field_buffer[...] = field
    fft.forward(field=fft.field_buffer, fourier_field=fft.fourier_field_buffer)
    want = scipy.fft.rfftn(field.astype(np.float64))
    got = fft.fourier_field_buffer.get()
    assert got.dtype == fft.complex_dtype
    assert_close(got.view(real_t), want.astype(fft.complex_dtype).view(real_t), real_t, factor=1.0, what="rfftn")
    # backward is normalised and must not disturb its input
    spectrum = want.astype(fft.complex_dtype)
    fft.fourier_field_buffer[...] = spectrum
    inv = fft.ctx.zeros(shape, real_t)
    fft.backward(fourier_field=fft.fourier_field_buffer, inv_fourier_field=inv)
    assert_close(inv.get(), field, real_t, factor=2.0, what="irfftn")
    assert np.array_equal(fft.fourier_field_buffer.get(), spectrum), "backward overwrote its input"


@pytest.mark.parametrize("real_t", PRECISIONS)
@pytest.mark.parametrize("shape", [(8, 8), (12, 16), (10, 6)])
def test_fft_mpi_2d_matches_scipy(smi, shape, real_t):
    from sopht_mpi_amd.numeric.eulerian_grid_ops import FFTMPI2D

    fft = FFTMPI2D(*shape, mpi_construct=_construct2(shape, real_t), real_t=real_t)
    field = np.random.default_rng(4).random(shape).astype(real_t)
    fft.field_buffer[...] = field
    fft.forward(field=fft.field_buffer, fourier_field=fft.fourier_field_buffer)
    want = scipy.fft.rfftn(field.astype(np.float64)).astype(fft.complex_dtype)
    assert_close(fft.fourier_field_buffer.get().view(real_t), want.view(real_t), real_t, factor=1.0, what="rfft2")
    inv = fft.ctx.zeros(shape, real_t)
    fft.backward(fourier_field=fft.fourier_field_buffer, inv_fourier_field=inv)
    assert_close(inv.get(), field, real_t, factor=2.0, what="irfft2")


def test_fft_api_errors(smi):
    from sopht_mpi_amd.numeric.eulerian_grid_ops import FFTMPI3D

    c = _construct3((8, 8, 8), np.float32)
    fft = FFTMPI3D(8, 8, 8, mpi_construct=c, real_t=np.float32)
    with pytest.raises(ValueError):
        fft.forward(field=fft.fourier_field_buffer, fourier_field=fft.fourier_field_buffer)
    with pytest.raises(ValueError):
        FFTMPI3D(8, 8, 8, mpi_construct=c, real_t=np.float64)  # precision of the construct differs


@pytest.mark.parametrize("real_t", PRECISIONS)
@pytest.mark.parametrize("g", [1, 2])
def test_domain_doubling_3d(smi, g, real_t):
    from sopht_mpi_amd.numeric.eulerian_grid_ops import MPIDomainDoublingCommunicator3D

    shape = (6, 8, 12)
    c = _construct3(shape, real_t)
    comm = MPIDomainDoublingCommunicator3D(ghost_size=g, mpi_construct=c)
    ctx = comm.ctx
    rng = np.random.default_rng(5)
    local = rand_field(rng, ctx.padded_shape, real_t)
    doubled0 = rand_field(rng, tuple(2 * n for n in shape), real_t)
    dl, dd = ctx.from_numpy(local), ctx.from_numpy(doubled0)
    comm.copy_to_doubled_domain(local_field=dl, local_doubled_field=dd)
    want = doubled0.copy()
    want[: shape[0], : shape[1], : shape[2]] = local[g:-g, g:-g, g:-g]
    assert np.array_equal(dd.get(), want)
    back0 = rand_field(rng, ctx.padded_shape, real_t)
    db = ctx.from_numpy(back0)
    comm.copy_from_doubled_domain(local_doubled_field=dd, local_field=db)
    want_back = back0.copy()
    want_back[g:-g, g:-g, g:-g] = want[: shape[0], : shape[1], : shape[2]]
    assert np.array_equal(db.get(), want_back)
    with pytest.raises(ValueError):
        comm.copy_to_doubled_domain(local_field=dd, local_doubled_field=dl)


@pytest.mark.parametrize("real_t", PRECISIONS)
def test_domain_doubling_2d(smi, real_t):
    from sopht_mpi_amd.numeric.eulerian_grid_ops import MPIDomainDoublingCommunicator2D

    shape, g = (10, 12), 2
    c = _construct2(shape, real_t)
    comm = MPIDomainDoublingCommunicator2D(ghost_size=g, mpi_construct=c)
    ctx = comm.ctx
    rng = np.random.default_rng(6)
    local = rand_field(rng, tuple(n + 2 * g for n in shape), real_t)
    doubled0 = rand_field(rng, tuple(2 * n for n in shape), real_t)
    dl, dd = ctx.from_numpy(local), ctx.from_numpy(doubled0)
    comm.copy_to_doubled_domain(dl, dd)
    want = doubled0.copy()
    want[: shape[0], : shape[1]] = local[g:-g, g:-g]
    assert np.array_equal(dd.get(), want)
    back0 = rand_field(rng, local.shape, real_t)
    db = ctx.from_numpy(back0)
    comm.copy_from_doubled_domain(dd, db)
    want_back = back0.copy()
    want_back[g:-g, g:-g] = want[: shape[0], : shape[1]]
    assert np.array_equal(db.get(), want_back)


@pytest.mark.parametrize("real_t", PRECISIONS)
def test_reference_solve_sequence_from_pieces(smi, real_t):
    """zero the doubled buffer, copy in, forward, multiply by the Green's spectrum times dx^3,
    backward, copy out (UnboundedPoissonSolverMPI3D.py:141-167) gives the solver's own result."""
    from sopht_mpi_amd.numeric.eulerian_grid_ops import (FFTMPI3D, MPIDomainDoublingCommunicator3D,
                                                         UnboundedPoissonSolverMPI3D)

    shape, g, x_range = (8, 12, 16), 2, 1.3
    c = _construct3(shape, real_t)
    comm = MPIDomainDoublingCommunicator3D(ghost_size=g, mpi_construct=c)
    ctx = comm.ctx
    dshape = tuple(2 * n for n in shape)
    fft = FFTMPI3D(*dshape, mpi_construct=c, real_t=real_t)
    oracle = UnboundedPoissonSolver3D(*shape, x_range=x_range, real_t=real_t)
    rng = np.random.default_rng(8)
    rhs = rand_field(rng, ctx.padded_shape, real_t)
    drhs = ctx.from_numpy(rhs)
    fft.field_buffer[...] = 0
    comm.copy_to_doubled_domain(local_field=drhs, local_doubled_field=fft.field_buffer)
    fft.forward(field=fft.field_buffer, fourier_field=fft.fourier_field_buffer)
    ghat = np.asarray(oracle.fourier_greens_function_times_dx_cubed, dtype=fft.complex_dtype)
    fft.fourier_field_buffer[...] = fft.fourier_field_buffer.get() * ghat
    fft.backward(fourier_field=fft.fourier_field_buffer, inv_fourier_field=fft.field_buffer)
    dsol = ctx.field()
    comm.copy_from_doubled_domain(local_doubled_field=fft.field_buffer, local_field=dsol)
    want = np.zeros_like(rhs)
    oracle.solve(want, rhs, g)
    assert_close(dsol.get(), want, real_t, factor=2.0, what="solve from pieces vs oracle")
    solver = UnboundedPoissonSolverMPI3D(*shape, mpi_construct=c, ghost_size=g, x_range=x_range, real_t=real_t)
    dsol2 = ctx.field()
    solver.solve(solution_field=dsol2, rhs_field=drhs)
    assert_close(dsol.get(), dsol2.get(), real_t, factor=2.0, what="solve from pieces vs library solver")


@pytest.mark.parametrize("real_t", PRECISIONS)
@pytest.mark.parametrize("shape", [(8, 8, 8), (8, 12, 16)])
def test_advection_flux_eno3_3d(smi, shape, real_t):
    from sopht_mpi_amd.numeric.eulerian_grid_ops import gen_advection_flux_conservative_eno3_pyst_mpi_kernel_3d
    from sopht_mpi_amd.utils import MPIGhostCommunicator3D

    g = 2
    c = _construct3(shape, real_t)
    ghost = MPIGhostCommunicator3D(ghost_size=g, mpi_construct=c)
    op = gen_advection_flux_conservative_eno3_pyst_mpi_kernel_3d(real_t, c, ghost)
    assert op.kernel_support == 2 and gen_advection_flux_conservative_eno3_pyst_mpi_kernel_3d.kernel_support == 2
    ctx = ghost.ctx
    rng = np.random.default_rng(9)
    field = rand_field(rng, ctx.padded_shape, real_t)
    vel = rand_field(rng, (3,) + ctx.padded_shape, real_t)
    flux0 = rand_field(rng, ctx.padded_shape, real_t)  # cells outside the boxes must survive
    dflux = ctx.from_numpy(flux0)
    inv_dx = real_t(-0.37)
    op(advection_flux=dflux, field=ctx.from_numpy(field), velocity=ctx.from_numpy(vel), inv_dx=inv_dx)
    want = flux0.copy()
    o3.advection_flux_eno3(want, field, vel, inv_dx, g)
    assert_close(dflux.get(), want, real_t, factor=1.0, what="advection flux 3d")
    untouched = want == flux0
    assert np.array_equal(dflux.get()[untouched], flux0[untouched])
    with pytest.raises(ValueError):
        gen_advection_flux_conservative_eno3_pyst_mpi_kernel_3d(
            real_t, c, MPIGhostCommunicator3D(ghost_size=1, mpi_construct=c))


@pytest.mark.parametrize("real_t", PRECISIONS)
def test_advection_flux_eno3_2d(smi, real_t):
    from sopht_mpi_amd.numeric.eulerian_grid_ops import (
        gen_advection_flux_conservative_eno3_pyst_mpi_kernel_2d,
        gen_advection_timestep_euler_forward_conservative_eno3_pyst_mpi_kernel_2d)
    from sopht_mpi_amd.utils import MPIGhostCommunicator2D

    shape, g = (12, 16), 2
    c = _construct2(shape, real_t)
    ghost = MPIGhostCommunicator2D(ghost_size=g, mpi_construct=c)
    flux_op = gen_advection_flux_conservative_eno3_pyst_mpi_kernel_2d(real_t, c, ghost)
    step_op = gen_advection_timestep_euler_forward_conservative_eno3_pyst_mpi_kernel_2d(real_t, c, ghost)
    ctx = ghost.ctx
    pshape = tuple(n + 2 * g for n in shape)
    rng = np.random.default_rng(10)
    field = rand_field(rng, pshape, real_t)
    vel = rand_field(rng, (2,) + pshape, real_t)
    dt_by_dx = real_t(0.21)
    # the timestep is "flux = 0; flux = op(-dt/dx); field += flux" (advection_timestep_mpi_2d.py)
    dfield, dvel = ctx.from_numpy(field), ctx.from_numpy(vel)
    dflux = ctx.zeros(pshape, real_t)
    flux_op(advection_flux=dflux, field=dfield, velocity=dvel, inv_dx=-dt_by_dx)
    want_field, want_flux = field.copy(), np.zeros(pshape, dtype=real_t)
    o2.advection_timestep_eno3(want_field, want_flux, vel, dt_by_dx, g)
    assert_close(dflux.get(), want_flux, real_t, factor=1.0, what="advection flux 2d")
    # cells outside the update set keep what they held
    sentinel = rand_field(rng, pshape, real_t)
    dflux2 = ctx.from_numpy(sentinel)
    flux_op(advection_flux=dflux2, field=dfield, velocity=dvel, inv_dx=-dt_by_dx)
    got2 = dflux2.get()
    outside = want_flux == 0
    assert np.array_equal(got2[outside & (got2 == sentinel)], sentinel[outside & (got2 == sentinel)])
    assert np.count_nonzero(got2 != sentinel) > 0
    # and the timestep operator agrees with field + flux
    scratch = ctx.zeros(pshape, real_t)
    step_op(field=dfield, advection_flux=scratch, velocity=dvel, dt_by_dx=dt_by_dx)
    assert_close(dfield.get(), want_field, real_t, factor=1.0, what="advection timestep 2d")


@pytest.mark.parametrize("nranks,grid,precision", [(2, "16x16x24", "single"), (4, "32x16x16", "single"),
                                                   (4, "16x32x24", "double"), (8, "32x32x16", "single")])
def test_fft_and_doubled_domain_types_across_slabs(nranks, grid, precision):
    """FFTMPI3D (mpi4py-fft's slab layouts: real array divided along z, spectrum along y) and
    MPIDomainDoublingCommunicator3D across 2, 4 and 8 z-slabs sharing the GPU over the host-staged transport:
    against scipy.fft.rfftn of the global array, and the reference's solve sequence assembled from the two types
    against the oracle (tests/_fft_api_ranks_worker.py)."""
    import os
    import subprocess
    import sys

    from test_gpu_slabs import _failure_digest

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env["SOPHT_MI_TRANSPORT"] = "host"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
           "--master-addr", "127.0.0.1", "--master-port", str(29650 + nranks),
           os.path.join(root, "tests", "_fft_api_ranks_worker.py"), grid, precision]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, _failure_digest(out)
    for r in range(nranks):
        assert f"rank {r} ok" in out.stdout
