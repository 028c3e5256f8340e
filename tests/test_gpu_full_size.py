"""GPU, at the benchmark's full size (512^3, f32): properties that do not need the CPU oracle
(which would take minutes at this size).

  * the two independent Poisson back ends (padded rocFFT vs pruned LDS transforms) agree;
  * the solve is linear;
  * the potential of a Gaussian blob matches its closed form;
  * u = curl(psi) has zero discrete divergence;
  * one fused step equals the operator-by-operator sequence of the same library.
"""
import math

import numpy as np
import pytest

from helpers import assert_close

pytestmark = pytest.mark.gpu

N = 512
G = 2


def _blobs(pos, real_t, seed, sigma=0.05, nblobs=6):
    rng = np.random.default_rng(seed)
    x, y, z = (pos[0][0, 0, :].astype(np.float64), pos[1][0, :, 0].astype(np.float64),
               pos[2][:, 0, 0].astype(np.float64))
    out = np.zeros((3,) + pos.shape[1:], dtype=real_t)
    for _ in range(nblobs):
        c = rng.uniform(0.3, 0.7, 3)
        amp = rng.uniform(-1, 1, 3)
        blob = (np.exp(-((z - c[2]) ** 2) / (2 * sigma**2))[:, None, None]
                * np.exp(-((y - c[1]) ** 2) / (2 * sigma**2))[None, :, None]
                * np.exp(-((x - c[0]) ** 2) / (2 * sigma**2))[None, None, :]).astype(real_t)
        for k in range(3):
            out[k] += real_t(amp[k]) * blob
    return out


@pytest.fixture(scope="module")
def sim(smi):
    from sopht_mpi_amd.simulator import UnboundedFlowSimulator3D

    return UnboundedFlowSimulator3D(grid_size=(N, N, N), x_range=1.0, kinematic_viscosity=1e-3,
                                    flow_type="navier_stokes", real_t=np.float32, ghost_size=G,
                                    poisson_backend="pruned")


def test_poisson_back_ends_agree_and_solve_is_linear(smi, sim):
    from sopht_mpi_amd.numeric.eulerian_grid_ops import UnboundedPoissonSolverMPI3D

    ctx = sim.ctx
    real_t = np.float32
    rocfft = UnboundedPoissonSolverMPI3D(N, N, N, sim.mpi_construct, G, x_range=1.0, real_t=real_t, backend="rocfft")
    pruned = sim.unbounded_poisson_solver
    assert pruned.backend == "pruned" and rocfft.backend == "rocfft"
    f1 = _blobs(sim.position_field, real_t, seed=1)
    f2 = _blobs(sim.position_field, real_t, seed=2)
    d1, d2 = ctx.from_numpy(f1), ctx.from_numpy(f2)
    s1, s2, s3 = ctx.field(3), ctx.field(3), ctx.field(3)
    pruned.vector_field_solve(s1, d1)
    rocfft.vector_field_solve(s2, d1)
    a = s1.get()
    inner = (slice(None),) + (slice(G, -G),) * 3
    assert_close(a[inner], s2.get()[inner], real_t, factor=3.0, what="pruned vs rocFFT at 512^3")
    # linearity: solve(2 f1 - 3 f2) = 2 solve(f1) - 3 solve(f2)
    pruned.vector_field_solve(s2, d2)
    b = s2.get()
    smi._lib.call("smi_elementwise_saxpby", ctx.handle, d1.ptr, d1.ptr, 2.0, d2.ptr, -3.0, 3)
    pruned.vector_field_solve(s3, d1)
    assert_close(s3.get()[inner], (2 * a - 3 * b)[inner], real_t, factor=2.0, what="linearity")


def test_poisson_gaussian_blob_closed_form(smi, sim):
    """-lap(psi) = w with w a Gaussian of width s: psi(r) = s^2 A sqrt(pi/2) s erf(r / (sqrt 2 s)) / r."""
    ctx = sim.ctx
    real_t = np.float32
    pos = sim.position_field
    s, c = 0.04, (0.5, 0.5, 0.5)
    x, y, z = pos[0][0, 0, :].astype(np.float64), pos[1][0, :, 0].astype(np.float64), pos[2][:, 0, 0].astype(np.float64)
    r2 = ((z - c[2]) ** 2)[:, None, None] + ((y - c[1]) ** 2)[None, :, None] + ((x - c[0]) ** 2)[None, None, :]
    w = np.exp(-r2 / (2 * s * s))
    rhs = np.zeros((3,) + w.shape, dtype=real_t)
    rhs[0] = w
    sol = ctx.field(3)
    sim.unbounded_poisson_solver.vector_field_solve(sol, ctx.from_numpy(rhs))
    r = np.sqrt(np.maximum(r2, 1e-30))
    erf = np.vectorize(math.erf)
    line = np.sqrt(np.maximum(r2[N // 2 + G, N // 2 + G, :], 1e-30))  # one grid line through the blob
    want = s**3 * math.sqrt(math.pi / 2) * erf(line / (math.sqrt(2) * s)) / line
    got = sol.get()[0][N // 2 + G, N // 2 + G, :]
    del r
    # second-order accurate Green's function convolution on a dx = 1/512 grid: ~1e-4 relative
    err = np.max(np.abs(got[G:-G] - want[G:-G])) / np.max(np.abs(want))
    assert err < 5e-4, err


def test_velocity_is_divergence_free_and_fused_step_matches_sequence(smi, sim):
    from sopht_mpi_amd.simulator import UnboundedFlowSimulator3D

    real_t = np.float32
    w0 = _blobs(sim.position_field, real_t, seed=3)
    sim.vorticity_field[...] = w0
    sim.compute_flow_velocity(free_stream_velocity=None)
    ctx = sim.ctx
    div = ctx.field()
    smi._lib.call("smi_divergence", ctx.handle, div.ptr, sim.velocity_field.ptr, float(1.0 / sim.dx))
    umax = float(np.abs(sim.velocity_field.get()).max())
    k = G + 3  # away from the boundary bands on which the curl is zeroed
    d = div.get()[k:-k, k:-k, k:-k]
    # central divergence of a central curl cancels term by term; what is left is rounding
    assert float(np.abs(d).max()) <= 64 * np.finfo(real_t).eps * umax / float(sim.dx)
    dt = float(sim.compute_stable_timestep())
    seq = UnboundedFlowSimulator3D(grid_size=(N, N, N), x_range=1.0, kinematic_viscosity=1e-3,
                                   flow_type="navier_stokes", real_t=real_t, ghost_size=G,
                                   poisson_backend="pruned", fused_step=False)
    seq.vorticity_field[...] = w0
    seq.compute_flow_velocity(free_stream_velocity=None)
    for _ in range(2):
        sim.time_step(dt=dt)
        seq.time_step(dt=dt)
    inner = (slice(None),) + (slice(G, -G),) * 3
    assert_close(sim.vorticity_field.get()[inner], seq.vorticity_field.get()[inner], real_t, factor=1.0,
                 what="fused vs sequenced vorticity")
    assert_close(sim.velocity_field.get()[inner], seq.velocity_field.get()[inner], real_t, factor=1.0,
                 what="fused vs sequenced velocity")
