"""CPU: pin the oracle with the known answers the reference's own tests and examples hold
(there are no golden vectors in the reference tree)."""
import numpy as np
import pytest
import scipy.fft

from oracle import immersed_boundary as oib
from oracle import stencils_3d as st
from oracle.decomp import ZSlabs
from oracle.flow_3d import FlowSimulator3D, get_test_tol
from oracle.poisson import UnboundedPoissonSolver2D, UnboundedPoissonSolver3D


@pytest.mark.parametrize("real_t", [np.float32, np.float64])
def test_compute_stable_timestep_closed_form(real_t):
    """tests/test_simulator/test_flow/test_flow_simulators_3d.py:333-360."""
    precision = "single" if real_t == np.float32 else "double"
    n, cfl, nu, dt_prefac = 8, 0.2, 1.0, 0.5
    dx = real_t(1.0 / n)
    sim = FlowSimulator3D((n, n, n), 1.0, nu, CFL=cfl, real_t=real_t)
    sim.velocity_field[...] = 2.0
    adv = cfl * dx / (3 * 2.0 + get_test_tol(precision))
    dif = 0.9 * dx**2 / 6 / (nu + get_test_tol(precision))
    assert sim.compute_stable_timestep(dt_prefac=dt_prefac, precision=precision) == dt_prefac * min(adv, dif)


def test_max_vorticity_and_divergence_norm():
    """test_flow_simulators_3d.py:363-500: spike of 10 is the maximum; divergence norm formula."""
    rng = np.random.default_rng(0)
    sim = FlowSimulator3D((8, 8, 8), 1.0, 1e-2, flow_type="navier_stokes_with_forcing", real_t=np.float64)
    w = rng.random(sim.vorticity_field.shape)
    w[0, 6, 6, 6] = 10.0
    sim.vorticity_field[...] = w
    assert sim.get_max_vorticity() == 10.0
    div = np.zeros(w.shape[1:])
    st.divergence(div, w, 1.0 / sim.dx, 2)
    assert sim.get_vorticity_divergence_l2_norm() == pytest.approx(
        np.sqrt(np.linalg.norm(div[2:-2, 2:-2, 2:-2]) ** 2) * sim.dx**1.5)
    # a curl field is divergence free under the same central differences (interior of the interior)
    psi = rng.random(w.shape)
    u = np.zeros_like(psi)
    st.curl(u, psi, 0.5 / sim.dx, 2)
    st.divergence(div, u, 1.0 / sim.dx, 2)
    assert np.abs(div[4:-4, 4:-4, 4:-4]).max() < 1e-12


def test_fft_convention_is_scipy_rfftn():
    """tests/.../test_fft_mpi_3d.py:105-115: forward unnormalised, backward normalised; the Poisson
    restatement therefore reduces to plain rfftn / irfftn calls."""
    p = UnboundedPoissonSolver3D(8, 8, 8, 1.0, np.float64)
    g = p.greens_function_field()
    np.testing.assert_allclose(p.fourier_greens_function_times_dx_cubed, scipy.fft.rfftn(g) * p.dx**3)
    back = scipy.fft.irfftn(scipy.fft.rfftn(g), s=g.shape)
    np.testing.assert_allclose(back, g, atol=1e-12)
    assert g[0, 0, 0] == pytest.approx(1 / (4 * np.pi * p.dx))
    # even symmetry of the doubled Green's function -> real spectrum
    assert np.abs(p.fourier_greens_function_times_dx_cubed.imag).max() < 1e-15


def test_poisson_3d_gaussian_analytic():
    """Unbounded solution of -lap(psi) = exp(-r^2 / 2 s^2): psi = s^3 sqrt(pi/2) erf(r / (sqrt2 s)) / r."""
    from scipy.special import erf

    n, g, s = 64, 2, 0.06
    p = UnboundedPoissonSolver3D(n, n, n, 1.0, np.float64)
    x = (np.arange(n) + 0.5) / n
    z, y, xx = np.meshgrid(x, x, x, indexing="ij")
    r = np.sqrt((xx - 0.5) ** 2 + (y - 0.5) ** 2 + (z - 0.5) ** 2)
    rhs = np.zeros((n + 2 * g,) * 3)
    rhs[g:-g, g:-g, g:-g] = np.exp(-(r**2) / (2 * s**2))
    sol = np.zeros_like(rhs)
    p.solve(sol, rhs, g)
    ana = s**3 * np.sqrt(np.pi / 2) * erf(r / (np.sqrt(2) * s)) / r
    assert np.abs(sol[g:-g, g:-g, g:-g] - ana).max() < 2e-2 * ana.max()


def test_poisson_2d_lamb_oseen_velocity():
    """examples/2d_examples/LambOseenVortexCase: stream function of a Lamb-Oseen vortex recovers
    its analytic azimuthal velocity (lamb_oseen_helpers.py:4-50) through the 2D solver + curl."""
    n, g, nu, t0 = 256, 2, 1e-3, 1.0
    gamma = 4 * np.pi * nu * t0
    p = UnboundedPoissonSolver2D(n, n, 1.0, np.float64)
    dx = p.dx
    x = (np.arange(n) + 0.5) * dx
    yy, xx = np.meshgrid(x, x, indexing="ij")
    r2 = (xx - 0.5) ** 2 + (yy - 0.5) ** 2
    omega = np.zeros((n + 2 * g,) * 2)
    omega[g:-g, g:-g] = gamma / (4 * np.pi * nu * t0) * np.exp(-r2 / (4 * nu * t0))
    psi = np.zeros_like(omega)
    p.solve(psi, omega, g)
    u = (psi[g + 1 : -g + 1, g:-g] - psi[g - 1 : -g - 1, g:-g]) / (2 * dx)      # u = d psi / dy
    v = -(psi[g:-g, g + 1 : -g + 1] - psi[g:-g, g - 1 : -g - 1]) / (2 * dx)    # v = -d psi / dx
    r = np.sqrt(r2)
    u_theta = gamma / (2 * np.pi * r) * (1 - np.exp(-r2 / (4 * nu * t0)))
    u_ana, v_ana = -u_theta * (yy - 0.5) / r, u_theta * (xx - 0.5) / r
    inner = (slice(4, -4),) * 2
    assert np.abs(u[inner] - u_ana[inner]).max() < 1e-2 * np.abs(u_ana).max()
    assert np.abs(v[inner] - v_ana[inner]).max() < 1e-2 * np.abs(v_ana).max()


@pytest.mark.parametrize("kernel", ["cosine", "peskin"])
def test_delta_kernel_partition_of_unity(kernel):
    """tests/.../test_eulerian_lagrangian_grid_communicator_mpi_3d.py:114-149: E->L interpolation
    of the constant field 2.0 returns 2.0, for nodes anywhere inside a cell."""
    rng = np.random.default_rng(1)
    n, g, m = 16, 2, 50
    dx = np.float32(1.0 / n)
    pos = rng.uniform(0.25, 0.75, (3, m))
    pos[:, 0] = (np.array([5, 6, 7]) + 0.5) * float(dx)
    idx = np.empty((3, m), dtype=np.int64)
    sup = np.empty((3, 4, 4, 4, m))
    w = np.empty((4, 4, 4, m))
    oib.local_eulerian_grid_support_of_lagrangian_grid(sup, idx, pos, dx, np.float32(dx / 2), 2, np.zeros(3) - g)
    (oib.cosine_interpolation_weights if kernel == "cosine" else oib.peskin_interpolation_weights)(
        w, sup, dx, np.float32)
    eul = np.full((3,) + (n + 2 * g,) * 3, 2.0, dtype=np.float32)
    lag = np.zeros((3, m))
    oib.eulerian_to_lagrangian_grid_interpolation(lag, eul, w, idx, dx)
    np.testing.assert_allclose(lag, 2.0, rtol=1e-6)
    # the nearest index is the cell whose centre is at or just below the node
    centre = (idx[0] + g * 0 - 0 + 0.5) * float(dx) - g * float(dx)
    assert np.all(centre <= pos[0] + 1e-12) and np.all(pos[0] < centre + float(dx) + 1e-12)
    # spreading conserves the total force: sum(eul) dx^3 == sum(lag)
    f = rng.standard_normal((3, m))
    field = np.zeros((3,) + (n + 2 * g,) * 3)
    oib.lagrangian_to_eulerian_grid_interpolation(field, f, w, idx)
    np.testing.assert_allclose(field.sum(axis=(1, 2, 3)) * float(dx) ** 3, f.sum(axis=1), rtol=1e-6)


def test_grid_deviation_norm_known_answer():
    """tests/test_simulator/immersed_body/test_immersed_body_interaction_mpi.py:181-188."""
    m = 7
    mismatch = np.full((3, m), 2.0)
    assert np.linalg.norm(mismatch) / np.sqrt(m) == pytest.approx(2 * np.sqrt(3))


def test_penalise_is_closed_form_product():
    """penalise_field_boundary_mpi_3d.py:196-242 applied in x, y, z order equals
    f[clamped index] * fx * fy * fz with first-interior-layer factor sin(0) = 0."""
    from oracle.flow_3d import local_coordinates

    rng = np.random.default_rng(2)
    n, g, w = 12, 2, 2
    dx = np.float64(1.0 / n)
    c = local_coordinates(n, 0, dx, g, np.float64)
    f = rng.random((n + 2 * g,) * 3)
    out = f.copy()
    st.penalise_field_boundary(out, w, dx, c, c, c, g)
    band = g + w
    idx = np.clip(np.arange(n + 2 * g), band - 1, n + 2 * g - band)
    fac = np.ones(n + 2 * g)
    fac[:band] = np.sin(np.pi / 2 / (w * dx) * (c[:band] - c[g]))
    fac[-band:] = np.sin(np.pi / 2 / (w * dx) * (c[-(g + 1)] - c[-band:]))
    want = f[np.ix_(idx, idx, idx)] * fac[:, None, None] * fac[None, :, None] * fac[None, None, :]
    np.testing.assert_allclose(out, want, rtol=1e-13, atol=1e-15)
    assert np.all(out[g, :, :] == 0) and np.all(out[:, :, -(g + 1)] == 0)


@pytest.mark.parametrize("nranks", [2, 4])
def test_slab_operators_match_single_rank(nranks):
    """The reference's test strategy (tests/.../test_diffusion_flux_mpi_3d.py:60-110): an operator
    run per rank on scattered slabs with halo exchange equals the serial operator on the
    gathered array, compared on interiors."""
    rng = np.random.default_rng(3)
    n, g = (16, 8, 12), 2
    dec = ZSlabs(n, nranks, g)
    glob = rng.random((3,) + n)
    single = np.zeros((3,) + tuple(k + 2 * g for k in n))
    single[:, g:-g, g:-g, g:-g] = glob
    locs = dec.scatter(glob)
    dec.exchange_faces(locs)
    # diffusion flux
    want = np.zeros_like(single)
    for k in range(3):
        st.diffusion_flux(want[k], single[k], 0.3, g)
    outs = [np.zeros_like(loc) for loc in locs]
    for r, loc in enumerate(locs):
        for k in range(3):
            st.diffusion_flux(outs[r][k], loc[k], 0.3, g, dec.phys(r))
    np.testing.assert_allclose(dec.gather(outs), want[:, g:-g, g:-g, g:-g], rtol=1e-13)
    # curl
    want = np.zeros_like(single)
    st.curl(want, single, 0.7, g)
    for r, loc in enumerate(locs):
        st.curl(outs[r], loc, 0.7, g, dec.phys(r))
    np.testing.assert_allclose(dec.gather(outs), want[:, g:-g, g:-g, g:-g], rtol=1e-13)
    # vorticity update (compared away from the physical boundary, like test_curl_mpi_3d.py:107-114)
    w_single = np.zeros_like(single)
    st.update_vorticity_from_velocity_forcing(w_single, single, 0.7, g)
    w_locs = [np.zeros_like(loc) for loc in locs]
    for r, loc in enumerate(locs):
        st.update_vorticity_from_velocity_forcing(w_locs[r], loc, 0.7, g)
    got = dec.gather(w_locs)
    np.testing.assert_allclose(got[:, 1:-1, 1:-1, 1:-1], w_single[:, g + 1 : -g - 1, g + 1 : -g - 1, g + 1 : -g - 1],
                               rtol=1e-13)


def test_ghost_sum_slabs_conserves_spread_force():
    """MPIGhostSumCommunicator3D semantics: spreading on slabs + ghost sum equals spreading on one rank."""
    rng = np.random.default_rng(4)
    n, g, m, P = (16, 16, 16), 2, 40, 2
    dx = np.float64(1.0 / 16)
    dec = ZSlabs(n, P, g)
    pos = rng.uniform(0.2, 0.8, (3, m))
    pos[2, :4] = 0.5 + np.array([-0.01, 0.0, 0.01, 0.03])  # nodes straddling the slab boundary z = 0.5
    f = rng.standard_normal((3, m))
    single = np.zeros((3,) + (20,) * 3)
    idx = np.empty((3, m), dtype=np.int64)
    sup = np.empty((3, 4, 4, 4, m))
    w = np.empty((4, 4, 4, m))
    oib.local_eulerian_grid_support_of_lagrangian_grid(sup, idx, pos, dx, dx / 2, 2, np.zeros(3) - g)
    oib.cosine_interpolation_weights(w, sup, dx, np.float64)
    oib.lagrangian_to_eulerian_grid_interpolation(single, f, w, idx)
    oib.clear_ghost_cells(single, g)
    owner = dec.lagrangian_owner(pos, dx, dx / 2)
    locs = [np.zeros((3, 8 + 2 * g, 20, 20)) for _ in range(P)]
    for r in range(P):
        sel = np.where(owner == r)[0]
        mr = len(sel)
        i2 = np.empty((3, mr), dtype=np.int64)
        s2 = np.empty((3, 4, 4, 4, mr))
        w2 = np.empty((4, 4, 4, mr))
        shift = np.array([0, 0, r * 8]) - g
        oib.local_eulerian_grid_support_of_lagrangian_grid(s2, i2, pos[:, sel], dx, dx / 2, 2, shift)
        oib.cosine_interpolation_weights(w2, s2, dx, np.float64)
        oib.lagrangian_to_eulerian_grid_interpolation(locs[r], f[:, sel], w2, i2)
    for k in range(3):
        dec.ghost_sum([loc[k] for loc in locs])
    np.testing.assert_allclose(dec.gather(locs), single[:, g:-g, g:-g, g:-g], rtol=1e-12, atol=1e-14)
