"""Analytic known-answer tests that pin the conventions the reference keeps in its un-vendored dependency
`sopht` (7-point diffusion flux, central curl and its orientation, `w += p curl(f)`, cross product, the
divergence prefactor) and the chain Poisson solve -> curl of `compute_flow_velocity`
(flow_simulators_mpi_3d.py:382-393), on closed-form fields.  Each test runs twice: on the CPU oracle
(`-m "not gpu"`) and on the HIP library through its C ABI / simulator API (`-m gpu`), with the same
assertions, so that the oracle and the product are tied to the same external answers rather than to each
other.  The 2D case is the reference's own Lamb-Oseen example (256^2, double precision,
examples/2d_examples/LambOseenVortexCase/lamb_oseen_vortex.py:21-180) with asserted error levels.
"""
import numpy as np
import pytest

import analytic_cases as ac
from helpers import make_ctx
from oracle import stencils_3d as st

BACKENDS = ["oracle", pytest.param("gpu", marks=pytest.mark.gpu)]


class _OracleOps:
    def __init__(self, shape, g, real_t):
        self.shape, self.g, self.real_t = shape, g, real_t

    def cross(self, a, b):
        out = np.zeros_like(a)
        st.elementwise_cross_product(out, a, b)
        return out

    def update_vorticity(self, w, f, p):
        out = w.copy()
        st.update_vorticity_from_velocity_forcing(out, f, self.real_t(p), self.g)
        return out

    def diffusion_flux(self, f, c):
        out = np.zeros_like(f)
        for k in range(3):
            st.diffusion_flux(out[k], f[k], self.real_t(c), self.g)
        return out

    def curl(self, f, p):
        out = np.zeros_like(f)
        st.curl(out, f, self.real_t(p), self.g)
        return out

    def divergence(self, f, inv_dx):
        out = np.zeros(f.shape[1:], dtype=f.dtype)
        st.divergence(out, f, self.real_t(inv_dx), self.g)
        return out


class _GpuOps:
    def __init__(self, shape, g, real_t):
        import sopht_mpi_amd

        self.smi = sopht_mpi_amd
        self.ctx = make_ctx(sopht_mpi_amd, shape, g, real_t)
        self.real_t = real_t

    def _call(self, name, *fields_and_scalars):
        # the DeviceField arguments stay referenced until the call has returned and the result was downloaded
        args = [a.ptr if hasattr(a, "ptr") else a for a in fields_and_scalars]
        self.smi._lib.call(name, self.ctx.handle, *args)

    def cross(self, a, b):
        out, da, db = self.ctx.field(3), self.ctx.from_numpy(a), self.ctx.from_numpy(b)
        self._call("smi_cross_product", out, da, db)
        return out.get()

    def update_vorticity(self, w, f, p):
        dw, df = self.ctx.from_numpy(w), self.ctx.from_numpy(f)
        self._call("smi_update_vorticity_from_velocity_forcing", dw, df, float(self.real_t(p)))
        return dw.get()

    def diffusion_flux(self, f, c):
        out, df = self.ctx.field(3), self.ctx.from_numpy(f)
        self._call("smi_diffusion_flux", out, df, float(self.real_t(c)), 3)
        return out.get()

    def curl(self, f, p):
        out, df = self.ctx.field(3), self.ctx.from_numpy(f)
        self._call("smi_curl", out, df, float(self.real_t(p)))
        return out.get()

    def divergence(self, f, inv_dx):
        out, df = self.ctx.field(), self.ctx.from_numpy(f)
        self._call("smi_divergence", out, df, float(self.real_t(inv_dx)))
        return out.get()


def _ops(backend, shape, g, real_t):
    return (_OracleOps if backend == "oracle" else _GpuOps)(shape, g, real_t)


def _sim3d(backend, **kw):
    if backend == "oracle":
        from oracle.flow_3d import FlowSimulator3D

        return FlowSimulator3D(**kw)
    from sopht_mpi_amd.simulator import UnboundedFlowSimulator3D

    return UnboundedFlowSimulator3D(**kw)


def _rel_err(got, want, region):
    return float(np.max(np.abs(np.asarray(got, dtype=np.float64)[region] - want[region])) / np.max(np.abs(want[region])))


# ------------------------------------------------------------------ operators on a trigonometric field
def _operator_errors(backend, n, real_t, g=2):
    shape = (n, n + 8, n + 16) if n <= 32 else (n, n, n)
    dx = 1.0 / n
    x, y, z = ac.padded_coordinates(shape, g, dx)
    f = ac.trig_field(x, y, z)
    ops = _ops(backend, shape, g, real_t)
    fr = f.astype(real_t)
    far = (slice(None),) + (slice(g + 1, -(g + 1)),) * 3   # flux-type operators zero g + 1 layers at the boundary
    errs = {}
    errs["curl"] = _rel_err(ops.curl(fr, 0.5 / dx), ac.trig_curl(x, y, z), far)
    w0 = ac.trig_field(z, x, y)  # any smooth start value
    errs["update"] = _rel_err(ops.update_vorticity(w0.astype(real_t), fr, 0.37 / (2 * dx)) - w0.astype(real_t),
                              0.37 * ac.trig_curl(x, y, z), far)
    errs["laplacian"] = _rel_err(ops.diffusion_flux(fr, 1.0 / dx**2), ac.trig_laplacian(x, y, z), far)
    errs["divergence"] = _rel_err(ops.divergence(fr, 1.0 / dx), ac.trig_divergence(x, y, z), far[1:])
    return errs


@pytest.mark.parametrize("backend", BACKENDS)
def test_stencil_operators_converge_to_analytic_derivatives(backend):
    """curl (curl_mpi_3d.py:10-196, prefactor 0.5 / dx as in flow_simulators_mpi_3d.py:388-392),
    w += p curl(f) (update_vorticity_from_velocity_forcing_mpi_3d.py:30-36), the 7-point flux
    (diffusion_flux_mpi_3d.py:10-228) with prefactor 1 / dx^2 and the divergence with inv_dx = 1 / dx
    (flow_simulators_mpi_3d.py:454-458) against the analytic curl, Laplacian and divergence of a
    trigonometric vector field: second-order convergence in double precision."""
    coarse = _operator_errors(backend, 32, np.float64)
    fine = _operator_errors(backend, 64, np.float64)
    for name in coarse:
        # leading error of a central difference at wave number k: (k dx)^2 / 6 (first derivative), / 12 (second)
        assert fine[name] < 2.5e-2, (name, fine)
        ratio = coarse[name] / fine[name]
        assert 3.6 < ratio < 4.4, (name, coarse[name], fine[name])


@pytest.mark.parametrize("backend", BACKENDS)
def test_stencil_operators_single_precision_levels(backend):
    errs = _operator_errors(backend, 64, np.float32)
    ref = _operator_errors("oracle", 64, np.float64)
    for name in errs:
        # the discretisation error dominates; single precision adds its rounding amplified by 1 / dx (or 1 / dx^2)
        assert abs(errs[name] - ref[name]) < (2e-3 if name == "laplacian" else 1e-4), (name, errs[name], ref[name])


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("real_t", [np.float32, np.float64])
def test_cross_product_is_right_handed(backend, real_t):
    """gen_elementwise_cross_product_pyst_kernel_3d (call site flow_simulators_mpi_3d.py:396-401):
    result = field_1 x field_2 with components (x, y, z) = (0, 1, 2); numpy's cross is the definition."""
    shape, g = (8, 12, 16), 2
    rng = np.random.default_rng(21)
    a = rng.standard_normal((3,) + tuple(n + 2 * g for n in shape)).astype(real_t)
    b = rng.standard_normal(a.shape).astype(real_t)
    got = _ops(backend, shape, g, real_t).cross(a, b)
    want = np.cross(a.astype(np.float64), b.astype(np.float64), axis=0)
    assert np.max(np.abs(got - want)) <= 4 * np.finfo(real_t).eps * np.max(np.abs(want))
    # unit vectors: x cross y = z
    ex, ey = np.zeros_like(a), np.zeros_like(a)
    ex[0], ey[1] = 1, 1
    ez = _ops(backend, shape, g, real_t).cross(ex, ey)
    assert np.all(ez[2] == 1) and not ez[:2].any()


# ------------------------------------------------------------------ Poisson solve + curl: compute_flow_velocity
def _blob_velocity_error(backend, n, real_t):
    sim = _sim3d(backend, grid_size=(n, n, n), x_range=1.0, kinematic_viscosity=1e-3, flow_type="navier_stokes",
                 real_t=real_t)
    g = sim.ghost_size
    x, y, z = ac.padded_coordinates((n, n, n), g, 1.0 / n)
    w, psi, u = ac.blob_fields(x, y, z)
    sim.vorticity_field[...] = w.astype(real_t)
    sim.compute_flow_velocity(free_stream_velocity=np.zeros(3))
    inner = (slice(None),) + (slice(g + 1, -(g + 1)),) * 3
    return (_rel_err(np.asarray(sim.velocity_field), u, inner), _rel_err(np.asarray(sim.stream_func_field), psi, inner),
            np.asarray(sim.velocity_field), u)


@pytest.mark.parametrize("backend", BACKENDS)
def test_compute_flow_velocity_recovers_analytic_blob(backend):
    """Vorticity of the divergence-free Gaussian stream function psi = grad(phi) x A (all three components
    non-zero, centre off the grid's symmetry planes) -> compute_flow_velocity -> the analytic psi and
    u = curl psi.  Pins the sign of the Poisson problem (-lap psi = w, UnboundedPoissonSolverMPI3D.py:82-167),
    the orientation and component order of the curl and its 0.5 / dx prefactor; second-order convergence."""
    real_t = np.float64 if backend == "oracle" else np.float32
    eu_c, ep_c, _, _ = _blob_velocity_error(backend, 48 if backend == "oracle" else 64, real_t)
    eu_f, ep_f, got, want = _blob_velocity_error(backend, 96 if backend == "oracle" else 128, real_t)
    # measured: 2.7 % (velocity) and 1.3 % (stream function) at 96^3, 1.5 % and 0.76 % at 128^3
    bound_u, bound_psi = (3.3e-2, 1.7e-2) if backend == "oracle" else (1.9e-2, 0.95e-2)
    assert eu_f < bound_u and ep_f < bound_psi, (eu_f, ep_f)
    assert 3.3 < eu_c / eu_f < 4.7, (eu_c, eu_f)
    assert 3.3 < ep_c / ep_f < 4.7, (ep_c, ep_f)
    # every component on its own (a permutation of components would pass a norm over all three only by accident)
    for c in range(3):
        assert np.max(np.abs(got[c] - want[c])) < bound_u * np.max(np.abs(want[c])), c


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("axis", [0, 1, 2])
def test_vortex_tube_along_each_axis_gives_lamb_oseen_velocity(backend, axis):
    """A straight Gaussian vortex tube along x, y, z in turn (grid four times longer along the tube) ->
    compute_flow_velocity -> the Lamb-Oseen azimuthal velocity of
    examples/2d_examples/LambOseenVortexCase/lamb_oseen_helpers.py:4-50, right-handed about the tube axis,
    times the finite-length factor of a straight vortex segment; compared on the three planes around the middle of
    the tube, between 1.5 core radii and 0.3 from the axis."""
    real_t = np.float64 if backend == "oracle" else np.float32
    n, g = 64, 2
    shape = [n, n, n]
    shape[2 - axis] = 4 * n   # array axes are (z, y, x)
    sim = _sim3d(backend, grid_size=tuple(shape), x_range=float(shape[2]) / n, kinematic_viscosity=1e-3,
                 flow_type="navier_stokes", real_t=real_t, penalty_zone_width=2)
    dx = 1.0 / n
    x, y, z = ac.padded_coordinates(tuple(shape), g, dx)
    centre = [0.5, 0.5, 0.5]
    centre[axis] = 2.0
    centre[(axis + 1) % 3] = 0.47   # off the symmetry plane
    four_nu_t, gamma = 0.08**2, 1.3
    w, u = ac.lamb_oseen_tube(x, y, z, axis, centre, four_nu_t, gamma, half_length=2.0)
    sim.vorticity_field[...] = w.astype(real_t)
    sim.compute_flow_velocity(free_stream_velocity=np.zeros(3))
    got = np.asarray(sim.velocity_field).astype(np.float64)
    r = [x, y, z]
    a, b = [(1, 2), (2, 0), (0, 1)][axis]
    rho = np.sqrt((r[a] - centre[a]) ** 2 + (r[b] - centre[b]) ** 2)
    region = (np.abs(r[axis] - centre[axis]) < 1.6 * dx) & (rho > 0.12) & (rho < 0.3)
    scale = np.max(np.abs(u[:, region]))
    for c in range(3):
        err = np.max(np.abs(got[c][region] - u[c][region])) / scale
        assert err < 1.5e-2, (axis, c, err)   # measured 0.97 % (second order: 1.6 % on a 48-cell grid)
    # the axial component vanishes, the in-plane ones do not
    assert np.max(np.abs(got[axis][region])) < 1e-3 * scale
    assert np.max(np.abs(got[a][region])) > 0.5 * scale and np.max(np.abs(got[b][region])) > 0.5 * scale


# ------------------------------------------------------------------ diagnostics, closed forms
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("real_t", [np.float32, np.float64])
def test_divergence_norm_closed_form(backend, real_t):
    """get_vorticity_divergence_l2_norm (flow_simulators_mpi_3d.py:451-469, formula restated by
    tests/test_simulator/test_flow/test_flow_simulators_3d.py:363-438) on w = (a x^2, b y^2, c z^2), whose central
    difference divergence is exact: 2 (a x + b y + c z); norm = sqrt(sum div^2) dx^1.5 over the interior."""
    n = 16
    sim = _sim3d(backend, grid_size=(n, n, n), x_range=1.0, kinematic_viscosity=1e-2,
                 flow_type="navier_stokes_with_forcing", real_t=real_t)
    g = sim.ghost_size
    dx = 1.0 / n
    x, y, z = ac.padded_coordinates((n, n, n), g, dx)
    coef = (0.7, -1.1, 0.4)
    sim.vorticity_field[...] = np.stack([coef[0] * x**2, coef[1] * y**2, coef[2] * z**2]).astype(real_t)
    div = 2 * (coef[0] * x + coef[1] * y + coef[2] * z)
    # the divergence wrapper zeroes g + 1 layers next to the physical boundaries (divergence_mpi_3d.py:167-198),
    # the norm runs over the interior, so one interior layer per face contributes zero
    core = (slice(g + 1, -(g + 1)),) * 3
    want = np.sqrt(np.sum(div[core] ** 2)) * dx**1.5
    got = float(sim.get_vorticity_divergence_l2_norm())
    assert abs(got - want) <= (2e-5 if real_t == np.float32 else 1e-12) * want


# ------------------------------------------------------------------ 2D: the reference's Lamb-Oseen case
# Error levels of the oracle at 256^2 in double precision (vortex carried by the free stream (1, 1) from
# t = 1.0 to 1.4 while it diffuses, 2059 steps): vorticity L2 4.7e-3, Linf 4.5e-2, velocity L2 5.6e-4, Linf 6.7e-3,
# peak vorticity 0.7336 against 1 / 1.4 = 0.7143.  The bounds are those levels with a margin of 1.5.
LAMB_OSEEN_256 = {"vorticity_l2": 7.0e-3, "vorticity_linf": 6.7e-2, "velocity_l2": 8.4e-4, "velocity_linf": 1.0e-2}


@pytest.mark.parametrize("backend", BACKENDS)
def test_lamb_oseen_256_double(backend):
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import lamb_oseen_2d as lo

    res = lo.run(256, "double", t_end=1.4, device="cpu" if backend == "oracle" else "gpu")
    print(res)
    for key, bound in LAMB_OSEEN_256.items():
        assert res[key] < bound, (key, res[key], bound)
    assert abs(res["max_vorticity"] - 1 / 1.4) < 3e-2
