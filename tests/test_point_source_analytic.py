"""Analytic end-to-end check of the passive transport path: a diffusing point source carried by a
uniform velocity (1, 1, 1), the case of
examples/3d_examples/PointSourceAdvectAndDiffuseCase/point_source_advection_diffusion.py:20-171
(nu = 1e-3, source at (0.3, 0.3, 0.3), t = 5.0 -> 5.4, flow_type "passive_vector"), against the heat
kernel of point_source_helpers.py:4-40.  The reference prints these errors without asserting a
threshold; here they must shrink with resolution.  It is the only pin on the advection flux, whose
stencil body is not in the reference tree: a support-2 scheme of lower order, or a sign slip in the
flux, fails this test.

Measured (f64, maximum error over the peak value).  Whole domain, the reference's time step: 7.4 % at
24^3, 2.3 % at 36^3, then a floor of 1.1 % at every finer grid and every smaller time step: the source
sits three standard deviations (sigma = 0.10) from the inflow faces, whose ghost cells keep their
initial values, exp(-4.5) = 1.1 % of the peak, while the analytic blob moves away.  Within 0.2 of the
blob centre and with a sixteenth of the stable time step (so that the forward Euler error, first order in
dt, stays below the spatial one): 3.5 % at 32^3, 1.0 % at 48^3, 0.40 % at 64^3, 0.079 % at 96^3, 0.021 % at
128^3, i.e. the flux is third order accurate or better.
"""
import numpy as np
import pytest


def heat_kernel(position_field, centre, nu, point_mag, t, real_t):
    x, y, z = position_field
    r2 = (x - centre) ** 2 + (y - centre) ** 2 + (z - centre) ** 2
    return (point_mag / (4 * real_t(np.pi) * nu * t) ** 1.5 * np.exp(-r2 / (4 * nu * t))).astype(real_t)


def run_point_source(sim_cls, n, real_t=np.float64, t_start=5.0, t_end=5.4, nu=1e-3, dt_prefac=1.0, window=None):
    sim = sim_cls(grid_size=(n, n, n), x_range=1.0, kinematic_viscosity=nu, flow_type="passive_vector",
                  real_t=real_t, time=t_start)
    point_mag = 4.0 * np.pi * nu * t_start**1.5
    pos = np.asarray(sim.position_field)
    w0 = heat_kernel(pos, 0.3, nu, point_mag, t_start, real_t)
    sim.primary_vector_field[...] = np.stack([w0, w0, w0])
    sim.velocity_field[...] = real_t(1.0)
    while sim.time < t_end - 1e-12:
        dt = min(float(sim.compute_stable_timestep(dt_prefac=dt_prefac)), t_end - sim.time)
        sim.time_step(dt=dt)
    g = sim.ghost_size
    inner = (slice(g, -g),) * 3
    centre = 0.3 + (t_end - t_start)
    want = heat_kernel(pos, centre, nu, point_mag, t_end, real_t)[inner]
    got = np.asarray(sim.primary_vector_field)[(slice(None),) + inner]
    scale = float(np.max(np.abs(want)))
    if window is not None:
        # only the cells within `window` of the blob centre: the source sits three standard deviations from
        # the inflow faces, whose frozen ghost values (1 % of the peak) are not part of the analytic problem
        near = np.all(np.abs(pos[:, g:-g, g:-g, g:-g] - centre) <= window, axis=0)
        return [float(np.max(np.abs(got[c] - want)[near])) / scale for c in range(3)], got
    return [float(np.max(np.abs(got[c] - want))) / scale for c in range(3)], got


def test_oracle_point_source_errors_shrink_with_resolution():
    from oracle.flow_3d import FlowSimulator3D

    err24, _ = run_point_source(FlowSimulator3D, 24)
    err36, _ = run_point_source(FlowSimulator3D, 36)
    assert max(err36) < 0.6 * max(err24), (err24, err36)
    assert max(err36) < 0.08, err36
    # the three components carry the same scalar
    assert max(err36) - min(err36) < 1e-12


@pytest.mark.gpu
def test_gpu_point_source_matches_heat_kernel(smi):
    from oracle.flow_3d import FlowSimulator3D
    from sopht_mpi_amd.simulator import UnboundedFlowSimulator3D

    kw = dict(t_end=5.2, dt_prefac=0.0625, window=0.2)
    err32, _ = run_point_source(UnboundedFlowSimulator3D, 32, **kw)
    err64, _ = run_point_source(UnboundedFlowSimulator3D, 64, **kw)
    err96, _ = run_point_source(UnboundedFlowSimulator3D, 96, **kw)
    assert max(err64) < max(err32) / 6, (err32, err64)   # third order: a factor of eight
    assert max(err96) < max(err64) / 3, (err64, err96)   # (2/3)^3 = 0.30
    assert max(err96) < 1.5e-3, err96
    # and the device run is the oracle run, step for step, at a size the oracle finishes in seconds
    e_gpu, f_gpu = run_point_source(UnboundedFlowSimulator3D, 24)
    e_cpu, f_cpu = run_point_source(FlowSimulator3D, 24)
    assert np.max(np.abs(f_gpu - f_cpu)) <= 1e-10 * np.max(np.abs(f_cpu))
