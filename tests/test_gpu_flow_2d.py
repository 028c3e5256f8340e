"""GPU parity of the 2D path: operators and whole steps of UnboundedFlowSimulator2D against the
oracle's 2D restatement, and the reference's Lamb-Oseen case on the GPU."""
import ctypes as C

import numpy as np
import pytest

from helpers import assert_close, rand_field
from oracle import stencils_2d as st2
from oracle.flow_2d import FlowSimulator2D as OracleSim2D
from oracle.poisson import UnboundedPoissonSolver2D

pytestmark = pytest.mark.gpu


def _ctx(smi, shape, g, real_t):
    from sopht_mpi_amd.utils import MPIConstruct2D

    return MPIConstruct2D(shape[0], shape[1], real_t=real_t).device_context(g)


@pytest.mark.parametrize("real_t", [np.float32, np.float64])
@pytest.mark.parametrize("shape,g", [((16, 16), 2), ((12, 20), 2), ((9, 13), 3)])
def test_2d_operators(smi, shape, g, real_t):
    rng = np.random.default_rng(3)
    ctx = _ctx(smi, shape, g, real_t)
    pshape = tuple(n + 2 * g for n in shape)
    call = smi._lib.call
    f = rand_field(rng, pshape, real_t)
    vel = rand_field(rng, (2,) + pshape, real_t)
    df, dvel = ctx.from_numpy(f), ctx.from_numpy(vel)
    dflux = ctx.from_numpy(rand_field(rng, pshape, real_t))
    call("smi2d_diffusion_flux", ctx.handle, dflux.ptr, df.ptr, 0.21)
    want = np.zeros_like(f)
    st2.diffusion_flux(want, f, real_t(0.21), g)
    assert_close(dflux.get(), want, real_t, what="2D diffusion flux")
    call("smi2d_diffusion_timestep", ctx.handle, df.ptr, dflux.ptr, 0.21)
    wf, wflux = f.copy(), np.zeros_like(f)
    st2.diffusion_timestep(wf, wflux, real_t(0.21), g)
    assert_close(df.get(), wf, real_t, what="2D diffusion timestep")
    da = ctx.from_numpy(f)
    call("smi2d_advection_timestep_eno3", ctx.handle, da.ptr, dflux.ptr, dvel.ptr, 0.05)
    wa, wflux = f.copy(), np.zeros_like(f)
    st2.advection_timestep_eno3(wa, wflux, vel, real_t(0.05), g)
    assert_close(da.get(), wa, real_t, factor=2.0, what="2D advection timestep")
    assert_close(dflux.get(), wflux, real_t, scale=np.abs(wa).max(), factor=2.0, what="2D advection flux")
    dcurl = ctx.from_numpy(vel)
    call("smi2d_outplane_field_curl", ctx.handle, dcurl.ptr, df.ptr, 0.5)
    wcurl = vel.copy()
    st2.outplane_field_curl(wcurl, wf, real_t(0.5), g)
    assert_close(dcurl.get(), wcurl, real_t, what="2D outplane curl")
    dw = ctx.from_numpy(f)
    call("smi2d_update_vorticity_from_velocity_forcing", ctx.handle, dw.ptr, dvel.ptr, 0.25)
    ww = f.copy()
    st2.update_vorticity_from_velocity_forcing(ww, vel, real_t(0.25), g)
    assert_close(dw.get(), ww, real_t, what="2D forcing update")
    vals = smi._lib.darray([0.5, -1.5])
    call("smi2d_add_fixed_val", ctx.handle, dcurl.ptr, dvel.ptr, 2, vals)
    assert_close(dcurl.get(), vel + np.array([0.5, -1.5], dtype=real_t).reshape(2, 1, 1), real_t, what="2D add")
    out = C.c_double(0.0)
    call("smi2d_reduce_max_abs_sum", ctx.handle, dvel.ptr, C.byref(out))
    assert real_t(out.value) == np.sum(np.fabs(vel), axis=0)[g:-g, g:-g].max()
    call("smi2d_reduce_max", ctx.handle, df.ptr, C.byref(out))
    assert real_t(out.value) == wf[g:-g, g:-g].max()


@pytest.mark.parametrize("real_t", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(16, 16), (24, 40), (12, 18)])
def test_2d_poisson(smi, shape, real_t):
    from sopht_mpi_amd.numeric.eulerian_grid_ops import UnboundedPoissonSolverMPI2D
    from sopht_mpi_amd.utils import MPIConstruct2D

    rng = np.random.default_rng(4)
    g = 2
    construct = MPIConstruct2D(*shape, real_t=real_t)
    solver = UnboundedPoissonSolverMPI2D(*shape, construct, g, x_range=1.3, real_t=real_t)
    pshape = tuple(n + 2 * g for n in shape)
    rhs = rand_field(rng, pshape, real_t)
    sol0 = rand_field(rng, pshape, real_t)
    drhs, dsol = solver.ctx.from_numpy(rhs), solver.ctx.from_numpy(sol0)
    solver.solve(dsol, drhs)
    oracle = UnboundedPoissonSolver2D(*shape, x_range=1.3, real_t=real_t)
    want = sol0.copy()
    oracle.solve(want, rhs, g)
    assert_close(dsol.get(), want, real_t, factor=4.0, what="2D Poisson")


@pytest.mark.parametrize("real_t", [np.float32, np.float64])
@pytest.mark.parametrize("flow_type,free_stream", [("passive_scalar", False), ("navier_stokes", True),
                                                   ("navier_stokes_with_forcing", True)])
def test_2d_flow_steps(smi, flow_type, free_stream, real_t):
    from sopht_mpi_amd.simulator import UnboundedFlowSimulator2D

    rng = np.random.default_rng(6)
    kw = dict(grid_size=(32, 48), x_range=1.0, kinematic_viscosity=5e-3, flow_type=flow_type, real_t=real_t,
              with_free_stream_flow=free_stream)
    gpu, cpu = UnboundedFlowSimulator2D(**kw), OracleSim2D(**kw)
    x, y = cpu.position_field
    w0 = (np.exp(-((x - 0.45) ** 2 + (y - 0.3) ** 2) / (2 * 0.06**2))
          - 0.5 * np.exp(-((x - 0.6) ** 2 + (y - 0.35) ** 2) / (2 * 0.05**2))).astype(real_t)
    gpu.primary_scalar_field[...] = w0
    cpu.primary_scalar_field[...] = w0
    fs = np.array([1.0, -0.5])
    if flow_type == "passive_scalar":
        vel = (rng.random(cpu.velocity_field.shape) - 0.5).astype(real_t)
        gpu.velocity_field[...] = vel
        cpu.velocity_field[...] = vel
    else:
        for sim in (gpu, cpu):
            sim.compute_velocity_from_vorticity()
        assert_close(gpu.velocity_field.get(), cpu.velocity_field, real_t, factor=4.0, what="initial 2D velocity")
    for step in range(4):
        dt = cpu.compute_stable_timestep(dt_prefac=0.5)
        assert abs(gpu.compute_stable_timestep(dt_prefac=0.5) - dt) <= 1e-5 * dt
        if flow_type == "navier_stokes_with_forcing":
            forcing = (rng.random(cpu.velocity_field.shape) - 0.5).astype(real_t) * real_t(0.1)
            gpu.eul_grid_forcing_field[...] = forcing
            cpu.eul_grid_forcing_field[...] = forcing
        kwargs = {} if flow_type == "passive_scalar" else {"free_stream_velocity": fs}
        gpu.time_step(dt=dt, **kwargs)
        cpu.time_step(dt, **kwargs)
        assert_close(gpu.primary_scalar_field.get(), cpu.primary_scalar_field, real_t, factor=2.0 + step,
                     what=f"2D {flow_type} field step {step}")
        if flow_type != "passive_scalar":
            assert_close(gpu.velocity_field.get(), cpu.velocity_field, real_t, factor=4.0 + step,
                         what=f"2D velocity step {step}")
    assert gpu.time == cpu.time
    if flow_type != "passive_scalar":
        assert abs(float(gpu.get_max_vorticity()) - float(cpu.get_max_vorticity())) <= 1e-5 * abs(
            float(cpu.get_max_vorticity()))
    if flow_type == "navier_stokes_with_forcing":
        assert not np.any(gpu.eul_grid_forcing_field.get())


def test_2d_invalid_arguments(smi):
    from sopht_mpi_amd.simulator import UnboundedFlowSimulator2D

    with pytest.raises(ValueError):
        UnboundedFlowSimulator2D(grid_size=(16, 16), x_range=1.0, kinematic_viscosity=1.0, flow_type="bogus")
    with pytest.raises(ValueError):
        UnboundedFlowSimulator2D(grid_size=(16, 16), x_range=1.0, kinematic_viscosity=1.0,
                                 flow_type="passive_scalar", with_free_stream_flow=True)
    with pytest.raises(ValueError):
        UnboundedFlowSimulator2D(grid_size=(16, 16), x_range=1.0, kinematic_viscosity=1.0, ghost_size=1)
