"""GPU parity of the 2D path: operators and whole steps of UnboundedFlowSimulator2D against the
oracle's 2D restatement.  (The reference's Lamb-Oseen case runs in test_analytic_pins.py, the 2D path over several
ranks in test_gpu_slabs.py.)"""
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
    assert_close(da.get(), wa, real_t, factor=1.0, what="2D advection timestep")
    assert_close(dflux.get(), wflux, real_t, scale=np.abs(wa).max(), factor=1.0, what="2D advection flux")
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
    # the pointwise operators of the penalisation method (stencil_ops_2d/brinkmann_penalise_mpi_2d.py,
    # char_func_from_level_set_mpi_2d.py), on the plane of a 2D context
    from oracle import stencils_3d as st3
    from sopht_mpi_amd.numeric import eulerian_grid_ops as ops

    chi, pen = rng.random(pshape).astype(real_t), rand_field(rng, (2,) + pshape, real_t)
    brinkmann = ops.gen_brinkmann_penalise_pyst_mpi_kernel_2d(real_t=real_t, field_type="vector")
    dout = ctx.from_numpy(np.ones((2,) + pshape, dtype=real_t))
    brinkmann(penalised_vector_field=dout, penalty_factor=real_t(0.3), char_field=ctx.from_numpy(chi),
              penalty_vector_field=ctx.from_numpy(pen), vector_field=dvel)
    want2 = np.ones((2,) + pshape, dtype=real_t)
    st3.brinkmann_penalise(want2, real_t(0.3), chi, pen, vel)
    assert_close(dout.get(), want2, real_t, what="2D Brinkmann penalisation")
    heaviside = ops.gen_char_func_from_level_set_via_sine_heaviside_pyst_mpi_kernel_2d(blend_width=0.15, real_t=real_t)
    phi = (rng.random(pshape) - 0.5).astype(real_t)
    dchi = ctx.zeros(pshape)
    heaviside(char_func_field=dchi, level_set_field=ctx.from_numpy(phi))
    wchi = np.zeros_like(phi)
    st3.char_func_from_level_set_via_sine_heaviside(wchi, phi, real_t(0.15))
    assert_close(dchi.get(), wchi, real_t, scale=1.0, factor=4.0, what="2D characteristic function")
    assert brinkmann.kernel_support == 0 and heaviside.kernel_support == 0


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
    assert_close(dsol.get(), want, real_t, factor=1.0, what="2D Poisson")


@pytest.mark.parametrize("real_t", [np.float32, np.float64])
@pytest.mark.parametrize("flow_type,free_stream", [("passive_scalar", False), ("navier_stokes", True),
                                                   ("navier_stokes_with_forcing", True)])
@pytest.mark.parametrize("fused_step", [True, False])
def test_2d_flow_steps(smi, flow_type, free_stream, real_t, fused_step):
    """fused_step: one library call per step (smi2d_flow_step_ns) or the operator-by-operator sequence."""
    from sopht_mpi_amd.simulator import UnboundedFlowSimulator2D

    rng = np.random.default_rng(6)
    kw = dict(grid_size=(32, 48), x_range=1.0, kinematic_viscosity=5e-3, flow_type=flow_type, real_t=real_t,
              with_free_stream_flow=free_stream)
    gpu, cpu = UnboundedFlowSimulator2D(fused_step=fused_step, **kw), OracleSim2D(**kw)
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
        assert_close(gpu.velocity_field.get(), cpu.velocity_field, real_t, factor=10.0, what="initial 2D velocity")
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
        assert_close(gpu.primary_scalar_field.get(), cpu.primary_scalar_field, real_t, factor=1.0 + 0.5 * step,
                     what=f"2D {flow_type} field step {step}")
        if flow_type != "passive_scalar":
            assert_close(gpu.velocity_field.get(), cpu.velocity_field, real_t, factor=1.0 + 0.5 * step,
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


@pytest.mark.parametrize("kernel_type", ["cosine", "peskin"])
@pytest.mark.parametrize("real_t", [np.float32, np.float64])
def test_2d_immersed_boundary_operators(smi, real_t, kernel_type):
    from oracle import immersed_boundary as oib
    from sopht_mpi_amd.numeric.immersed_boundary_ops import EulerianLagrangianGridCommunicatorMPI2D
    from sopht_mpi_amd.utils import MPIConstruct2D

    rng = np.random.default_rng(31)
    n, g, m = (24, 32), 2, 41
    dx = real_t(1.0 / n[1])
    shift = real_t(dx / 2)
    construct = MPIConstruct2D(*n, real_t=real_t)
    comm = EulerianLagrangianGridCommunicatorMPI2D(dx=dx, eul_grid_coord_shift=shift, interp_kernel_width=2,
                                                   real_t=real_t, mpi_construct=construct, ghost_size=g,
                                                   n_components=2, interp_kernel_type=kernel_type)
    ctx = comm.ctx
    pos = rng.uniform(0.15, 0.6, (2, m))
    pos[:, 0] = (np.array([5, 6]) + 0.5) * float(dx)   # on a cell centre
    pos[:, 1] = np.array([5, 6]) * float(dx)           # on cell faces
    pshape = tuple(k + 2 * g for k in n)
    dpos = ctx.from_numpy(pos)
    dsup, didx = ctx.zeros((2, 4, 4, m), np.float64), ctx.zeros((2, m), np.int64)
    dwts = ctx.zeros((4, 4, m), np.float64)
    comm.local_eulerian_grid_support_of_lagrangian_grid_kernel(dsup, didx, dpos)
    w_sup, w_idx = np.empty((2, 4, 4, m)), np.empty((2, m), dtype=np.int64)
    oib.local_eulerian_grid_support_of_lagrangian_grid(w_sup, w_idx, pos, dx, shift, 2, np.zeros(2, dtype=np.int64) - g)
    assert np.array_equal(didx.get(), w_idx), "nearest index must be bit exact"
    assert_close(dsup.get(), w_sup, np.float64, scale=float(dx), factor=1, what="2D support")
    comm.interpolation_weights_kernel(dwts, dsup)
    w_wts = np.empty((4, 4, m))
    (oib.cosine_interpolation_weights if kernel_type == "cosine" else oib.peskin_interpolation_weights)(
        w_wts, w_sup, dx, real_t)
    assert_close(dwts.get(), w_wts, np.float64, factor=1, what="2D weights")
    const = ctx.from_numpy(np.full((2,) + pshape, 2.0, dtype=real_t))
    dlag = ctx.zeros((2, m), np.float64)
    comm.eulerian_to_lagrangian_grid_interpolation_kernel(dlag, const, dwts, didx)
    np.testing.assert_allclose(dlag.get(), 2.0, rtol=1e-6 if real_t == np.float32 else 1e-12)
    eul = rand_field(rng, (2,) + pshape, real_t)
    comm.eulerian_to_lagrangian_grid_interpolation_kernel(dlag, ctx.from_numpy(eul), dwts, didx)
    w_lag = np.zeros((2, m))
    oib.eulerian_to_lagrangian_grid_interpolation(w_lag, eul, w_wts, w_idx, dx)
    assert_close(dlag.get(), w_lag, np.float64, factor=2, what="2D E->L")
    lag = rng.standard_normal((2, m))
    dfor = ctx.from_numpy(eul)
    comm.lagrangian_to_eulerian_grid_interpolation_kernel(dfor, ctx.from_numpy(lag), dwts, didx)
    w_for = eul.copy()
    oib.lagrangian_to_eulerian_grid_interpolation(w_for, lag, w_wts, w_idx)
    oib.clear_ghost_cells(w_for, g, 2)
    assert_close(dfor.get(), w_for, real_t, factor=1, what="2D L->E")


@pytest.mark.parametrize("real_t", [np.float32, np.float64])
def test_2d_virtual_boundary_forcing(smi, real_t):
    from oracle import immersed_boundary as oib
    from sopht_mpi_amd.numeric.immersed_boundary_ops import VirtualBoundaryForcingMPI
    from sopht_mpi_amd.utils import MPIConstruct2D

    rng = np.random.default_rng(32)
    n, g, m = (32, 32), 2, 120
    dx = real_t(1.0 / n[1])
    construct = MPIConstruct2D(*n, real_t=real_t)
    theta = np.linspace(0, 2 * np.pi, m, endpoint=False)
    pos = np.stack([0.5 + 0.2 * np.cos(theta), 0.45 + 0.15 * np.sin(theta)])
    vel = rng.standard_normal((2, m)) * 0.1
    k, c = -4.0e2, -2.0e1
    gpu = VirtualBoundaryForcingMPI(mpi_construct=construct, ghost_size=g, virtual_boundary_stiffness_coeff=k,
                                    virtual_boundary_damping_coeff=c, grid_dim=2, dx=dx,
                                    global_lag_grid_position_field=pos)
    cpu = oib.VirtualBoundaryForcing(k, c, 2, dx, m, g, eul_real_t=real_t)
    ctx = gpu.ctx
    pshape = tuple(q + 2 * g for q in n)
    for step in range(3):
        u = rand_field(rng, (2,) + pshape, real_t)
        du, dforce = ctx.from_numpy(u), ctx.from_numpy(rand_field(rng, (2,) + pshape, real_t))
        gpu.compute_interaction_forcing(local_eul_grid_forcing_field=dforce, local_eul_grid_velocity_field=du,
                                        global_lag_grid_position_field=pos, global_lag_grid_velocity_field=vel)
        w_force = np.zeros((2,) + pshape, dtype=real_t)
        cpu.compute_interaction_force_on_eul_and_lag_grid(w_force, u, pos, vel)
        assert_close(gpu.global_lag_grid_forcing_field, cpu.lag_forcing, np.float64, factor=1,
                     what=f"2D lag forcing step {step}")
        assert_close(dforce.get(), w_force, real_t, factor=2, what=f"2D eul forcing step {step}")
        gpu.time_step(dt=1e-3)
        cpu.time_step(1e-3)
        pos = pos + 1e-3 * rng.standard_normal((2, m)) * 0.1
    assert_close(gpu.local_lag_grid_position_mismatch_field.get(), cpu.lag_position_mismatch, np.float64,
                 factor=1, what="2D position mismatch")


def test_2d_captured_step_follows_its_arguments(smi):
    """The one-call 2D step is replayed from a captured graph on a single rank.  A replay is only valid for the arrays
    and tables it was captured with: when one of them changes (a new scratch array here, then another free stream,
    then a step without free stream values changing the coefficients only) the step must still match the oracle."""
    from sopht_mpi_amd.simulator import UnboundedFlowSimulator2D

    real_t = np.float32
    kw = dict(grid_size=(32, 48), x_range=1.0, kinematic_viscosity=5e-3, flow_type="navier_stokes", real_t=real_t,
              with_free_stream_flow=True)
    gpu, cpu = UnboundedFlowSimulator2D(**kw), OracleSim2D(**kw)
    x, y = cpu.position_field
    w0 = np.exp(-((x - 0.45) ** 2 + (y - 0.3) ** 2) / (2 * 0.06**2)).astype(real_t)
    gpu.primary_scalar_field[...] = w0
    cpu.primary_scalar_field[...] = w0
    for sim in (gpu, cpu):
        sim.compute_velocity_from_vorticity()
    fs = np.array([1.0, -0.5])
    for step in range(10):
        if step == 5:
            gpu.buffer_scalar_field = gpu.ctx.zeros(gpu.buffer_scalar_field.shape)   # another array: new capture
        if step == 7:
            fs = np.array([-0.25, 0.75])                                              # coefficients only: same graph
        dt = cpu.compute_stable_timestep(dt_prefac=0.5 if step % 2 else 0.25)         # dt changes every step
        gpu.time_step(dt=dt, free_stream_velocity=fs)
        cpu.time_step(dt, free_stream_velocity=fs)
        assert_close(gpu.primary_scalar_field.get(), cpu.primary_scalar_field, real_t, factor=1.0 + 0.5 * step,
                     what=f"2D vorticity step {step}")
        assert_close(gpu.velocity_field.get(), cpu.velocity_field, real_t, factor=1.0 + 0.5 * step,
                     what=f"2D velocity step {step}")
