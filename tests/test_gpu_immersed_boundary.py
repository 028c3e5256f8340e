"""GPU parity: immersed-boundary kernels and the virtual boundary forcing against the oracle."""
import numpy as np
import pytest

from helpers import assert_close, rand_field
from oracle import immersed_boundary as oib

pytestmark = pytest.mark.gpu


def _nodes(rng, m, n, dx, g_margin=3.0):
    """Random Lagrangian nodes at least ``g_margin`` cells away from the domain faces, plus
    nodes sitting exactly on cell centres and cell faces (floor-division edge cases)."""
    lo, hi = g_margin * dx, (np.array(n[::-1]) - g_margin) * dx  # x, y, z extents
    pos = rng.uniform(0, 1, (3, m)) * (hi - lo).reshape(3, 1) + lo
    pos[:, 0] = (np.array([5, 6, 7]) + 0.5) * dx  # exactly on a cell centre
    pos[:, 1] = np.array([5, 6, 7]) * dx          # exactly on cell faces
    return pos


@pytest.mark.parametrize("kernel_type", ["cosine", "peskin"])
@pytest.mark.parametrize("real_t", [np.float32, np.float64])
@pytest.mark.parametrize("n", [(16, 16, 16), (16, 20, 24)])
def test_ib_operators(smi, n, real_t, kernel_type):
    from sopht_mpi_amd.numeric.immersed_boundary_ops import EulerianLagrangianGridCommunicatorMPI3D
    from sopht_mpi_amd.utils import MPIConstruct3D

    rng = np.random.default_rng(21)
    g, m = 2, 37
    dx = real_t(1.0 / n[2])
    shift = real_t(dx / 2)
    construct = MPIConstruct3D(*n, real_t=real_t)
    comm = EulerianLagrangianGridCommunicatorMPI3D(dx=dx, eul_grid_coord_shift=shift, interp_kernel_width=2,
                                                   real_t=real_t, mpi_construct=construct, ghost_size=g,
                                                   n_components=3, interp_kernel_type=kernel_type)
    ctx = comm.ctx
    pos = _nodes(rng, m, n, float(dx))
    dpos = ctx.from_numpy(pos)
    dsup = ctx.zeros((3, 4, 4, 4, m), np.float64)
    didx = ctx.zeros((3, m), np.int64)
    dwts = ctx.zeros((4, 4, 4, m), np.float64)
    comm.local_eulerian_grid_support_of_lagrangian_grid_kernel(dsup, didx, dpos)
    w_sup = np.empty((3, 4, 4, 4, m))
    w_idx = np.empty((3, m), dtype=np.int64)
    oib.local_eulerian_grid_support_of_lagrangian_grid(w_sup, w_idx, pos, dx, shift, 2, np.zeros(3, dtype=np.int64) - g)
    assert np.array_equal(didx.get(), w_idx), "nearest index must be bit exact"
    assert_close(dsup.get(), w_sup, np.float64, scale=float(dx), factor=8, what="support")
    comm.interpolation_weights_kernel(dwts, dsup)
    w_wts = np.empty((4, 4, 4, m))
    (oib.cosine_interpolation_weights if kernel_type == "cosine" else oib.peskin_interpolation_weights)(
        w_wts, w_sup, dx, real_t)
    assert_close(dwts.get(), w_wts, np.float64, factor=8, what="weights")
    assert_close(dsup.get(), w_sup, np.float64, factor=4, what="support after weights (mutated in place)")
    # partition of unity: interpolating the constant 2.0 returns 2.0
    # (tests/.../test_eulerian_lagrangian_grid_communicator_mpi_3d.py:114-149)
    const = ctx.from_numpy(np.full((3,) + ctx.padded_shape, 2.0, dtype=real_t))
    dlag = ctx.zeros((3, m), np.float64)
    comm.eulerian_to_lagrangian_grid_interpolation_kernel(dlag, const, dwts, didx)
    np.testing.assert_allclose(dlag.get(), 2.0, rtol=1e-6 if real_t == np.float32 else 1e-12)
    # E -> L of a random field
    eul = rand_field(rng, (3,) + ctx.padded_shape, real_t)
    deul = ctx.from_numpy(eul)
    comm.eulerian_to_lagrangian_grid_interpolation_kernel(dlag, deul, dwts, didx)
    w_lag = np.zeros((3, m))
    oib.eulerian_to_lagrangian_grid_interpolation(w_lag, eul, w_wts, w_idx, dx)
    assert_close(dlag.get(), w_lag, np.float64, factor=8, what="E->L")
    # L -> E with overlapping windows (+ single-rank ghost sum = ghost clearing)
    lag = rng.standard_normal((3, m))
    dl = ctx.from_numpy(lag)
    dfor = ctx.from_numpy(eul)
    comm.lagrangian_to_eulerian_grid_interpolation_kernel(dfor, dl, dwts, didx)
    w_for = eul.copy()
    oib.lagrangian_to_eulerian_grid_interpolation(w_for, lag, w_wts, w_idx)
    oib.clear_ghost_cells(w_for, g)
    assert_close(dfor.get(), w_for, real_t, factor=5, what="L->E")


@pytest.mark.parametrize("kernel_type", ["cosine", "peskin"])
@pytest.mark.parametrize("real_t", [np.float32, np.float64])
def test_virtual_boundary_forcing(smi, real_t, kernel_type):
    from sopht_mpi_amd.numeric.immersed_boundary_ops import VirtualBoundaryForcingMPI
    from sopht_mpi_amd.utils import MPIConstruct3D

    rng = np.random.default_rng(22)
    n, g, m = (16, 16, 24), 2, 200
    dx = real_t(1.0 / n[2])
    construct = MPIConstruct3D(*n, real_t=real_t)
    pos = _nodes(rng, m, n, float(dx))
    vel = rng.standard_normal((3, m))
    k, c = -5.0e2, -3.0e1
    gpu = VirtualBoundaryForcingMPI(construct, g, k, c, 3, dx, enable_eul_grid_forcing_reset=True,
                                    global_lag_grid_position_field=pos, interp_kernel_type=kernel_type)
    cpu = oib.VirtualBoundaryForcing(k, c, 3, dx, m, g, eul_real_t=real_t, interp_kernel_type=kernel_type)
    ctx = gpu.ctx
    u = rand_field(rng, (3,) + ctx.padded_shape, real_t)
    du = ctx.from_numpy(u)
    dforce = ctx.from_numpy(rand_field(rng, (3,) + ctx.padded_shape, real_t))
    for it in range(3):
        pos_t = pos + 0.01 * it
        gpu.compute_interaction_forcing(dforce, du, pos_t, vel)
        wforce = np.zeros_like(u)
        cpu.compute_interaction_force_on_eul_and_lag_grid(wforce, u, pos_t, vel)
        assert_close(gpu.local_lag_grid_flow_velocity_field.get(), cpu.lag_flow_velocity, np.float64, factor=8,
                     what="flow velocity on lag grid")
        assert_close(gpu.global_lag_grid_forcing_field, cpu.lag_forcing, np.float64, factor=1, what="lag forcing")
        assert_close(dforce.get(), wforce, real_t, factor=8, what="eul forcing")
        gpu.time_step(0.01)
        cpu.time_step(0.01)
        assert_close(gpu.local_lag_grid_position_mismatch_field.get(), cpu.lag_position_mismatch, np.float64,
                     factor=1, what="position mismatch")
    assert gpu.time == pytest.approx(cpu.time)


def test_interactor_api_and_grid_deviation(smi):
    """ImmersedBodyFlowInteractionMPI surface with a prescribed rigid node set; the grid
    deviation norm known answer (tests/test_simulator/immersed_body/
    test_immersed_body_interaction_mpi.py:181-188): a mismatch of 2 in every component gives 2 sqrt(d)."""
    from sopht_mpi_amd.simulator import PrescribedForcingGrid, RigidBodyFlowInteractionMPI, UnboundedFlowSimulator3D

    real_t, n = np.float32, (16, 16, 16)
    sim = UnboundedFlowSimulator3D(grid_size=n, x_range=1.0, kinematic_viscosity=1e-2,
                                   flow_type="navier_stokes_with_forcing", real_t=real_t,
                                   with_free_stream_flow=True)
    theta = np.linspace(0, 2 * np.pi, 40, endpoint=False)
    pos = np.stack([0.5 + 0.2 * np.cos(theta), 0.5 + 0.2 * np.sin(theta), 0.5 + 0 * theta])
    inter = RigidBodyFlowInteractionMPI(
        mpi_construct=sim.mpi_construct, mpi_ghost_exchange_communicator=sim.mpi_ghost_exchange_communicator,
        rigid_body=None, eul_grid_forcing_field=sim.eul_grid_forcing_field,
        eul_grid_velocity_field=sim.velocity_field, virtual_boundary_stiffness_coeff=-1e3,
        virtual_boundary_damping_coeff=-1e1, dx=sim.dx, grid_dim=3, forcing_grid_cls=PrescribedForcingGrid,
        positions=pos, max_spacing=float(sim.dx),
    )
    sim.velocity_field[...] = 1.0
    inter()
    assert np.any(sim.eul_grid_forcing_field.get() != 0)
    inter.compute_flow_forces_and_torques()
    assert inter.body_flow_forces.shape == (3, 1) and np.all(np.isfinite(inter.body_flow_forces))
    inter.local_lag_grid_position_mismatch_field[...] = 2.0
    assert inter.get_grid_deviation_error_l2_norm() == pytest.approx(2 * np.sqrt(3))
    inter.time_step(dt=1e-3)
    sim.time_step(dt=1e-3, free_stream_velocity=[1.0, 0.0, 0.0])
    assert np.all(np.isfinite(sim.vorticity_field.get()))


def test_pointwise_kernels_one_at_a_time(smi):
    """VirtualBoundaryForcingMPI.py:278-331: the velocity mismatch, the Euler-forward update of the position mismatch
    and the penalty force as separate calls on device arrays."""
    from sopht_mpi_amd.numeric.immersed_boundary_ops import VirtualBoundaryForcingMPI as VBF
    from sopht_mpi_amd.utils import MPIConstruct3D

    ctx = MPIConstruct3D(16, 16, 16, real_t=np.float32).device_context(2)
    rng = np.random.default_rng(21)
    flow, body, pm = (rng.standard_normal((3, 77)) for _ in range(3))
    d_flow, d_body, d_pm = ctx.from_numpy(flow), ctx.from_numpy(body), ctx.from_numpy(pm)
    d_vm, d_force = ctx.zeros((3, 77), np.float64), ctx.zeros((3, 77), np.float64)
    VBF.compute_lag_grid_velocity_mismatch_field(d_vm, d_flow, d_body)
    assert np.array_equal(d_vm.get(), flow - body)
    VBF.update_lag_grid_position_mismatch_field_via_euler_forward(d_pm, d_vm, 0.125)
    np.testing.assert_allclose(d_pm.get(), pm + 0.125 * (flow - body), rtol=0, atol=4e-16 * np.abs(pm).max())
    VBF.compute_lag_grid_forcing_field(d_force, d_pm, d_vm, -3.0e2, -1.5e1)
    want = -3.0e2 * d_pm.get() + -1.5e1 * (flow - body)
    np.testing.assert_allclose(d_force.get(), want, rtol=0, atol=4e-16 * np.abs(want).max())
    with pytest.raises(TypeError):
        VBF.compute_lag_grid_velocity_mismatch_field(np.zeros((3, 4)), d_flow, d_body)


@pytest.mark.parametrize("dim", [2, 3])
def test_ghost_sum_communicator_classes(smi, dim):
    """MPIGhostSumCommunicator3D / 2D on one rank: every neighbour is absent, so the ghost sum leaves the interior
    as it is and zeroes the ghost cells, exactly like clear_ghost_cells."""
    from oracle import immersed_boundary as oib
    from sopht_mpi_amd.numeric import immersed_boundary_ops as ibo
    from sopht_mpi_amd.utils import MPIConstruct2D, MPIConstruct3D

    g, n = 2, (8, 12, 16)[-dim:]
    construct = (MPIConstruct3D if dim == 3 else MPIConstruct2D)(*n, real_t=np.float32)
    comm = (ibo.MPIGhostSumCommunicator3D if dim == 3 else ibo.MPIGhostSumCommunicator2D)(g, construct)
    assert list(comm.field_size) == [k + 2 * g for k in n]
    rng = np.random.default_rng(5)
    for ncomp in (None, dim):
        shape = tuple(k + 2 * g for k in n) if ncomp is None else (ncomp,) + tuple(k + 2 * g for k in n)
        host = rng.standard_normal(shape).astype(np.float32)
        want = host.copy()
        oib.clear_ghost_cells(want if ncomp else want[None], g, dim)
        for method in (comm.ghost_sum, comm.clear_ghost_cells):
            dev = comm.ctx.from_numpy(host)
            method(dev)
            assert np.array_equal(dev.get(), want)
    with pytest.raises(ValueError):
        type(comm)(-1, construct)
