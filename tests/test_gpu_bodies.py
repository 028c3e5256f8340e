"""GPU: the two immersed-body configurations of the benchmark list with this package's own forcing grids.

* flow past a rigid sphere (examples/3d_examples/FlowPastSphereCase/flow_past_sphere_case.py:18-81, 151-202):
  `SphereForcingGrid` on a reduced 64^3 grid, several interactor / flow step cycles against the oracle's
  simulator and `oracle.immersed_boundary.VirtualBoundaryForcing`;
* flow past a Cosserat rod (examples/3d_examples/FlowPastRodCase/flow_past_rod_case.py:110-132, 261-271):
  `CosseratRodSurfaceForcingGrid`, filter order 1, a grid with a 192-point axis, the interpolation called once
  per body sub-step and the spread once per flow step, as the rod's time integrator does;
* the sphere at full size (256^3, 11.7 thousand nodes whose windows overlap heavily): properties that hold at any
  size - spreading conserves the total force, interpolation is a partition of unity, everything stays finite.

The parity runs use the node-ordered spread (DeviceContext.set_ordered_spread), which accumulates overlapping
windows in the reference's serial node order, so that the Eulerian forcing does not depend on atomic scheduling.
"""
import numpy as np
import pytest

from helpers import assert_close
from oracle import immersed_boundary as oib
from oracle.flow_3d import FlowSimulator3D as OracleSim

pytestmark = pytest.mark.gpu


def _blob(pos, centre, sigma, amp, real_t):
    x, y, z = pos
    r2 = (x - centre[0]) ** 2 + (y - centre[1]) ** 2 + (z - centre[2]) ** 2
    return np.stack([a * np.exp(-r2 / (2 * sigma**2)) for a in amp]).astype(real_t)


def _run_cycles(gpu, cpu, interactor, ref, forcing_grid, cycles, substeps, free_stream, real_t, move=None):
    g = gpu.ghost_size
    inner = (slice(None),) + (slice(g, -g),) * 3
    for cycle in range(cycles):
        dt = float(cpu.compute_stable_timestep(dt_prefac=0.5))
        assert abs(gpu.compute_stable_timestep(dt_prefac=0.5) - dt) <= 1e-5 * dt
        if move is not None:
            move(cycle, dt)
        # the body's integrator: penalty forces on the nodes at every sub-step (interpolation only) ...
        for _ in range(substeps):
            interactor.compute_flow_forces_and_torques()
            ref.compute_interaction_force_on_lag_grid(cpu.velocity_field, forcing_grid.position_field,
                                                      forcing_grid.velocity_field)
            assert_close(interactor.global_lag_grid_forcing_field, ref.lag_forcing, np.float32, factor=2.0 + 2.0 * cycle,
                         what=f"node forces, cycle {cycle}")
            assert np.all(np.isfinite(interactor.body_flow_forces)) and np.all(np.isfinite(interactor.body_flow_torques))
        # ... then the mismatch state advances, the forces are spread, and the flow steps
        interactor.time_step(dt)
        ref.time_step(dt)
        interactor()
        ref.compute_interaction_force_on_eul_and_lag_grid(cpu.eul_grid_forcing_field, cpu.velocity_field,
                                                          forcing_grid.position_field, forcing_grid.velocity_field)
        # the forces amplify the round-off of the interpolated velocity by the (large) penalty coefficients:
        # measured 2.4e-6 of the largest forcing in the second cycle of the rod case
        assert_close(gpu.eul_grid_forcing_field.get(), cpu.eul_grid_forcing_field, real_t, factor=4.0 + 2.0 * cycle,
                     what=f"Eulerian forcing {cycle}")
        gpu.time_step(dt=dt, free_stream_velocity=free_stream)
        cpu.time_step(dt=dt, free_stream_velocity=free_stream)
        assert_close(gpu.vorticity_field.get()[inner], cpu.vorticity_field[inner], real_t, factor=1.0 + cycle,
                     what=f"vorticity, cycle {cycle}")
        assert_close(gpu.velocity_field.get()[inner], cpu.velocity_field[inner], real_t, factor=2.0 + cycle,
                     what=f"velocity, cycle {cycle}")
    assert_close(interactor.local_lag_grid_position_mismatch_field.get(), ref.lag_position_mismatch, np.float32,
                 factor=4.0, what="position mismatch")


def _interactor_pair(gpu, cpu, cls, body_kw, body, grid_cls, stiffness, damping, **grid_kw):
    interactor = cls(
        mpi_construct=gpu.mpi_construct, mpi_ghost_exchange_communicator=gpu.mpi_ghost_exchange_communicator,
        eul_grid_forcing_field=gpu.eul_grid_forcing_field, eul_grid_velocity_field=gpu.velocity_field,
        virtual_boundary_stiffness_coeff=stiffness, virtual_boundary_damping_coeff=damping, dx=gpu.dx, grid_dim=3,
        forcing_grid_cls=grid_cls, **{body_kw: body}, **grid_kw)
    grid = interactor.forcing_grid
    scale = grid.get_maximum_lagrangian_grid_spacing() ** 2
    ref = oib.VirtualBoundaryForcing(stiffness * scale, damping * scale, 3, cpu.dx, grid.num_lag_nodes,
                                     gpu.ghost_size, eul_real_t=gpu.real_t)
    return interactor, ref, grid


@pytest.mark.parametrize("real_t", [np.float32, np.float64])
def test_flow_past_sphere_reduced(smi, real_t):
    from sopht_mpi_amd.simulator import RigidBodyFlowInteractionMPI, UnboundedFlowSimulator3D
    from sopht_mpi_amd.simulator.immersed_body.forcing_grids import RigidSphereBody, SphereForcingGrid

    n = 64
    kw = dict(grid_size=(n, n, n), x_range=1.0, kinematic_viscosity=4e-3, flow_type="navier_stokes_with_forcing",
              real_t=real_t, with_free_stream_flow=True)
    gpu, cpu = UnboundedFlowSimulator3D(**kw), OracleSim(**kw)
    gpu.ctx.set_ordered_spread(True)
    sphere = RigidSphereBody(center=(0.3, 0.5, 0.5), base_radius=0.15)
    interactor, ref, grid = _interactor_pair(gpu, cpu, RigidBodyFlowInteractionMPI, "rigid_body", sphere,
                                             SphereForcingGrid, -6e5 / 4, -3.5e2 / 4,
                                             num_forcing_points_along_equator=48)
    assert 600 < grid.num_lag_nodes < 900
    w0 = _blob(cpu.position_field, (0.55, 0.5, 0.45), 0.07, (0.4, -0.7, 0.9), real_t)
    free_stream = np.array([1.0, 0.0, 0.0])
    for sim in (gpu, cpu):
        sim.vorticity_field[...] = w0
        sim.compute_flow_velocity(free_stream_velocity=free_stream)
    _run_cycles(gpu, cpu, interactor, ref, grid, cycles=3, substeps=1, free_stream=free_stream, real_t=real_t)
    assert interactor.get_grid_deviation_error_l2_norm() > 0


def test_flow_past_rod_reduced(smi):
    from sopht_mpi_amd.simulator import CosseratRodFlowInteraction, UnboundedFlowSimulator3D
    from sopht_mpi_amd.simulator.immersed_body.forcing_grids import CosseratRodSurfaceForcingGrid, StraightRodBody

    real_t = np.float32
    # y axis of 192 cells (the 8 x 3 x 8 mixed-radix plan of the pruned Poisson solve), rod along y
    kw = dict(grid_size=(32, 192, 32), x_range=0.3, kinematic_viscosity=3e-3, flow_type="navier_stokes_with_forcing",
              real_t=real_t, with_free_stream_flow=True, filter_vorticity=True,
              filter_setting_dict={"order": 1, "type": "multiplicative"})
    gpu, cpu = UnboundedFlowSimulator3D(**kw), OracleSim(**kw)
    assert gpu.unbounded_poisson_solver.backend == "pruned"
    gpu.ctx.set_ordered_spread(True)
    rod = StraightRodBody(n_elements=40, start=(0.155, 0.45, 0.14), direction=(0.0, 1.0, 0.0), normal=(1.0, 0.0, 0.0),
                          base_length=0.9, base_radius=0.03)
    interactor, ref, grid = _interactor_pair(gpu, cpu, CosseratRodFlowInteraction, "cosserat_rod", rod,
                                             CosseratRodSurfaceForcingGrid, -5e4, -2e1,
                                             surface_grid_density_for_largest_element=16)
    assert grid.num_lag_nodes == 40 * 16
    w0 = _blob(cpu.position_field, (0.15, 0.9, 0.16), 0.05, (0.5, 0.8, -0.6), real_t)
    free_stream = np.array([1.0, 0.0, 0.0])
    for sim in (gpu, cpu):
        sim.vorticity_field[...] = w0
        sim.compute_flow_velocity(free_stream_velocity=free_stream)

    def move(cycle, dt):
        # prescribed rigid kinematics: the rod heaves in x and spins about its axis
        rod.velocity_collection[0, :] = 0.3
        rod.position_collection[0, :] += 0.3 * dt
        rod.omega_collection[2, :] = 2.0

    _run_cycles(gpu, cpu, interactor, ref, grid, cycles=3, substeps=3, free_stream=free_stream, real_t=real_t, move=move)
    assert interactor.body_flow_forces.shape == (3, 41) and interactor.body_flow_torques.shape == (3, 40)
    assert np.any(interactor.body_flow_torques != 0)


@pytest.mark.parametrize("body", ["sphere", "rod"])
def test_nodes_formed_on_the_device_match_the_host_grid(smi, body, monkeypatch):
    """A framed forcing grid forms its node arrays on the device (smi_ib_frame_nodes) and reduces the node forces to
    body loads there (smi_ib_frame_loads); the host methods of the same grid class are the check: identical
    kinematics for a translating, spinning body, and the same loads from the same interaction."""
    from sopht_mpi_amd.simulator import (CosseratRodFlowInteraction, RigidBodyFlowInteractionMPI,
                                         UnboundedFlowSimulator3D)
    from sopht_mpi_amd.simulator.immersed_body.forcing_grids import (CosseratRodSurfaceForcingGrid, RigidSphereBody,
                                                                     SphereForcingGrid, StraightRodBody)

    real_t, n = np.float32, 48
    sim = UnboundedFlowSimulator3D(grid_size=(n, n, n), x_range=1.0, kinematic_viscosity=1e-2,
                                   flow_type="navier_stokes_with_forcing", real_t=real_t, with_free_stream_flow=True)
    sim.ctx.set_ordered_spread(True)
    rng = np.random.default_rng(21)
    sim.velocity_field[...] = rng.normal(size=sim.velocity_field.shape).astype(real_t)
    th = 0.4
    spin = np.array([[np.cos(th), np.sin(th), 0.0], [-np.sin(th), np.cos(th), 0.0], [0.0, 0.0, 1.0]])

    def make(host):
        monkeypatch.setenv("SOPHT_MI_HOST_FORCING_GRID", "1" if host else "0")
        forcing = sim.ctx.field(3)
        common = dict(mpi_construct=sim.mpi_construct, mpi_ghost_exchange_communicator=sim.mpi_ghost_exchange_communicator,
                      eul_grid_forcing_field=forcing, eul_grid_velocity_field=sim.velocity_field,
                      virtual_boundary_stiffness_coeff=-5e4, virtual_boundary_damping_coeff=-2e1, dx=sim.dx, grid_dim=3)
        if body == "sphere":
            b = RigidSphereBody(center=(0.45, 0.5, 0.55), base_radius=0.2)
            b.velocity_collection[:, 0] = (0.3, -0.1, 0.2)
            b.omega_collection[:, 0] = (0.5, 1.5, -0.7)
            b.director_collection[:, :, 0] = spin
            it = RigidBodyFlowInteractionMPI(rigid_body=b, forcing_grid_cls=SphereForcingGrid,
                                             num_forcing_points_along_equator=40, **common)
        else:
            b = StraightRodBody(n_elements=24, start=(0.2, 0.5, 0.45), direction=(1.0, 0.2, 0.0), normal=(0.0, 0.0, 1.0),
                                base_length=0.6, base_radius=0.04)
            b.radius = b.radius * np.linspace(1.0, 0.6, b.n_elems)       # tapered: fewer points on thin elements
            b.velocity_collection[...] = rng.normal(size=b.velocity_collection.shape) * 0.2
            b.omega_collection[...] = np.linspace(-1, 1, b.n_elems) * np.array([[0.5], [1.0], [2.0]])
            it = CosseratRodFlowInteraction(cosserat_rod=b, forcing_grid_cls=CosseratRodSurfaceForcingGrid,
                                            surface_grid_density_for_largest_element=12, **common)
        return it, forcing

    rng = np.random.default_rng(22)
    dev, f_dev = make(host=False)
    rng = np.random.default_rng(22)
    host, f_host = make(host=True)
    assert dev.nodes_on_device and not host.nodes_on_device
    turn = np.array([[1.0, 0.0, 0.0], [0.0, np.cos(0.3), np.sin(0.3)], [0.0, -np.sin(0.3), np.cos(0.3)]])
    for cycle in range(2):
        for it in (dev, host):
            # the body moves and turns between the cycles: node arrays and moment arms must follow its pose
            moving = getattr(it.forcing_grid, "sphere", None) or it.forcing_grid.cosserat_rod
            moving.director_collection[...] = np.einsum("ij,jkn->ikn", turn, moving.director_collection)
            moving.position_collection[0] += 0.01
            it.compute_flow_forces_and_torques()
            it.time_step(1e-3)
            it()
        # the reference-style host transfer of the device interactor's node forces uses the arms of the CURRENT pose
        dev.compute_flow_forces_and_torques()
        body_f, body_t = np.zeros_like(dev.body_flow_forces), np.zeros_like(dev.body_flow_torques)
        dev.forcing_grid.transfer_forcing_from_grid_to_body(body_f, body_t, dev.global_lag_grid_forcing_field)
        assert_close(body_f, dev.body_flow_forces, np.float64, factor=1e3, what="host transfer of device forces")
        assert_close(body_t, dev.body_flow_torques, np.float64, factor=1e3, what="host transfer of device torques")
        host.compute_flow_forces_and_torques()
        gd, gh = dev.forcing_grid, host.forcing_grid
        scale = np.abs(gh.position_field).max()
        assert np.abs(gd.position_field - gh.position_field).max() <= 4e-16 * scale
        assert np.abs(gd.velocity_field - gh.velocity_field).max() <= 1e-15 * max(np.abs(gh.velocity_field).max(), 1.0)
        assert_close(dev.global_lag_grid_forcing_field, host.global_lag_grid_forcing_field, np.float64, factor=1e3,
                     what="node forces")
        assert_close(dev.body_flow_forces, host.body_flow_forces, np.float64, factor=1e3, what="body forces")
        assert_close(dev.body_flow_torques, host.body_flow_torques, np.float64, factor=1e3, what="body torques")
        assert_close(f_dev.get(), f_host.get(), real_t, factor=1.0, what="Eulerian forcing")
    assert np.any(dev.body_flow_torques != 0)


def test_ordered_spread_is_reproducible_and_follows_node_order(smi):
    """Many nodes in few cells: the node-ordered spread gives the same bits on every run and the oracle's
    serially accumulated values; the atomic spread agrees with both to rounding."""
    from sopht_mpi_amd.numeric.immersed_boundary_ops import EulerianLagrangianGridCommunicatorMPI3D
    from sopht_mpi_amd.utils import MPIConstruct3D

    real_t, n, g, m = np.float32, (16, 16, 16), 2, 2000
    rng = np.random.default_rng(77)
    dx = real_t(1.0 / 16)
    construct = MPIConstruct3D(*n, real_t=real_t)
    comm = EulerianLagrangianGridCommunicatorMPI3D(dx=dx, eul_grid_coord_shift=real_t(dx / 2), interp_kernel_width=2,
                                                   real_t=real_t, mpi_construct=construct, ghost_size=g, n_components=3)
    ctx = comm.ctx
    pos = rng.uniform(0.4, 0.6, (3, m))    # 2000 nodes in about 3^3 cells: every window overlaps hundreds of others
    lag = rng.standard_normal((3, m)) * 10.0 ** rng.uniform(-3, 3, (3, m))   # magnitudes that make the order matter
    dpos, dlag = ctx.from_numpy(pos), ctx.from_numpy(lag)
    dsup, didx, dwts = ctx.zeros((3, 4, 4, 4, m), np.float64), ctx.zeros((3, m), np.int64), ctx.zeros((4, 4, 4, m), np.float64)
    comm.local_eulerian_grid_support_of_lagrangian_grid_kernel(dsup, didx, dpos)
    comm.interpolation_weights_kernel(dwts, dsup)
    want = np.zeros((3,) + ctx.padded_shape, dtype=real_t)
    oib.lagrangian_to_eulerian_grid_interpolation(want, lag, dwts.get(), didx.get())

    def spread():
        out = ctx.field(3)
        comm.lagrangian_to_eulerian_grid_interpolation_kernel_without_ghost_sum(out, dlag, dwts, didx)
        return out.get()

    ctx.set_ordered_spread(True)
    first, second = spread(), spread()
    ctx.set_ordered_spread(False)
    atomic = spread()
    assert np.array_equal(first, second), "the node-ordered spread must be reproducible bit for bit"
    assert np.array_equal(first, want), "the node-ordered spread must equal the serial accumulation of the oracle"
    assert_close(atomic, want, real_t, factor=4.0, what="atomic spread vs serial order")


def test_sphere_case_full_size_properties(smi):
    """The reference's flow-past-sphere configuration at its real size: 256^3, sphere of diameter 0.4 with 192 points along the equator."""
    from sopht_mpi_amd.simulator import RigidBodyFlowInteractionMPI, UnboundedFlowSimulator3D
    from sopht_mpi_amd.simulator.immersed_body.forcing_grids import RigidSphereBody, SphereForcingGrid

    real_t, n = np.float32, 256
    sim = UnboundedFlowSimulator3D(grid_size=(n, n, n), x_range=1.0, kinematic_viscosity=1e-3,
                                   flow_type="navier_stokes_with_forcing", real_t=real_t, with_free_stream_flow=True)
    sphere = RigidSphereBody(center=(0.25, 0.5, 0.5), base_radius=0.2)
    interactor = RigidBodyFlowInteractionMPI(
        mpi_construct=sim.mpi_construct, mpi_ghost_exchange_communicator=sim.mpi_ghost_exchange_communicator,
        rigid_body=sphere, eul_grid_forcing_field=sim.eul_grid_forcing_field, eul_grid_velocity_field=sim.velocity_field,
        virtual_boundary_stiffness_coeff=-6e5 / 4, virtual_boundary_damping_coeff=-3.5e2 / 4, dx=sim.dx, grid_dim=3,
        forcing_grid_cls=SphereForcingGrid, num_forcing_points_along_equator=192)
    m = interactor.forcing_grid.num_lag_nodes
    assert 11000 < m < 12500
    free_stream = np.array([1.0, 0.0, 0.0])
    # partition of unity at every node: a uniform velocity interpolates to itself, so the penalty force is
    # damping * (U - 0) exactly where the mismatch is still zero
    sim.velocity_field[...] = np.broadcast_to(np.array([1.0, -2.0, 0.5], dtype=real_t).reshape(3, 1, 1, 1),
                                              sim.velocity_field.shape)
    interactor.compute_flow_forces_and_torques()
    np.testing.assert_allclose(interactor.local_lag_grid_flow_velocity_field.get(),
                               np.broadcast_to(np.array([[1.0], [-2.0], [0.5]]), (3, m)), rtol=2e-6)
    # a few coupled steps: forces stay finite, the spread carries the whole Lagrangian force
    sim.vorticity_field[...] = 0
    sim.compute_flow_velocity(free_stream_velocity=free_stream)
    for _ in range(3):
        dt = sim.compute_stable_timestep(dt_prefac=0.5)
        interactor.time_step(dt)
        interactor()
        forcing = sim.eul_grid_forcing_field.get().astype(np.float64)
        lag = interactor.global_lag_grid_forcing_field
        assert np.all(np.isfinite(forcing)) and np.all(np.isfinite(lag))
        total = forcing.sum(axis=(1, 2, 3)) * float(sim.dx) ** 3
        np.testing.assert_allclose(total, lag.sum(axis=1), rtol=2e-5, atol=2e-5 * np.abs(lag).sum(axis=1).max())
        sim.time_step(dt=dt, free_stream_velocity=free_stream)
    assert np.isfinite(float(sim.get_max_vorticity()))
    assert not np.any(sim.eul_grid_forcing_field.get())
    assert interactor.get_grid_deviation_error_l2_norm() > 0
