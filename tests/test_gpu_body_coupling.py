"""GPU: two-way coupling of a rigid stand-in body with the flow, and its restart.

The loop is the reference's (examples/3d_examples/FlowPastRodCase/flow_past_rod_case.py:253-276): per flow step the
body's integrator takes several sub-steps, each of which asks the interactor for the flow loads and is followed by
the interactor's own sub-step; then the interaction is spread and the flow steps.  PyElastica is replaced by
``RigidBodyIntegrator``.  The restart follows flow_past_freely_rotating_rod_case.py:212-247, 269-292: flow fields
through ``MPIIO``, the forcing grid's mismatch fields through a second ``MPIIO``, the body through its own state
file; a run that is saved after N cycles, rebuilt from the files and continued for N cycles must end bit for bit
where the uninterrupted run of 2 N cycles ends (node-ordered spread: no atomics in the data path).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 48
FREE_STREAM = np.array([1.0, 0.0, 0.0])
SUBSTEPS = 3


def _build(kind):
    from sopht_mpi_amd.simulator import (CosseratRodFlowInteraction, RigidBodyFlowInteractionMPI, RigidBodyIntegrator,
                                         UnboundedFlowSimulator3D, rod_mass_and_inertia, sphere_mass_and_inertia)
    from sopht_mpi_amd.simulator.immersed_body.forcing_grids import (CosseratRodSurfaceForcingGrid, RigidSphereBody,
                                                                     SphereForcingGrid, StraightRodBody)
    from sopht_mpi_amd.utils import MPIIO

    real_t = np.float32
    sim = UnboundedFlowSimulator3D(grid_size=(N, N, N), x_range=1.0, kinematic_viscosity=5e-3,
                                   flow_type="navier_stokes_with_forcing", real_t=real_t, with_free_stream_flow=True)
    sim.ctx.set_ordered_spread(True)
    common = dict(mpi_construct=sim.mpi_construct, mpi_ghost_exchange_communicator=sim.mpi_ghost_exchange_communicator,
                  eul_grid_forcing_field=sim.eul_grid_forcing_field, eul_grid_velocity_field=sim.velocity_field,
                  virtual_boundary_stiffness_coeff=-2e4, virtual_boundary_damping_coeff=-1e1, dx=sim.dx, grid_dim=3)
    if kind == "sphere":
        body = RigidSphereBody(center=(0.4, 0.5, 0.5), base_radius=0.12)
        interactor = RigidBodyFlowInteractionMPI(rigid_body=body, forcing_grid_cls=SphereForcingGrid,
                                                 num_forcing_points_along_equator=36, **common)
        mass, inertia = sphere_mass_and_inertia(0.12, density=2.0)
        stepper = RigidBodyIntegrator(body, interactor, mass, inertia)
    else:
        # the freely rotating rod: pinned at its first node, inclined to the stream
        body = StraightRodBody(n_elements=16, start=(0.35, 0.5, 0.5), direction=(0.8, 0.6, 0.0), normal=(0.0, 0.0, 1.0),
                               base_length=0.32, base_radius=0.03)
        interactor = CosseratRodFlowInteraction(cosserat_rod=body, forcing_grid_cls=CosseratRodSurfaceForcingGrid,
                                                surface_grid_density_for_largest_element=10, **common)
        mass, inertia = rod_mass_and_inertia(0.32, 0.03, density=2.0)
        stepper = RigidBodyIntegrator(body, interactor, mass, inertia, pinned_node=0)
    sim.compute_flow_velocity(free_stream_velocity=FREE_STREAM)
    flow_io = MPIIO(mpi_construct=sim.mpi_construct, real_dtype=real_t)
    flow_io.define_eulerian_grid(origin=np.full(3, sim.dx / 2), dx=np.full(3, sim.dx), grid_size=np.array((N, N, N)),
                                 ghost_size=sim.ghost_size)
    flow_io.add_as_eulerian_fields_for_io(vorticity=sim.vorticity_field, velocity=sim.velocity_field)
    grid_io = MPIIO(mpi_construct=sim.mpi_construct, real_dtype=np.float64)
    grid_io.add_as_lagrangian_fields_for_io(
        lagrangian_grid=interactor.forcing_grid.position_field, lagrangian_grid_master_rank=interactor.master_rank,
        lagrangian_grid_name="mismatch_field",
        position_mismatch=interactor.global_lag_grid_position_mismatch_field,
        velocity_mismatch=interactor.global_lag_grid_velocity_mismatch_field)
    return dict(sim=sim, body=body, interactor=interactor, stepper=stepper, flow_io=flow_io, grid_io=grid_io)


def _cycles(run, count):
    sim, interactor, stepper = run["sim"], run["interactor"], run["stepper"]
    for _ in range(count):
        flow_dt = float(sim.compute_stable_timestep(dt_prefac=0.5))
        local_dt = flow_dt / SUBSTEPS
        body_time = sim.time
        for _ in range(SUBSTEPS):
            body_time = stepper.do_step(body_time, local_dt)
            interactor.time_step(dt=local_dt)
        interactor()
        sim.time_step(dt=flow_dt, free_stream_velocity=FREE_STREAM)


def _save(run, tmp_path):
    run["interactor"].sync_global_mismatch_fields()
    run["flow_io"].save(str(tmp_path / "flow.h5"), time=run["sim"].time)
    run["grid_io"].save(str(tmp_path / "forcing_grid.h5"), time=run["sim"].time)
    run["stepper"].save_state(str(tmp_path / "body.npz"), run["sim"].time)


def _load(run, tmp_path):
    run["sim"].time = run["flow_io"].load(str(tmp_path / "flow.h5"))
    run["grid_io"].load(str(tmp_path / "forcing_grid.h5"))
    assert run["stepper"].load_state(str(tmp_path / "body.npz")) == run["sim"].time
    run["interactor"].scatter_global_mismatch_fields(run["interactor"].forcing_grid.position_field)


@pytest.mark.parametrize("kind", ["sphere", "rod"])
def test_body_moves_under_the_flow_and_restart_is_bit_exact(smi, tmp_path, kind):
    cycles = 4
    whole = _build(kind)
    start = whole["body"].position_collection.copy()
    _cycles(whole, cycles)
    _save(whole, tmp_path)
    mid = whole["body"].position_collection.copy()
    _cycles(whole, cycles)
    body = whole["body"]
    if kind == "sphere":
        # the free stream pushes the sphere downstream, and it keeps accelerating
        assert body.velocity_collection[0, 0] > 0 and body.position_collection[0, 0] > mid[0, 0] > start[0, 0]
        assert abs(body.position_collection[1, 0] - start[1, 0]) < 1e-3 * (body.position_collection[0, 0] - start[0, 0])
    else:
        # the pin holds, the inclined rod swings towards the stream direction (its y extent shrinks)
        assert np.array_equal(body.position_collection[:, 0], start[:, 0])
        assert abs(body.position_collection[1, -1] - start[1, 0]) < abs(start[1, -1] - start[1, 0])
        omega_lab = body.director_collection[:, :, 0].T @ body.omega_collection[:, 0]
        assert omega_lab[2] < 0 and abs(omega_lab[2]) > 10 * max(abs(omega_lab[0]), abs(omega_lab[1]))
    assert np.all(np.isfinite(whole["sim"].vorticity_field.get())) and float(whole["sim"].get_max_vorticity()) > 0

    resumed = _build(kind)
    _load(resumed, tmp_path)
    assert np.array_equal(resumed["body"].position_collection, mid)
    _cycles(resumed, cycles)
    g = whole["sim"].ghost_size
    inner = (slice(None),) + (slice(g, -g),) * 3
    assert resumed["sim"].time == whole["sim"].time
    for name in ("position_collection", "velocity_collection", "director_collection", "omega_collection"):
        assert np.array_equal(getattr(resumed["body"], name), getattr(body, name)), name
    assert np.array_equal(resumed["sim"].vorticity_field.get()[inner], whole["sim"].vorticity_field.get()[inner])
    assert np.array_equal(resumed["sim"].velocity_field.get()[inner], whole["sim"].velocity_field.get()[inner])
    assert np.array_equal(resumed["interactor"].local_lag_grid_position_mismatch_field.get(),
                          whole["interactor"].local_lag_grid_position_mismatch_field.get())
