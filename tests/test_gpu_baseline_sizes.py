"""GPU against the oracle at the sizes the benchmark configurations are stated at.

* configuration 2 (3D 256^3 `navier_stokes`, the benchmark's seeded Gaussian blobs, single precision, pruned
  Poisson back end, one-call step): five steps on the GPU and on the oracle from the same state and with the same
  time step; after EVERY step the vorticity meets the pointwise north-star bar (rtol 1e-5) and one `TOL` in the max
  norm, the velocity 1e-5 of its largest value (flow_simulators_mpi_3d.py:382-424);
* configuration 4 (3D 256^3 flow past a rigid sphere: `navier_stokes_with_forcing`, free stream (1, 0, 0), sphere of
  diameter 0.4 at (0.25, 0.5, 0.5), 192 forcing points along the equator, stiffness -6e5/4 and damping -3.5e2/4,
  flow_past_sphere_case.py:18-81, 151-202): five interactor / flow step cycles with the node-ordered spread;
* one 512^3 step (the benchmark's own grid) against the oracle.

The measured errors of each step are printed (and logged with SMI_TEST_ERRLOG) in units of the bound.
"""
import numpy as np
import pytest

from helpers import TOL, assert_close, assert_pointwise
from oracle import immersed_boundary as oib
from oracle import poisson as opoisson
from oracle.flow_3d import FlowSimulator3D as OracleSim

pytestmark = pytest.mark.gpu

G = 2


def _velocity_factor(n):
    """Velocity bound in units of TOL (2e-6 of the largest value); 5 TOL = 1e-5, the north-star figure.

    The velocity is the curl of three Poisson solutions, and in single precision the pruned back end does not carry
    the rounding of the reference's mirrored coordinates 2 L - x (UnboundedPoissonSolverMPI3D.py:102-106; a relative
    error of up to 2 nx eps of the kernel next to the mirrored singularity): measured 3.3 TOL at 256^3 and 6.5 TOL at
    512^3 against the single-precision oracle - while that oracle is itself 2.0 / 4.7 TOL away from the same solve in
    double precision and the GPU result 2.2 / 5.3 TOL (asserted below: no further than the single-precision oracle
    within a quarter).  The vorticity, which the north star names, stays at 0.1 TOL."""
    return 5.0 if n <= 256 else 8.0


def _bench_vorticity(position_field, real_t):
    import bench

    return bench.synthetic_vorticity(position_field, real_t)


def _report(tag, step, gpu_field, cpu_field, real_t):
    inner = (slice(None),) + (slice(G, -G),) * 3
    a, b = gpu_field[inner].astype(np.float64), cpu_field[inner].astype(np.float64)
    scale = float(np.abs(b).max())
    err = float(np.abs(a - b).max()) / scale
    print(f"{tag} step {step}: max |gpu - oracle| / max|oracle| = {err:.3e} ({err / TOL[np.dtype(real_t)]:.2f} TOL)")
    return err


def _pair(n, flow_type, **kw):
    from sopht_mpi_amd.simulator import UnboundedFlowSimulator3D

    common = dict(grid_size=(n, n, n), x_range=1.0, kinematic_viscosity=1e-3, flow_type=flow_type,
                  real_t=np.float32, ghost_size=G, penalty_zone_width=2, **kw)
    gpu = UnboundedFlowSimulator3D(poisson_backend="pruned", **common)
    assert gpu.unbounded_poisson_solver.backend == "pruned" and gpu.fused_step
    return gpu, OracleSim(**common)


def _steps_against_oracle(n, steps):
    real_t = np.float32
    opoisson._WORKERS = 16
    gpu, cpu = _pair(n, "navier_stokes")
    w0 = _bench_vorticity(cpu.position_field, real_t)
    inner = (slice(None),) + (slice(G, -G),) * 3
    for sim in (gpu, cpu):
        sim.vorticity_field[...] = w0
        sim.compute_flow_velocity(free_stream_velocity=None)
    # the same solve in double precision: the GPU's single-precision velocity must be no further from it than the
    # single-precision oracle's
    dbl = OracleSim(grid_size=(n, n, n), x_range=1.0, kinematic_viscosity=1e-3, flow_type="navier_stokes",
                    real_t=np.float64, ghost_size=G, penalty_zone_width=2)
    dbl.vorticity_field[...] = w0
    del w0
    dbl.compute_flow_velocity(free_stream_velocity=None)
    u64 = dbl.velocity_field[inner].copy()
    del dbl
    scale = float(np.abs(u64).max())
    e_gpu = float(np.abs(gpu.velocity_field.get()[inner] - u64).max()) / scale
    e_ref = float(np.abs(cpu.velocity_field[inner] - u64).max()) / scale
    del u64
    print(f"{n}^3 initial velocity against the double-precision oracle: gpu {e_gpu:.3e}, single-precision oracle "
          f"{e_ref:.3e}")
    assert e_gpu <= 1.25 * e_ref + 1e-6, (e_gpu, e_ref)
    _report(f"{n}^3 velocity", 0, gpu.velocity_field.get(), cpu.velocity_field, real_t)
    assert_close(gpu.velocity_field.get()[inner], cpu.velocity_field[inner], real_t, factor=_velocity_factor(n),
                 what=f"{n}^3 initial velocity")
    dt = float(cpu.compute_stable_timestep())
    assert abs(float(gpu.compute_stable_timestep()) - dt) <= 1e-5 * dt
    for step in range(1, steps + 1):
        gpu.time_step(dt=dt)
        cpu.time_step(dt=dt)
        w = gpu.vorticity_field.get()
        _report(f"{n}^3 vorticity", step, w, cpu.vorticity_field, real_t)
        assert_pointwise(w[inner], cpu.vorticity_field[inner], real_t, what=f"{n}^3 vorticity after step {step}")
        assert_close(w[inner], cpu.vorticity_field[inner], real_t, factor=1.0, what=f"{n}^3 vorticity, step {step}")
        del w
        u = gpu.velocity_field.get()
        _report(f"{n}^3 velocity", step, u, cpu.velocity_field, real_t)
        assert_close(u[inner], cpu.velocity_field[inner], real_t, factor=_velocity_factor(n),
                     what=f"{n}^3 velocity, step {step}")
        del u
    assert abs(float(gpu.get_max_vorticity()) - float(cpu.get_max_vorticity())) <= 1e-5 * float(cpu.get_max_vorticity())


def test_configuration_2_five_steps_at_256_cubed(smi):
    _steps_against_oracle(256, steps=5)


def test_one_step_at_512_cubed(smi):
    """The benchmark's own grid: initial velocity solve and one step (the oracle holds about 30 GB for it)."""
    _steps_against_oracle(512, steps=1)


def test_configuration_4_sphere_at_256_cubed(smi):
    from sopht_mpi_amd.simulator import RigidBodyFlowInteractionMPI
    from sopht_mpi_amd.simulator.immersed_body.forcing_grids import RigidSphereBody, SphereForcingGrid

    real_t = np.float32
    opoisson._WORKERS = 16
    gpu, cpu = _pair(256, "navier_stokes_with_forcing", with_free_stream_flow=True)
    gpu.ctx.set_ordered_spread(True)
    sphere = RigidSphereBody(center=(0.25, 0.5, 0.5), base_radius=0.2)
    stiffness, damping = -6e5 / 4, -3.5e2 / 4
    interactor = RigidBodyFlowInteractionMPI(
        mpi_construct=gpu.mpi_construct, mpi_ghost_exchange_communicator=gpu.mpi_ghost_exchange_communicator,
        rigid_body=sphere, eul_grid_forcing_field=gpu.eul_grid_forcing_field, eul_grid_velocity_field=gpu.velocity_field,
        virtual_boundary_stiffness_coeff=stiffness, virtual_boundary_damping_coeff=damping, dx=gpu.dx, grid_dim=3,
        forcing_grid_cls=SphereForcingGrid, num_forcing_points_along_equator=192)
    grid = interactor.forcing_grid
    assert 11000 < grid.num_lag_nodes < 12500
    scale = grid.get_maximum_lagrangian_grid_spacing() ** 2
    ref = oib.VirtualBoundaryForcing(stiffness * scale, damping * scale, 3, cpu.dx, grid.num_lag_nodes, G,
                                     eul_real_t=real_t)
    free_stream = np.array([1.0, 0.0, 0.0])
    inner = (slice(None),) + (slice(G, -G),) * 3
    for sim in (gpu, cpu):
        sim.vorticity_field[...] = 0
        sim.compute_flow_velocity(free_stream_velocity=free_stream)
    for cycle in range(1, 6):
        dt = float(cpu.compute_stable_timestep(dt_prefac=0.5))
        assert abs(float(gpu.compute_stable_timestep(dt_prefac=0.5)) - dt) <= 1e-5 * dt
        interactor.time_step(dt)
        ref.time_step(dt)
        interactor()
        ref.compute_interaction_force_on_eul_and_lag_grid(cpu.eul_grid_forcing_field, cpu.velocity_field,
                                                          grid.position_field, grid.velocity_field)
        assert_close(interactor.global_lag_grid_forcing_field, ref.lag_forcing, np.float32, factor=4.0,
                     what=f"sphere node forces, cycle {cycle}")
        f = gpu.eul_grid_forcing_field.get()
        _report("sphere forcing", cycle, f, cpu.eul_grid_forcing_field, real_t)
        assert_close(f, cpu.eul_grid_forcing_field, real_t, factor=4.0, what=f"sphere Eulerian forcing, cycle {cycle}")
        del f
        gpu.time_step(dt=dt, free_stream_velocity=free_stream)
        cpu.time_step(dt=dt, free_stream_velocity=free_stream)
        w = gpu.vorticity_field.get()
        _report("sphere vorticity", cycle, w, cpu.vorticity_field, real_t)
        assert_pointwise(w[inner], cpu.vorticity_field[inner], real_t, what=f"sphere vorticity after cycle {cycle}")
        del w
        u = gpu.velocity_field.get()
        _report("sphere velocity", cycle, u, cpu.velocity_field, real_t)
        assert_close(u[inner], cpu.velocity_field[inner], real_t, factor=_velocity_factor(256),
                     what=f"sphere velocity, cycle {cycle}")
        del u
    assert_close(interactor.local_lag_grid_position_mismatch_field.get(), ref.lag_position_mismatch, np.float32,
                 factor=4.0, what="sphere position mismatch")
