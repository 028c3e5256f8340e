"""The reference's flow-past-sphere configuration (3D 256^3 flow past a rigid sphere with an immersed-boundary
forcing grid) on one MI355X: navier_stokes_with_forcing, free stream (1, 0, 0), Re = 100,
sphere D = 0.4 at (0.25, 0.5, 0.5) (examples/3d_examples/FlowPastSphereCase/
flow_past_sphere_case.py:18-67), SphereForcingGrid with 1.875 D / dx
points along the equator (11 700 nodes at 256^3), fixed sphere.  Prints the step time and the per-kernel event timings,
the interpolation + forcing and the spreading kernels among them.

  python tools/sphere_ib_bench.py [grid] [steps]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sopht-mpi_amd")]
from sopht_mpi_amd import _lib  # noqa: E402
from sopht_mpi_amd.simulator import RigidBodyFlowInteractionMPI, UnboundedFlowSimulator3D  # noqa: E402
from sopht_mpi_amd.simulator.immersed_body import RigidSphereBody, SphereForcingGrid  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    real_t = np.float32
    u_inf, diameter, reynolds = 1.0, 0.4, 100.0
    nu = u_inf * diameter / reynolds
    sim = UnboundedFlowSimulator3D(grid_size=(n, n, n), x_range=1.0, kinematic_viscosity=nu,
                                   flow_type="navier_stokes_with_forcing", with_free_stream_flow=True,
                                   real_t=real_t, ghost_size=2)
    sphere = RigidSphereBody(center=(0.25, 0.5, 0.5), base_radius=diameter / 2)
    num_forcing_points_along_equator = int(1.875 * diameter / 1.0 * n)   # flow_past_sphere_case.py:65-67
    interactor = RigidBodyFlowInteractionMPI(
        mpi_construct=sim.mpi_construct, mpi_ghost_exchange_communicator=sim.mpi_ghost_exchange_communicator,
        rigid_body=sphere, eul_grid_forcing_field=sim.eul_grid_forcing_field,
        eul_grid_velocity_field=sim.velocity_field, virtual_boundary_stiffness_coeff=-6e5 / 4,
        virtual_boundary_damping_coeff=-3.5e2 / 4, dx=sim.dx, grid_dim=3, forcing_grid_cls=SphereForcingGrid,
        num_forcing_points_along_equator=num_forcing_points_along_equator)
    m = interactor.forcing_grid.num_lag_nodes
    spacing = interactor.forcing_grid.get_maximum_lagrangian_grid_spacing()
    fs = np.array([u_inf, 0.0, 0.0])
    sim.compute_flow_velocity(free_stream_velocity=fs)

    def step():
        dt = sim.compute_stable_timestep(dt_prefac=0.25)
        interactor()
        interactor.time_step(dt=dt)
        sim.time_step(dt=dt, free_stream_velocity=fs)

    for _ in range(5):
        step()
    sim.ctx.sync()
    _lib.call("smi_prof_enable", sim.ctx.handle, 1)
    _lib.call("smi_prof_reset", sim.ctx.handle)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sim.ctx.sync()
    el = time.perf_counter() - t0
    prof = _lib.prof_collect(sim.ctx)
    interactor.compute_flow_forces_and_torques()
    out = {"grid": n, "lag_nodes": m, "lag_spacing_over_dx": spacing / float(sim.dx), "steps": steps,
           "ms_per_step": 1e3 * el / steps, "grid_point_updates_per_s": n**3 * steps / el,
           "max_vorticity": float(sim.get_max_vorticity()),
           "grid_deviation_l2": float(interactor.get_grid_deviation_error_l2_norm()),
           "drag_force_x": float(interactor.body_flow_forces[0, 0]),
           "kernels": {k: {"launches": v["launches"], "avg_ms": v["ms"] / v["launches"]}
                       for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
