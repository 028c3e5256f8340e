"""The shape of the reference's flow-past-rod configuration (3D 384^3 flow past one Cosserat rod, immersed-boundary
surface forcing grid, vorticity filter of order 1) on one MI355X, with a rigid straight rod
standing in for the PyElastica body (no body mechanics here): geometry as
examples/3d_examples/FlowPastRodCase/flow_past_rod_case.py:40-60, 104-134, 314-316
(rod inclined in the x-z plane, D = y_range / 5, n_elem = 5 nx / 16, nx / 8 surface points around
the rod).  The interaction runs once per flow step.  Prints step time and per-kernel timings.

  python tools/rod_ib_bench.py [n] [steps]       cubic n^3 grid (default 384)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sopht-mpi_amd")]
from sopht_mpi_amd import _lib  # noqa: E402
from sopht_mpi_amd.simulator import CosseratRodFlowInteraction, UnboundedFlowSimulator3D  # noqa: E402
from sopht_mpi_amd.simulator.immersed_body import CosseratRodSurfaceForcingGrid, StraightRodBody  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 384
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    real_t = np.float32
    grid_size = (n, n, n)
    base_length, u_inf, reynolds = 1.0, 1.0, 100.0
    x_range = 1.8 * base_length
    y_range = z_range = x_range
    base_diameter = y_range / 5.0
    nu = u_inf * base_diameter / reynolds
    sim = UnboundedFlowSimulator3D(grid_size=grid_size, x_range=x_range, kinematic_viscosity=nu,
                                   flow_type="navier_stokes_with_forcing", with_free_stream_flow=True,
                                   real_t=real_t, filter_vorticity=True,
                                   filter_setting_dict={"order": 1, "type": "multiplicative"})
    incline = np.deg2rad(30.0)
    rod = StraightRodBody(n_elements=5 * n // 16, start=(0.2 * x_range, 0.5 * y_range, 0.75 * z_range),
                          direction=(np.sin(incline), 0.0, -np.cos(incline)), normal=(0.0, 1.0, 0.0),
                          base_length=base_length * 0.5, base_radius=base_diameter / 2 * 0.25)
    interactor = CosseratRodFlowInteraction(
        mpi_construct=sim.mpi_construct, mpi_ghost_exchange_communicator=sim.mpi_ghost_exchange_communicator,
        cosserat_rod=rod, eul_grid_forcing_field=sim.eul_grid_forcing_field,
        eul_grid_velocity_field=sim.velocity_field, virtual_boundary_stiffness_coeff=-2e4,
        virtual_boundary_damping_coeff=-1e1, dx=sim.dx, grid_dim=3, forcing_grid_cls=CosseratRodSurfaceForcingGrid,
        surface_grid_density_for_largest_element=n // 8)
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
    out = {"grid": list(grid_size), "poisson_backend": sim.unbounded_poisson_solver.backend,
           "lag_nodes": int(interactor.forcing_grid.num_lag_nodes), "rod_elements": rod.n_elems, "steps": steps,
           "ms_per_step": 1e3 * el / steps, "grid_point_updates_per_s": n**3 * steps / el,
           "max_vorticity": float(sim.get_max_vorticity()),
           "grid_deviation_l2": float(interactor.get_grid_deviation_error_l2_norm()),
           "net_force_on_rod": [float(v) for v in interactor.body_flow_forces.sum(axis=1)],
           "kernels": {k: {"launches": v["launches"], "avg_ms": v["ms"] / v["launches"]}
                       for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
