"""Host and launch overhead of the 3D one-call step on small grids (32^3, 64^3, 128^3): time to enqueue a step, wall
time per step with the GPU kept busy, the same with the time-step query of a real loop (a device-to-host read per
step), and a Python profile of the enqueue path at 64^3.  usage: python tools/small_grid_step.py"""
import sys, time, os
import numpy as np
ROOT = os.getcwd()
sys.path[:0] = [ROOT, os.path.join(ROOT, "sopht-mpi_amd")]
from sopht_mpi_amd.simulator import UnboundedFlowSimulator3D
import cProfile, pstats
for n in (32, 64, 128):
    sim = UnboundedFlowSimulator3D(grid_size=(n, n, n), x_range=1.0, kinematic_viscosity=1e-3, flow_type="navier_stokes", real_t=np.float32, ghost_size=2, penalty_zone_width=2)
    sim.vorticity_field[...] = (1e-3 * np.random.default_rng(0).standard_normal((3,) + sim.ctx.padded_shape)).astype(np.float32)
    sim.compute_flow_velocity(free_stream_velocity=None)
    dt = 1e-4
    for _ in range(50): sim.time_step(dt=dt)
    sim.ctx.sync()
    t0 = time.perf_counter()
    for _ in range(2000): sim.time_step(dt=dt)
    t1 = time.perf_counter()
    sim.ctx.sync()
    t2 = time.perf_counter()
    print(n, "host enqueue us/step", 1e6 * (t1 - t0) / 2000, "wall us/step", 1e6 * (t2 - t0) / 2000, flush=True)
    # with the dt query of a real loop
    t0 = time.perf_counter()
    for _ in range(2000):
        dt = sim.compute_stable_timestep()
        sim.time_step(dt=dt)
    sim.ctx.sync()
    print(n, "with compute_stable_timestep: wall us/step", 1e6 * (time.perf_counter() - t0) / 2000, flush=True)
    if n == 64:
        pr = cProfile.Profile(); pr.enable()
        for _ in range(2000): sim.time_step(dt=dt)
        pr.disable(); sim.ctx.sync()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
