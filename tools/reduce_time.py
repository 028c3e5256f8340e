import os, sys, time
import numpy as np
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "sopht-mpi_amd")]
from sopht_mpi_amd.simulator import UnboundedFlowSimulator3D
n = 512
sim = UnboundedFlowSimulator3D(grid_size=(n, n, n), x_range=1.0, kinematic_viscosity=1e-3, flow_type="navier_stokes", real_t=np.float32)
sim.velocity_field[...] = np.random.default_rng(0).standard_normal((3,) + sim.ctx.padded_shape).astype(np.float32)
for f in (sim.compute_stable_timestep, sim.get_max_vorticity, sim.get_vorticity_divergence_l2_norm):
    f(); sim.ctx.sync()
    t = time.perf_counter()
    for _ in range(20): f()
    sim.ctx.sync()
    print(f.__name__, (time.perf_counter() - t) / 20 * 1e3, "ms")
