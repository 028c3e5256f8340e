"""A hundred 64^3 steps without the in-library profiler, for a rocprofv3 kernel trace of everything a step queues:
rocprofv3 --kernel-trace --stats -d out -o t -- python3 tools/trace_small_step.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sopht-mpi_amd")]
from sopht_mpi_amd.simulator import UnboundedFlowSimulator3D  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sim = UnboundedFlowSimulator3D(grid_size=(n, n, n), x_range=1.0, kinematic_viscosity=1e-3, flow_type="navier_stokes",
                               real_t=np.float32, ghost_size=2, penalty_zone_width=2)
sim.vorticity_field[...] = (1e-3 * np.random.default_rng(0).standard_normal((3,) + sim.ctx.padded_shape)).astype(np.float32)
sim.compute_flow_velocity(free_stream_velocity=None)
for _ in range(100):
    sim.time_step(dt=1e-4)
sim.ctx.sync()
