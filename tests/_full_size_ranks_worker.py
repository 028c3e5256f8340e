"""Worker of the full-size multi-rank test: the benchmark's own workload (seeded Gaussian blobs, navier_stokes,
ghost 2) on the ranks it is launched on, a few steps, then diagnostics, per-plane sums of the interior and - for the
global planes named in the fourth argument (comma separated) - the whole interior planes of vorticity and velocity,
written to files that the test compares between a single-rank run and a run over slabs.  A fifth argument "pencil"
asks for the reference's default decomposition over z and y instead (the files then hold this rank's block of rows of
every plane, and its share of the plane sums)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sopht-mpi_amd"), os.path.join(ROOT, "tests")]

from bench import synthetic_vorticity  # noqa: E402
from sopht_mpi_amd.simulator import UnboundedFlowSimulator3D  # noqa: E402
from sopht_mpi_amd.world import get_world  # noqa: E402

world = get_world()
rank, size = world.rank, world.size
n, steps, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dump_planes = [int(z) for z in sys.argv[4].split(",")] if len(sys.argv) > 4 and sys.argv[4] else []
pencil = len(sys.argv) > 5 and sys.argv[5] == "pencil"
real_t = np.float32
sim = UnboundedFlowSimulator3D(grid_size=(n, n, n), x_range=1.0, kinematic_viscosity=1e-3, flow_type="navier_stokes",
                               real_t=real_t, ghost_size=2, penalty_zone_width=2,
                               rank_distribution=(0, 0, 1) if pencil else None)
assert sim.unbounded_poisson_solver.backend == "pruned"
sim.vorticity_field[...] = synthetic_vorticity(sim.position_field, real_t)
sim.compute_flow_velocity(free_stream_velocity=None)
dt = float(sim.compute_stable_timestep())
for _ in range(steps):
    sim.time_step(dt=dt)
diag = {"dt": dt, "max_vorticity": float(sim.get_max_vorticity()),
        "divergence_l2": float(sim.get_vorticity_divergence_l2_norm()), "dt_after": float(sim.compute_stable_timestep())}
g = sim.ghost_size
inner = (slice(g, -g),) * 3
w, u = sim.vorticity_field.get(), sim.velocity_field.get()
planes = np.stack([w[k][inner].astype(np.float64).sum(axis=(1, 2)) for k in range(3)]
                  + [np.abs(u[k][inner]).astype(np.float64).sum(axis=(1, 2)) for k in range(3)])
np.save(os.path.join(out_dir, f"planes_{size}_{rank}.npy"), planes)
topo = [int(t) for t in sim.mpi_construct.grid_topology]
coords = [int(t) for t in sim.mpi_construct.grid.coords]
nzl = n // topo[0]
if pencil and size > 1:
    assert topo[1] > 1, topo
    np.save(os.path.join(out_dir, f"block_{size}_{rank}.npy"), np.array(coords))
for z in dump_planes:
    if coords[0] * nzl <= z < (coords[0] + 1) * nzl:
        zl = g + z - coords[0] * nzl
        name = f"plane_{size}_{z}.npy" if topo[1] == 1 else f"plane_{size}_{z}_y{coords[1]}.npy"
        np.save(os.path.join(out_dir, name),
                np.stack([w[k][zl, g:-g, g:-g] for k in range(3)] + [u[k][zl, g:-g, g:-g] for k in range(3)]))
if rank == 0:
    with open(os.path.join(out_dir, f"diag_{size}.json"), "w") as f:
        json.dump(diag, f)
world.barrier()
print(f"rank {rank} ok, transport {sim.ctx.transport}", flush=True)
