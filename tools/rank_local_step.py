"""What ONE rank of a multi-GPU step costs in kernels, measured on one GPU.

One process plays rank R of N z-slabs of the benchmark grid through the library's RCCL branches (its streams, events,
groups, chunked transposes and launch order), linked against the RCCL stand-in in its single-rank dry mode
(SMI_STANDIN_ALONE=1: sends and receives are accepted and dropped).  Nothing travels, so the fields hold garbage after
the first exchange - the TIMINGS are the point: the per-kernel table and the step time of the rank-local work, i.e.
the compute term of the multi-GPU model (kernels of a 1 / N share of the grid, their launch tails, the chunk-wise x
transforms), with the link time taken as zero.

usage: SOPHT_MPI_AMD_LIB=tests/rccl_standin/libsophtmi_standin.so SMI_STANDIN_ALONE=1 SOPHT_MI_TRANSPORT=rccl \
       python tools/rank_local_step.py [grid 512] [ranks 8] [rank 3] [steps 10]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sopht-mpi_amd")]
assert os.environ.get("SMI_STANDIN_ALONE") == "1" and "standin" in os.environ.get("SOPHT_MPI_AMD_LIB", ""), __doc__
os.environ["SOPHT_MI_TRANSPORT"] = "rccl"

from sopht_mpi_amd import _lib  # noqa: E402
from sopht_mpi_amd.simulator import UnboundedFlowSimulator3D  # noqa: E402
from sopht_mpi_amd.world import SingleWorld  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank = int(sys.argv[3]) if len(sys.argv) > 3 else nranks // 2 - 1
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10


class OneOfMany(SingleWorld):
    """This process as rank `rank` of `nranks`: collectives of the host-side world return the local value."""

    def sendrecv(self, send_arrays, recv_arrays):
        pass


OneOfMany.rank, OneOfMany.size = rank, nranks
sim = UnboundedFlowSimulator3D(grid_size=(n, n, n), x_range=1.0, kinematic_viscosity=1e-3, flow_type="navier_stokes",
                               real_t=np.float32, ghost_size=2, penalty_zone_width=2, world=OneOfMany())
ctx = sim.ctx
assert ctx.transport == "rccl" and list(sim.mpi_construct.grid_topology) == [nranks, 1, 1]
rng = np.random.default_rng(rank)
sim.vorticity_field[...] = (1e-3 * rng.standard_normal((3,) + ctx.padded_shape)).astype(np.float32)
sim.compute_flow_velocity(free_stream_velocity=None)
dt = 1e-4
for _ in range(3):
    sim.time_step(dt=dt)
ctx.sync()
t0 = time.perf_counter()
for _ in range(steps):
    sim.time_step(dt=dt)
ctx.sync()
wall_ms = 1e3 * (time.perf_counter() - t0) / steps
_lib.call("smi_prof_enable", ctx.handle, 1)
for _ in range(steps):
    sim.time_step(dt=dt)
ctx.sync()
prof = _lib.prof_collect(ctx)
chunks = _lib.C.c_int(0)
_lib.call("smi_ctx_get_option", ctx.handle, b"transpose_chunks", _lib.C.byref(chunks))
kernels = {k: {"launches_per_step": v["launches"] / steps, "ms_per_step": v["ms"] / steps}
           for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]) if k not in ("step", "flow_step")}
print(json.dumps({
    "what": "rank-local work of one rank of a multi-GPU step, nothing travelling (see the file's docstring)",
    "grid": n, "ranks": nranks, "rank": rank, "steps": steps, "transpose_chunks": chunks.value,
    "ms_per_step_wall_no_profiling": wall_ms,
    "ms_per_step_kernels_sum": sum(v["ms_per_step"] for v in kernels.values()),
    "kernels": kernels}))
