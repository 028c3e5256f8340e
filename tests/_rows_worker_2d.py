"""Worker of the multi-rank 2D GPU test: P ranks share the one GPU of the test box, each owns a slab of rows of the
plane (the reference's default 2D decomposition), transport is host-staged through the socket world.  Rank 0 gathers
the slabs and compares with the single-rank oracle on the whole plane: ghost-row exchange of every stencil operator,
the divided Poisson solve (rows, all-to-all, spectral columns), node ownership, the ghost sum of the spread, the reductions."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sopht-mpi_amd"), os.path.join(ROOT, "tests")]
os.environ["SOPHT_MI_TRANSPORT"] = os.environ.get("SOPHT_MI_TRANSPORT", "host")

from helpers import assert_close  # noqa: E402
from oracle import immersed_boundary as oib  # noqa: E402
from oracle.flow_2d import FlowSimulator2D as OracleSim2D  # noqa: E402
from sopht_mpi_amd.numeric.immersed_boundary_ops import VirtualBoundaryForcingMPI  # noqa: E402
from sopht_mpi_amd.simulator import UnboundedFlowSimulator2D  # noqa: E402
from sopht_mpi_amd.utils import MPIFieldCommunicator2D, MPIGhostCommunicator2D  # noqa: E402
from sopht_mpi_amd.world import get_world  # noqa: E402

world = get_world()
rank, size = world.rank, world.size
flow_type = sys.argv[1]
fused = sys.argv[2] == "fused"
n = tuple(int(v) for v in sys.argv[3].split("x"))
real_t = np.float64 if "double" in sys.argv[4:] else np.float32
kw = dict(grid_size=n, x_range=1.0, kinematic_viscosity=5e-3, flow_type=flow_type, real_t=real_t,
          with_free_stream_flow=flow_type != "passive_scalar")
gpu = UnboundedFlowSimulator2D(fused_step=fused, **kw)
cpu = OracleSim2D(**kw)
construct = gpu.mpi_construct
assert list(construct.grid_topology) == [size, 1]
assert list(construct.local_grid_size) == [n[0] // size, n[1]]
assert construct.previous_grid_along[0] == (rank - 1 if rank else -1)
g = gpu.ghost_size
comm = MPIFieldCommunicator2D(g, construct)
inner = (slice(g, -g),) * 2
rows = slice(g + rank * (n[0] // size), g + (rank + 1) * (n[0] // size))

# the block's coordinates are the rows of the whole plane's (each block runs its own linspace, as in the reference:
# equal up to rounding)
x, y = cpu.position_field
np.testing.assert_allclose(gpu.position_field[:, g:-g, g:-g], cpu.position_field[:, rows, g:-g],
                           rtol=4 * np.finfo(real_t).eps)

rng = np.random.default_rng(11)
w0 = (np.exp(-((x - 0.45) ** 2 + (y - 0.3 * n[0] / n[1]) ** 2) / (2 * 0.06**2))
      - 0.5 * np.exp(-((x - 0.6) ** 2 + (y - 0.55 * n[0] / n[1]) ** 2) / (2 * 0.05**2))).astype(real_t)
comm.scatter_global_scalar_field(gpu.primary_scalar_field, w0[inner])
cpu.primary_scalar_field[inner] = w0[inner]
fs = np.array([1.0, -0.5])


def gathered(field, ncomp=None):
    out = np.zeros(n if ncomp is None else (ncomp,) + n, dtype=real_t) if rank == 0 else None
    (comm.gather_local_scalar_field if ncomp is None else comm.gather_local_vector_field)(out, field)
    return out


# ghost-row exchange on its own: after it, every rank's ghost rows hold the neighbours' edge rows
halo = MPIGhostCommunicator2D(g, construct)
probe = gpu.ctx.zeros(gpu.primary_scalar_field.shape)
full = rng.standard_normal(tuple(k + 2 * g for k in n)).astype(real_t)
comm.scatter_global_scalar_field(probe, full[inner])
halo.exchange_scalar_field_init(probe)
halo.exchange_finalise()
got = probe.get()
want = full[rows.start - g : rows.stop + g, :].copy()
want[:, :g] = 0
want[:, -g:] = 0
if rank == 0:
    want[:g] = 0
if rank == size - 1:
    want[-g:] = 0
assert np.array_equal(got, want), "2D ghost rows after the exchange"

if flow_type == "passive_scalar":
    vel = (rng.random((2,) + tuple(k + 2 * g for k in n)) - 0.5).astype(real_t)
    comm.scatter_global_vector_field(gpu.velocity_field, vel[(slice(None),) + inner])
    cpu.velocity_field[(slice(None),) + inner] = vel[(slice(None),) + inner]
else:
    gpu.compute_velocity_from_vorticity()
    cpu.compute_velocity_from_vorticity()
    got_u = gathered(gpu.velocity_field, 2)
    if rank == 0:
        assert_close(got_u, cpu.velocity_field[(slice(None),) + inner], real_t, factor=10.0, what="initial 2D velocity")

body = ref = None
if flow_type == "navier_stokes_with_forcing":
    m = 90
    theta = np.linspace(0, 2 * np.pi, m, endpoint=False)
    # a ring that crosses every boundary between the row slabs (y extent 0.2 .. 0.8 of the y range)
    yr = n[0] / n[1]
    pos = np.stack([0.5 + 0.2 * np.cos(theta), yr * (0.5 + 0.3 * np.sin(theta))])
    lag_vel = np.zeros_like(pos)
    k_s, c_d = -4.0e2, -2.0e1
    body = VirtualBoundaryForcingMPI(mpi_construct=construct, ghost_size=g, virtual_boundary_stiffness_coeff=k_s,
                                     virtual_boundary_damping_coeff=c_d, grid_dim=2, dx=gpu.dx,
                                     global_lag_grid_position_field=pos)
    ref = oib.VirtualBoundaryForcing(k_s, c_d, 2, cpu.dx, m, g, eul_real_t=real_t)
    assert body.global_num_lag_nodes == m
    assert world.allreduce(float(body.local_num_lag_nodes), "sum") == m

for step in range(4):
    dt = cpu.compute_stable_timestep(dt_prefac=0.5)
    dt_gpu = gpu.compute_stable_timestep(dt_prefac=0.5)
    assert abs(dt_gpu - dt) <= 1e-5 * dt, (dt_gpu, dt)
    if body is not None:
        # the ring drifts along y, 0.7 of a cell per step: nodes change slabs between steps
        pos_now = pos + np.array([0.0, 0.7 * step * float(gpu.dx)]).reshape(2, 1)
        body.time_step(dt)
        ref.time_step(dt)
        # the interpolation windows reach into the ghost rows: current ghosts first, as the reference's interactor
        # does (immersed_body_flow_interaction_mpi.py:123-140)
        halo.exchange_vector_field_init(gpu.velocity_field)
        halo.exchange_finalise()
        body.compute_interaction_forcing(
            local_eul_grid_forcing_field=gpu.eul_grid_forcing_field, local_eul_grid_velocity_field=gpu.velocity_field,
            global_lag_grid_position_field=pos_now, global_lag_grid_velocity_field=lag_vel)
        cpu.eul_grid_forcing_field[...] = 0
        ref.compute_interaction_force_on_eul_and_lag_grid(cpu.eul_grid_forcing_field, cpu.velocity_field, pos_now,
                                                          lag_vel)
        got_f = gathered(gpu.eul_grid_forcing_field, 2)
        if rank == 0:
            assert_close(body.global_lag_grid_forcing_field, ref.lag_forcing, real_t, factor=1.0 + step,
                         what=f"2D lag forcing step {step}")
            assert_close(got_f, cpu.eul_grid_forcing_field[(slice(None),) + inner], real_t, factor=2.0 + step,
                         what=f"2D eul forcing step {step}")
    kwargs = {} if flow_type == "passive_scalar" else {"free_stream_velocity": fs}
    gpu.time_step(dt=dt, **kwargs)
    cpu.time_step(dt, **kwargs)
    got_w = gathered(gpu.primary_scalar_field)
    if rank == 0:
        assert_close(got_w, cpu.primary_scalar_field[inner], real_t, factor=1.0 + 0.5 * step,
                     what=f"2D {flow_type} field step {step}")
    if flow_type != "passive_scalar":
        got_u = gathered(gpu.velocity_field, 2)
        if rank == 0:
            assert_close(got_u, cpu.velocity_field[(slice(None),) + inner], real_t, factor=1.0 + 0.5 * step,
                         what=f"2D velocity step {step}")
if body is not None:
    body.time_step(dt)
    ref.time_step(dt)
    body.sync_global_mismatch_fields()
    if rank == 0:
        assert_close(body.global_lag_grid_position_mismatch_field, ref.lag_position_mismatch, real_t, factor=6.0,
                     what="2D position mismatch after nodes changed slabs")   # measured: 2.2 units
if flow_type != "passive_scalar":
    mv, want_mv = float(gpu.get_max_vorticity()), float(cpu.get_max_vorticity())
    assert abs(mv - want_mv) <= 1e-5 * abs(want_mv)
world.barrier()
print(f"rank {rank} ok, transport {gpu.mpi_construct.device_context(g).transport}")
