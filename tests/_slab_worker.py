"""Worker of the multi-rank GPU test: P ranks share the one GPU of the test box, each owns a
z-slab, transport is host-staged through the socket world.  Rank 0 gathers the slabs and
compares with the single-rank oracle on the global grid (the reference's test strategy:
MPI operator on scattered slabs == serial operator on the gathered array)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sopht-mpi_amd"), os.path.join(ROOT, "tests")]
os.environ["SOPHT_MI_TRANSPORT"] = os.environ.get("SOPHT_MI_TRANSPORT", "host")

from helpers import assert_close  # noqa: E402
from oracle.flow_3d import FlowSimulator3D as OracleSim  # noqa: E402
from sopht_mpi_amd.simulator import (  # noqa: E402
    PrescribedForcingGrid,
    RigidBodyFlowInteractionMPI,
    UnboundedFlowSimulator3D,
)
from sopht_mpi_amd.utils import MPIFieldCommunicator3D  # noqa: E402
from sopht_mpi_amd.world import get_world  # noqa: E402
from oracle import immersed_boundary as oib  # noqa: E402

world = get_world()
rank, size = world.rank, world.size
real_t = np.float32
flow_type = sys.argv[1] if len(sys.argv) > 1 else "navier_stokes"
fused = (sys.argv[2] == "fused") if len(sys.argv) > 2 else True
n = tuple(int(v) for v in sys.argv[3].split("x")) if len(sys.argv) > 3 else (32, 16, 32)
with_filter = "nofilter" not in sys.argv[4:]
# "resync": gather and re-scatter the mismatch state before every interaction, as the reference does on every call
# (VirtualBoundaryForcingMPI.py:243-251); a timing aid, the results are the same
resync_every_call = "resync" in sys.argv[4:]
kw = dict(grid_size=n, x_range=1.0, kinematic_viscosity=1e-2, flow_type=flow_type, real_t=real_t,
          with_free_stream_flow=True, filter_vorticity=with_filter)
if with_filter:
    kw["filter_setting_dict"] = {"order": 1, "type": "multiplicative"}
# "pencil": the reference's default decomposition over z and y (utils/mpi_utils_3d.py:32-48); "yslab": slabs along y
# "xslab" / "yx" / "zx": the decompositions of the reference's operator tests that divide x - (1, 1, 0), (1, 0, 0),
# (0, 1, 0) (tests/test_numeric/test_eulerian_grid_ops/test_stencil_ops_3d/test_diffusion_flux_mpi_3d.py:20-23)
_named = {"pencil": (0, 0, 1), "yslab": (1, 0, 1), "xslab": (1, 1, 0), "yx": (1, 0, 0), "zx": (0, 1, 0)}
rank_distribution = next((d for name, d in _named.items() if name in sys.argv[4:]), None)
gpu = UnboundedFlowSimulator3D(fused_step=fused, rank_distribution=rank_distribution, **kw)
# grids whose sizes have a pruned plan take it; any other size goes through the divided rocFFT transform
_solver = gpu.unbounded_poisson_solver
assert _solver.backend == ("pruned" if _solver.backend_supports("pruned", *n, nranks=size) else "rocfft")
topo = gpu.mpi_construct.grid_topology
assert int(np.prod(topo)) == size
if rank_distribution is None:
    assert list(topo) == [size, 1, 1]
elif rank_distribution == (0, 0, 1):
    assert topo[1] > 1 or size < 4, topo
else:
    assert all(t == 1 for t, aligned in zip(topo, rank_distribution) if aligned), topo
assert list(gpu.mpi_construct.local_grid_size) == [n[0] // topo[0], n[1] // topo[1], n[2] // topo[2]]
g = gpu.ghost_size
comm = MPIFieldCommunicator3D(g, gpu.mpi_construct)
rng = np.random.default_rng(7)
cpu = OracleSim(**kw)
x, y, z = cpu.position_field
w0 = np.zeros(cpu.vorticity_field.shape, dtype=real_t)
for k in range(3):
    c = rng.uniform(0.35, 0.65, 3)
    w0[k] = rng.uniform(-1, 1) * np.exp(-((x - c[0]) ** 2 + (y - c[1] * 0.5) ** 2 + (z - c[2]) ** 2) / (2 * 0.07**2))
inner = (slice(None),) + (slice(g, -g),) * 3
comm.scatter_global_vector_field(gpu.vorticity_field, w0[inner])
cpu.vorticity_field[inner] = w0[inner]
fs = np.array([1.0, 0.25, -0.5])
gpu.compute_flow_velocity(free_stream_velocity=fs)
cpu.compute_flow_velocity(free_stream_velocity=fs)

interactor = None
if flow_type == "navier_stokes_with_forcing":
    theta = np.linspace(0, 2 * np.pi, 60, endpoint=False)
    # a ring that crosses every slab boundary (z extent 0.2 .. 0.8 of the unit z range)
    pos = np.stack([0.5 + 0.2 * np.cos(theta), 0.25 + 0.1 * np.sin(theta), 0.5 + 0.3 * np.sin(theta)])
    interactor = RigidBodyFlowInteractionMPI(
        mpi_construct=gpu.mpi_construct, mpi_ghost_exchange_communicator=gpu.mpi_ghost_exchange_communicator,
        rigid_body=None, eul_grid_forcing_field=gpu.eul_grid_forcing_field,
        eul_grid_velocity_field=gpu.velocity_field, virtual_boundary_stiffness_coeff=-2e2,
        virtual_boundary_damping_coeff=-1e1, dx=gpu.dx, grid_dim=3, forcing_grid_cls=PrescribedForcingGrid,
        positions=pos, max_spacing=float(gpu.dx))
    ref = oib.VirtualBoundaryForcing(-2e2 * float(gpu.dx) ** 2, -1e1 * float(gpu.dx) ** 2, 3, cpu.dx, pos.shape[1], g,
                                     eul_real_t=real_t)

from sopht_mpi_amd import _lib  # noqa: E402

_lib.call("smi_prof_enable", gpu.ctx.handle, 1)
interactor_ms = []
for step in range(4):
    dt = cpu.compute_stable_timestep(dt_prefac=0.5)
    dt_gpu = gpu.compute_stable_timestep(dt_prefac=0.5)
    assert abs(dt_gpu - dt) <= 1e-5 * dt, (dt_gpu, dt)
    if interactor is not None:
        # the ring drifts along z, 0.7 of a cell per step: nodes change slabs between steps (the mismatch state
        # must travel with them), while most calls leave the ownership alone
        pos_now = pos + np.array([0.0, 0.0, 0.7 * step * float(gpu.dx)]).reshape(3, 1)
        if hasattr(interactor.forcing_grid, "reference_positions"):   # the master rank holds the body
            interactor.forcing_grid.reference_positions[...] = pos_now
        interactor.time_step(dt)
        t_host = time.perf_counter()
        interactor.resync_every_call = resync_every_call
        interactor()
        interactor_ms.append(1e3 * (time.perf_counter() - t_host))
        ref.time_step(dt)
        ref.compute_interaction_force_on_eul_and_lag_grid(cpu.eul_grid_forcing_field, cpu.velocity_field, pos_now,
                                                          np.zeros_like(pos))
        if rank == 0:
            assert_close(interactor.global_lag_grid_forcing_field, ref.lag_forcing, np.float32, factor=1.0,
                         what=f"lag forcing step {step}")  # f64 arrays interpolated from the f32 velocity
    gpu.time_step(dt=dt, free_stream_velocity=fs)
    cpu.time_step(dt=dt, free_stream_velocity=fs)
    got_w = np.zeros((3,) + n, dtype=real_t) if rank == 0 else None
    got_u = np.zeros((3,) + n, dtype=real_t) if rank == 0 else None
    comm.gather_local_vector_field(got_w, gpu.vorticity_field)
    comm.gather_local_vector_field(got_u, gpu.velocity_field)
    if rank == 0:
        if os.environ.get("SMI_TEST_SHOW_WORST"):
            err = np.abs(got_w - cpu.vorticity_field[inner])
            order = np.argsort(err.ravel())[::-1][:12]
            print("worst cells (comp, z, y, x):", [tuple(int(v) for v in np.unravel_index(i, err.shape)) for i in order],
                  [float(err.ravel()[i]) for i in order], flush=True)
        assert_close(got_w, cpu.vorticity_field[inner], real_t, factor=1.0 + 0.5 * step, what=f"vorticity step {step}")
        assert_close(got_u, cpu.velocity_field[inner], real_t, factor=1.0 + 0.5 * step, what=f"velocity step {step}")
if interactor is not None:
    # the mismatch state followed the nodes from slab to slab: gathered, it equals the single-rank oracle's
    interactor.time_step(dt)
    ref.time_step(dt)
    interactor.sync_global_mismatch_fields()
    if rank == 0:
        assert_close(interactor.global_lag_grid_position_mismatch_field, ref.lag_position_mismatch, np.float32,
                     what="position mismatch after nodes changed slabs")
        print(f"host ms per interactor() call at {size} ranks: " + " ".join(f"{t:.2f}" for t in interactor_ms))
if fused and (n[2] + 2 * g) % 4 == 0:
    # the one-call step took the one-pass advection + diffusion kernel on every step (slabs and pencils alike)
    launched = _lib.prof_collect(gpu.ctx).get("fused_advect_diffuse", {"launches": 0})["launches"]
    assert launched >= 4, f"one-pass advection + diffusion kernel launched {launched} times in 4 steps"
if flow_type == "navier_stokes":
    # w += p curl(u_pen - u) across the ranks (update_vorticity_from_velocity_forcing_mpi_3d.py:181-330): interiors
    # scattered, both inputs exchanged by the operator, against the oracle on the whole grid
    from oracle import stencils_3d as st3  # noqa: E402
    from sopht_mpi_amd.numeric import eulerian_grid_ops as ops  # noqa: E402

    update = ops.gen_update_vorticity_from_penalised_velocity_pyst_mpi_kernel_3d(
        real_t, gpu.mpi_construct, gpu.mpi_ghost_exchange_communicator)
    glob = rng.standard_normal((3, 3) + n).astype(real_t)   # velocity, penalised velocity, vorticity
    fields = [gpu.ctx.field(3) for _ in range(3)]
    for k in range(3):
        comm.scatter_global_vector_field(fields[k], glob[k] if rank == 0 else None)
    update(vorticity_field=fields[2], penalised_velocity_field=fields[1], velocity_field=fields[0], prefactor=real_t(0.37))
    got = np.zeros((3,) + n, dtype=real_t) if rank == 0 else None
    comm.gather_local_vector_field(got, fields[2])
    if rank == 0:
        padded = [np.zeros((3,) + tuple(v + 2 * g for v in n), dtype=real_t) for _ in range(3)]
        for k in range(3):
            padded[k][inner] = glob[k]
        st3.update_vorticity_from_penalised_velocity(padded[2], padded[1], padded[0], real_t(0.37), g)
        assert_close(got, padded[2][inner], real_t, what="update from penalised velocity across ranks")
mv = gpu.get_max_vorticity()
assert abs(float(mv) - float(cpu.get_max_vorticity())) <= 1e-5 * abs(float(cpu.get_max_vorticity()))
world.barrier()
print(f"rank {rank} ok, transport {gpu.mpi_construct.device_context(g).transport}")
