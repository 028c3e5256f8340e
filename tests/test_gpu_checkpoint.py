"""GPU: a flow simulation saved with MPIIO, loaded into a fresh simulator and continued, ends where the
uninterrupted run ends.  The device fields go through the file bit for bit.  The continued run is compared at the
level of rounding, not bit for bit: a checkpoint holds interiors only (mpi_io.py:306-341), so the restarted
simulator starts from zero ghost cells where the running one holds the penalised band and the free stream, and the
stencils next to the boundary read those."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dim", [2, 3])
def test_restart_continues_the_run(smi, tmp_path, dim):
    from sopht_mpi_amd.simulator import UnboundedFlowSimulator2D, UnboundedFlowSimulator3D
    from sopht_mpi_amd.utils import MPIIO, h5lite

    real_t = np.float32
    n = (32, 32, 32)[-dim:]
    kw = dict(grid_size=n, x_range=1.0, kinematic_viscosity=1e-2, flow_type="navier_stokes", real_t=real_t,
              with_free_stream_flow=True)
    sim_cls = UnboundedFlowSimulator3D if dim == 3 else UnboundedFlowSimulator2D
    fs = np.array([1.0, 0.25, -0.5])[:dim]

    def make_io(sim):
        io = MPIIO(mpi_construct=sim.mpi_construct, real_dtype=real_t)
        io.define_eulerian_grid(origin=np.full(dim, sim.dx / 2), dx=np.full(dim, sim.dx), grid_size=np.array(n),
                                ghost_size=sim.ghost_size)
        io.add_as_eulerian_fields_for_io(vorticity=sim.vorticity_field, velocity=sim.velocity_field)
        return io

    def advance(sim, steps):
        for _ in range(steps):
            sim.time_step(dt=sim.compute_stable_timestep(dt_prefac=0.5), free_stream_velocity=fs)

    first = sim_cls(**kw)
    pos = first.position_field
    r2 = sum((pos[k] - 0.5 * (n[-1 - k] / n[-1])) ** 2 for k in range(dim))
    w0 = np.exp(-r2 / (2 * 0.08**2)).astype(real_t)
    if dim == 3:
        first.vorticity_field[...] = np.stack([w0, -0.5 * w0, 0.25 * w0])
        first.compute_flow_velocity(free_stream_velocity=fs)
    else:
        first.vorticity_field[...] = w0
        first.compute_velocity_from_vorticity()
        first.update_velocity_with_free_stream(fs)
    advance(first, 2)
    path = str(tmp_path / "restart.h5")
    make_io(first).save(path, time=first.time)
    g = first.ghost_size
    inner = (slice(g, -g),) * dim
    saved = h5lite.Reader(path)
    name = "Eulerian/Vector/vorticity_0" if dim == 3 else "Eulerian/Scalar/vorticity"
    want = first.vorticity_field.get()
    assert np.array_equal(saved[name][...][0], (want[0] if dim == 3 else want)[inner])
    advance(first, 2)

    second = sim_cls(**kw)
    second.time = make_io(second).load(path)
    assert second.time == pytest.approx(first.time, abs=1.0) and second.time > 0
    advance(second, 2)
    assert second.time == first.time
    from helpers import assert_close

    assert_close(second.vorticity_field.get()[..., g:-g, g:-g], first.vorticity_field.get()[..., g:-g, g:-g], real_t,
                 factor=4.0, what="vorticity two steps after the restart")
    assert_close(second.velocity_field.get()[..., g:-g, g:-g], first.velocity_field.get()[..., g:-g, g:-g], real_t,
                 factor=4.0, what="velocity two steps after the restart")
