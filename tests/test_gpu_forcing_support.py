"""The flow step may sweep only the part of the grid that the immersed-boundary forcing can reach
(smi_set_forcing_support, and the record DeviceField keeps of where a field is non-zero).  The boxed
update and reset must give exactly what the full-array operators give, and the record must be
dropped by everything that could write elsewhere."""
import ctypes as C

import numpy as np
import pytest

from helpers import assert_close

pytestmark = pytest.mark.gpu


def _sim(real_t, n=(24, 20, 32), **kw):
    from sopht_mpi_amd.simulator import UnboundedFlowSimulator3D

    return UnboundedFlowSimulator3D(grid_size=n, x_range=1.0, kinematic_viscosity=1e-2,
                                    flow_type="navier_stokes_with_forcing", real_t=real_t,
                                    with_free_stream_flow=True, **kw)


def _blob(sim, real_t):
    x, y, z = np.asarray(sim.position_field)
    w = np.exp(-((x - 0.5) ** 2 + (y - 0.3) ** 2 + (z - 0.35) ** 2) / (2 * 0.08**2))
    return np.stack([w, -0.5 * w, 0.25 * w]).astype(real_t)


@pytest.mark.parametrize("real_t", [np.float32, np.float64])
@pytest.mark.parametrize("box", [((6, 5, 9), (12, 11, 20)), ((0, 0, 0), (3, 4, 5)), ((20, 17, 28), (27, 23, 35)),
                                 ((9, 9, 9), (9, 9, 9)), "empty", ((-3, -2, -1), (40, 40, 50))])
def test_boxed_step_equals_full_step(smi, box, real_t):
    """Direct use of the C entry point: a forcing field that is non-zero inside a box only."""
    fs = [1.0, 0.2, -0.1]
    results = []
    for promise in (False, True):
        sim = _sim(real_t)
        sim.vorticity_field[...] = _blob(sim, real_t)
        sim.compute_flow_velocity(free_stream_velocity=fs)
        shape = sim.eul_grid_forcing_field.shape
        forcing = np.zeros(shape, dtype=real_t)
        if box != "empty":
            lo = [max(v, 0) for v in box[0]]
            hi = [min(v, s - 1) for v, s in zip(box[1], shape[1:])]
            sl = tuple(slice(a, b + 1) for a, b in zip(lo, hi))
            forcing[(slice(None),) + sl] = (np.random.default_rng(6).random((3,) + tuple(
                b - a + 1 for a, b in zip(lo, hi))) - 0.5).astype(real_t)
        sim.eul_grid_forcing_field[...] = forcing  # a host write: the record becomes "unknown"
        assert sim.eul_grid_forcing_field.support is None
        if promise:
            # what a writer that knows its footprint leaves behind
            sim.eul_grid_forcing_field._alloc.support = box
        sim.time_step(dt=2e-3, free_stream_velocity=fs)
        results.append((sim.vorticity_field.get(), sim.velocity_field.get(), sim.eul_grid_forcing_field.get()))
        assert sim.eul_grid_forcing_field.support == "empty"
    (w0, u0, f0), (w1, u1, f1) = results
    assert np.array_equal(w0, w1), "boxed forcing update changed the vorticity"
    assert np.array_equal(u0, u1)
    assert not f0.any() and not f1.any(), "forcing was not reset everywhere it was non-zero"


def test_interactor_loop_with_and_without_the_record(smi):
    """An immersed-boundary loop: with the record kept (spread footprint known) and with it dropped before
    every step, the fields agree; and the record follows the writers."""
    from sopht_mpi_amd.simulator import PrescribedForcingGrid, RigidBodyFlowInteractionMPI

    real_t = np.float32
    fs = [1.0, 0.0, 0.0]
    theta = np.linspace(0, 2 * np.pi, 60, endpoint=False)
    pos = np.stack([0.45 + 0.15 * np.cos(theta), 0.3 + 0.1 * np.sin(theta), 0.35 + 0.05 * np.cos(3 * theta)])
    runs = []
    for keep in (True, False):
        sim = _sim(real_t)
        assert sim.eul_grid_forcing_field.support == "empty"
        inter = RigidBodyFlowInteractionMPI(
            mpi_construct=sim.mpi_construct, mpi_ghost_exchange_communicator=sim.mpi_ghost_exchange_communicator,
            rigid_body=None, eul_grid_forcing_field=sim.eul_grid_forcing_field,
            eul_grid_velocity_field=sim.velocity_field, virtual_boundary_stiffness_coeff=-5e2,
            virtual_boundary_damping_coeff=-5.0, dx=sim.dx, grid_dim=3, forcing_grid_cls=PrescribedForcingGrid,
            positions=pos, max_spacing=float(sim.dx))
        sim.vorticity_field[...] = _blob(sim, real_t)
        sim.compute_flow_velocity(free_stream_velocity=fs)
        for _ in range(4):
            inter()
            support = sim.eul_grid_forcing_field.support
            assert isinstance(support, tuple), support
            f = sim.eul_grid_forcing_field.get()
            assert sim.eul_grid_forcing_field.support == support, "reading a field must not drop its record"
            lo, hi = support
            inside = np.zeros(f.shape[1:], dtype=bool)
            inside[tuple(slice(max(a, 0), b + 1) for a, b in zip(lo, hi))] = True
            assert f.any() and not f[:, ~inside].any(), "forcing outside the recorded footprint"
            if not keep:
                _ = sim.eul_grid_forcing_field.ptr  # anyone holding the raw pointer might write anywhere
                assert sim.eul_grid_forcing_field.support is None
            inter.time_step(dt=2e-3)
            sim.time_step(dt=2e-3, free_stream_velocity=fs)
            assert sim.eul_grid_forcing_field.support == "empty"
            assert not sim.eul_grid_forcing_field.get().any()
        runs.append((sim.vorticity_field.get(), sim.velocity_field.get(),
                     inter.local_lag_grid_position_mismatch_field.get()))
    # not bit for bit: the spread accumulates with floating-point atomics, whose order differs from run to run
    for a, b, what in zip(*runs, ("vorticity", "velocity", "position mismatch")):
        assert_close(a, b, real_t, what=what)  # the Lagrangian arrays are f64 but follow the f32 flow field
    # a view shares the allocation, and with it the record
    sim = _sim(real_t)
    _ = sim.eul_grid_forcing_field[1].ptr
    assert sim.eul_grid_forcing_field.support is None


def test_promise_is_for_one_step_and_one_rank(smi):
    from sopht_mpi_amd._lib import SophtMIError, call

    sim = _sim(np.float32)
    lo, hi = (C.c_int * 3)(2, 2, 2), (C.c_int * 3)(5, 5, 5)
    call("smi_set_forcing_support", sim.ctx.handle, lo, hi)
    call("smi_set_forcing_support", sim.ctx.handle, None, None)  # withdrawn
    with pytest.raises(SophtMIError):
        call("smi_set_forcing_support", sim.ctx.handle, lo, None)


@pytest.mark.parametrize("real_t", [np.float32, np.float64])
@pytest.mark.parametrize("free_stream", [None, [0.7, -0.2, 0.1]])
def test_step_gathers_the_velocity_maximum(smi, real_t, free_stream):
    """compute_stable_timestep after a fused step takes max |u|_1 from the step's curl kernel; it must be the
    value a reduction over the velocity field gives, and any write to the field must send it back there."""
    from sopht_mpi_amd.simulator import UnboundedFlowSimulator3D

    sim = UnboundedFlowSimulator3D(grid_size=(24, 20, 32), x_range=1.0, kinematic_viscosity=1e-3,
                                   flow_type="navier_stokes", real_t=real_t,
                                   with_free_stream_flow=free_stream is not None)
    sim.vorticity_field[...] = 40 * _blob(sim, real_t)
    sim.compute_flow_velocity(free_stream_velocity=free_stream)
    assert "step_max_abs_sum" not in sim.velocity_field._alloc.derived
    for _ in range(3):
        sim.time_step(dt=1e-3, free_stream_velocity=free_stream)
        assert "step_max_abs_sum" in sim.velocity_field._alloc.derived
        dt_fast = sim.compute_stable_timestep()
        assert "step_max_abs_sum" in sim.velocity_field._alloc.derived, "reading must not drop the by-product"
        u = sim.velocity_field.get()
        g = sim.ghost_size
        want = np.max(np.sum(np.abs(u), axis=0)[g:-g, g:-g, g:-g])
        _ = sim.velocity_field.ptr  # drop it: the next call reduces the field
        dt_slow = sim.compute_stable_timestep()
        assert dt_fast == pytest.approx(dt_slow, rel=1e-6 if real_t == np.float32 else 1e-13)
        assert dt_fast == pytest.approx(sim.CFL * sim.dx / (want + (1e-4)), rel=1e-5)
    sim.time_step(dt=1e-3, free_stream_velocity=free_stream)
    sim.velocity_field[...] = 0  # a host write
    dt0 = sim.compute_stable_timestep()
    assert dt0 == pytest.approx(min(sim.CFL * sim.dx / 1e-4, 0.9 * sim.dx**2 / 6 / (1e-3 + 1e-4)), rel=1e-5)


def test_a_promise_that_does_not_hold_is_caught_when_checking_is_on(smi):
    """The contract of smi_set_forcing_support is the caller's to keep; with the context option
    "check_forcing_support" the step verifies the box first and refuses one that does not hold."""
    from sopht_mpi_amd._lib import SophtMIError, call

    real_t = np.float32
    sim = _sim(real_t)
    fs = [1.0, 0.0, 0.0]
    sim.vorticity_field[...] = _blob(sim, real_t)
    sim.compute_flow_velocity(free_stream_velocity=fs)
    forcing = np.zeros(sim.eul_grid_forcing_field.shape, dtype=real_t)
    forcing[:, 8:12, 8:12, 8:12] = 0.25
    forcing[1, 20, 3, 30] = 1e-3                       # one cell outside the box about to be promised
    sim.eul_grid_forcing_field[...] = forcing
    box = ((7, 7, 7), (12, 12, 12))
    lo, hi = (C.c_int * 3)(*box[0]), (C.c_int * 3)(*box[1])
    call("smi_set_forcing_support", sim.ctx.handle, lo, hi)
    outside = C.c_longlong(-1)
    call("smi_check_forcing_support", sim.ctx.handle, sim.eul_grid_forcing_field.tracked_ptr(), C.byref(outside))
    assert outside.value == 1
    call("smi_ctx_set_option", sim.ctx.handle, b"check_forcing_support", 1)
    sim.eul_grid_forcing_field._alloc.support = box
    with pytest.raises(SophtMIError, match="outside the box"):
        sim.time_step(dt=2e-3, free_stream_velocity=fs)
    # the same box with the stray cell removed passes the check and steps like the full sweep
    forcing[1, 20, 3, 30] = 0
    sim.eul_grid_forcing_field[...] = forcing
    sim.eul_grid_forcing_field._alloc.support = box
    sim.time_step(dt=2e-3, free_stream_velocity=fs)
    assert not sim.eul_grid_forcing_field.get().any()
