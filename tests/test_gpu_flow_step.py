"""GPU parity: whole flow steps of UnboundedFlowSimulator3D against the oracle, through the
reference's own simulator API (operator-by-operator and fused), several steps in a row."""
import numpy as np
import pytest

from helpers import assert_close, assert_pointwise
from oracle.flow_3d import FlowSimulator3D as OracleSim

pytestmark = pytest.mark.gpu


def _blob_vorticity(sim_shape, pos, rng, real_t):
    """Smooth compact vorticity (a few Gaussian blobs), zero near the boundaries."""
    x, y, z = pos
    w = np.zeros((3,) + sim_shape, dtype=np.float64)
    for _ in range(4):
        c = rng.uniform(0.35, 0.65, 3)
        amp = rng.uniform(-1, 1, 3)
        r2 = (x - c[0]) ** 2 + (y - c[1] * y.max()) ** 2 + (z - c[2] * z.max()) ** 2
        for k in range(3):
            w[k] += amp[k] * np.exp(-r2 / (2 * 0.06**2))
    return w.astype(real_t)


def _pair(smi, n, flow_type, real_t, **kw):
    from sopht_mpi_amd.simulator import UnboundedFlowSimulator3D

    common = dict(grid_size=n, x_range=1.0, kinematic_viscosity=1e-2, flow_type=flow_type, real_t=real_t,
                  ghost_size=kw.pop("ghost_size", 2))
    fused = kw.pop("fused_step", True)
    backend = kw.pop("poisson_backend", None)
    gpu = UnboundedFlowSimulator3D(fused_step=fused, poisson_backend=backend, **common, **kw)
    cpu = OracleSim(**common, **kw)
    return gpu, cpu


NS_CASES = [
    ((16, 16, 16), "navier_stokes", False, False),
    ((16, 16, 32), "navier_stokes", True, False),
    ((32, 32, 32), "navier_stokes_with_forcing", True, False),
    ((16, 16, 16), "navier_stokes_with_forcing", True, True),
]


def _ns_params():
    from sopht_mpi_amd._lib import lib

    out = []
    for backend, code in (("rocfft", 0), ("pruned", 1)):
        for n, flow_type, free_stream, filt in NS_CASES:
            marks = [] if lib.smi_poisson_supports(n[0], n[1], n[2], 1, code) else [
                pytest.mark.skip(reason=f"smi_poisson_supports: no {backend} plan for grid {n}")]
            out.append(pytest.param(n, flow_type, free_stream, filt, backend, marks=marks))
    return out


@pytest.mark.parametrize("filt", [False, True])
@pytest.mark.parametrize("ghost,width", [(2, 0), (2, 1), (2, 2), (2, 3), (3, 1), (3, 2)])
def test_one_call_step_with_other_penalty_zones(smi, ghost, width, filt):
    """Inside the one-call step the x bands of the penalisation are applied in the store of the kernel that writes
    the new vorticity when ghost + zone width = 4 (a band is one 16-byte vector), and by the operator's own x pass
    otherwise: both against the oracle, with the one-pass filter and without, whole arrays including ghosts."""
    real_t, n = np.float32, (24 if ghost == 3 else 16, 16, 32)
    rng = np.random.default_rng(3)
    kw = dict(with_free_stream_flow=True, filter_vorticity=filt, ghost_size=ghost, penalty_zone_width=width)
    if filt:
        kw["filter_setting_dict"] = {"order": 1, "type": "multiplicative"}
    gpu, cpu = _pair(smi, n, "navier_stokes", real_t, fused_step=True, poisson_backend="rocfft", **kw)
    # vorticity that reaches the bands, so that the penalisation has something to act on
    w0 = (0.2 * rng.standard_normal(cpu.vorticity_field.shape)).astype(real_t)
    gpu.vorticity_field[...] = w0
    cpu.vorticity_field[...] = w0
    fs = np.array([0.5, -0.25, 1.0])
    gpu.compute_flow_velocity(free_stream_velocity=fs)
    cpu.compute_flow_velocity(free_stream_velocity=fs)
    for step in range(2):
        dt = cpu.compute_stable_timestep(dt_prefac=0.25)
        gpu.time_step(dt=dt, free_stream_velocity=fs)
        cpu.time_step(dt=dt, free_stream_velocity=fs)
        assert_close(gpu.vorticity_field.get(), cpu.vorticity_field, real_t, factor=1.0 + 0.5 * step,
                     what=f"vorticity incl. ghosts, ghost {ghost} width {width} step {step}")
        assert_close(gpu.velocity_field.get(), cpu.velocity_field, real_t, factor=2.0 + 0.5 * step,
                     what=f"velocity incl. ghosts, ghost {ghost} width {width} step {step}")


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("real_t", [np.float32, np.float64])
@pytest.mark.parametrize("n,flow_type,free_stream,filt,backend", _ns_params())
def test_navier_stokes_steps(smi, n, flow_type, free_stream, filt, real_t, fused, backend):
    rng = np.random.default_rng(11)
    kw = dict(with_free_stream_flow=free_stream, filter_vorticity=filt)
    if filt:
        kw["filter_setting_dict"] = {"order": 1, "type": "multiplicative"}
    gpu, cpu = _pair(smi, n, flow_type, real_t, fused_step=fused, poisson_backend=backend, **kw)
    assert gpu.unbounded_poisson_solver.backend == backend
    w0 = _blob_vorticity(cpu.vorticity_field.shape[1:], cpu.position_field, rng, real_t)
    gpu.vorticity_field[...] = w0
    cpu.vorticity_field[...] = w0
    fs = np.array([1.0, 0.5, -0.25])
    gpu.compute_flow_velocity(free_stream_velocity=fs)
    cpu.compute_flow_velocity(free_stream_velocity=fs)
    assert_close(gpu.velocity_field.get(), cpu.velocity_field, real_t, factor=2.0, what="initial velocity")
    smi._lib.call("smi_prof_enable", gpu.ctx.handle, 1)
    smi._lib.call("smi_prof_reset", gpu.ctx.handle)
    for step in range(3):
        dt_gpu = gpu.compute_stable_timestep(dt_prefac=0.5)
        dt_cpu = cpu.compute_stable_timestep(dt_prefac=0.5)
        assert abs(dt_gpu - dt_cpu) <= 1e-5 * dt_cpu
        if flow_type == "navier_stokes_with_forcing":
            forcing = (rng.random(cpu.velocity_field.shape) - 0.5).astype(real_t) * real_t(0.1)
            gpu.eul_grid_forcing_field[...] = forcing
            cpu.eul_grid_forcing_field[...] = forcing
        gpu.time_step(dt=dt_cpu, free_stream_velocity=fs)
        cpu.time_step(dt=dt_cpu, free_stream_velocity=fs)
        g = gpu.ghost_size
        inner = (slice(None),) + (slice(g, -g),) * 3
        # vorticity to 2e-6 of its maximum (a fifth of the north-star 1e-5) and pointwise to rtol 1e-5; errors of the steps add up
        assert_close(gpu.vorticity_field.get()[inner], cpu.vorticity_field[inner], real_t, factor=1.0 + 0.5 * step,
                     what=f"vorticity step {step}")
        assert_pointwise(gpu.vorticity_field.get()[inner], cpu.vorticity_field[inner], real_t, what=f"vorticity step {step}")
        assert_close(gpu.velocity_field.get()[inner], cpu.velocity_field[inner], real_t, factor=2.0 + 0.5 * step,
                     what=f"velocity step {step}")
        assert_close(gpu.vorticity_field.get(), cpu.vorticity_field, real_t, factor=1.0 + 0.5 * step,
                     what=f"vorticity incl. ghosts step {step}")
    assert gpu.time == cpu.time
    slots = smi._lib.prof_collect(gpu.ctx)
    smi._lib.call("smi_prof_enable", gpu.ctx.handle, 0)
    if fused and gpu.ctx.padded_shape[2] % 4 == 0:
        # both precisions take advection and diffusion through the one-pass kernel
        assert slots.get("fused_advect_diffuse", {}).get("launches", 0) == 3, sorted(slots)
    if flow_type == "navier_stokes_with_forcing":
        assert not np.any(gpu.eul_grid_forcing_field.get())
    assert abs(float(gpu.get_max_vorticity()) - float(cpu.get_max_vorticity())) <= 1e-5 * abs(
        float(cpu.get_max_vorticity()))
    d_gpu, d_cpu = gpu.get_vorticity_divergence_l2_norm(), cpu.get_vorticity_divergence_l2_norm()
    assert abs(d_gpu - d_cpu) <= 1e-4 * max(d_cpu, 1e-12)


@pytest.mark.parametrize("real_t", [np.float32, np.float64])
@pytest.mark.parametrize("flow_type", ["passive_scalar", "passive_vector"])
def test_passive_transport_steps(smi, flow_type, real_t):
    """flow_simulators_mpi_3d.py:356-380: advect with a prescribed velocity, then diffuse."""
    rng = np.random.default_rng(5)
    gpu, cpu = _pair(smi, (16, 16, 32), flow_type, real_t)
    pshape = cpu.velocity_field.shape[1:]
    w0 = _blob_vorticity(pshape, cpu.position_field, rng, real_t)
    field0 = w0[0] if flow_type == "passive_scalar" else w0
    vel = (rng.random((3,) + pshape) - 0.5).astype(real_t)
    gpu.velocity_field[...] = vel
    cpu.velocity_field[...] = vel
    name = "primary_scalar_field" if flow_type == "passive_scalar" else "primary_vector_field"
    getattr(gpu, name)[...] = field0
    getattr(cpu, name)[...] = field0
    for step in range(3):
        dt = cpu.compute_stable_timestep(dt_prefac=0.5)
        assert abs(gpu.compute_stable_timestep(dt_prefac=0.5) - dt) <= 1e-5 * dt
        gpu.time_step(dt=dt)
        cpu.time_step(dt=dt)
        assert_close(getattr(gpu, name).get(), getattr(cpu, name), real_t, factor=1.0 + 0.5 * step,
                     what=f"{flow_type} step {step}")
    assert gpu.time == cpu.time


def test_simulator_known_answers(smi):
    """The reference's own closed-form checks (tests/test_simulator/test_flow/
    test_flow_simulators_3d.py:333-360, 441-500) run against the GPU simulator."""
    from sopht_mpi_amd.simulator import UnboundedFlowSimulator3D
    from sopht_mpi_amd.sopht_compat import get_test_tol

    for precision, real_t in (("single", np.float32), ("double", np.float64)):
        n, cfl, nu = 8, 0.2, 1.0
        dx = real_t(1.0 / n)
        sim = UnboundedFlowSimulator3D(grid_size=(n, n, n), x_range=1.0, CFL=cfl, kinematic_viscosity=nu,
                                       real_t=real_t)
        sim.velocity_field[...] = 2.0
        got = sim.compute_stable_timestep(dt_prefac=0.5, precision=precision)
        adv = cfl * dx / (3 * 2.0 + get_test_tol(precision))
        dif = 0.9 * dx**2 / 6 / (nu + get_test_tol(precision))
        assert got == 0.5 * min(adv, dif)
        sim = UnboundedFlowSimulator3D(grid_size=(n, n, n), x_range=1.0, kinematic_viscosity=1e-2, real_t=real_t,
                                       flow_type="navier_stokes_with_forcing")
        w = np.random.default_rng(0).random(sim.vorticity_field.shape).astype(real_t)
        w[0, 6, 6, 6] = 10.0
        sim.vorticity_field[...] = w
        assert sim.get_max_vorticity() == 10.0


def test_invalid_arguments(smi):
    from sopht_mpi_amd.simulator import UnboundedFlowSimulator3D

    with pytest.raises(ValueError):
        UnboundedFlowSimulator3D(grid_size=(8, 8, 8), x_range=1.0, kinematic_viscosity=1.0, flow_type="bogus")
    with pytest.raises(ValueError):  # ghost size smaller than the ENO3 support
        UnboundedFlowSimulator3D(grid_size=(8, 8, 8), x_range=1.0, kinematic_viscosity=1.0, ghost_size=1)


@pytest.mark.parametrize("dim", [2, 3])
def test_named_step_callables(smi, dim):
    """The reference exposes each flow family's step under its own name next to ``time_step``
    (flow_simulators_mpi_3d.py:335-424, _2d.py:242-298): here they run the same stage lists."""
    from sopht_mpi_amd.simulator import UnboundedFlowSimulator2D, UnboundedFlowSimulator3D

    real_t = np.float32
    n = (16, 24, 32)[-dim:]
    cls = UnboundedFlowSimulator3D if dim == 3 else UnboundedFlowSimulator2D
    kw = dict(grid_size=n, x_range=1.0, kinematic_viscosity=1e-2, real_t=real_t, with_free_stream_flow=True)
    rng = np.random.default_rng(12)
    fs = np.array([1.0, -0.5, 0.25])[:dim]
    forced = [cls(flow_type="navier_stokes_with_forcing", **kw) for _ in range(3)]
    plain = cls(flow_type="navier_stokes", fused_step=False, **kw)
    shape = forced[0].vorticity_field.shape
    w0 = (rng.random(shape) - 0.5).astype(real_t)
    f0 = (rng.random(forced[0].eul_grid_forcing_field.shape) - 0.5).astype(real_t) * real_t(0.1)
    for sim in forced + [plain]:
        sim.vorticity_field[...] = w0
        if dim == 3:
            sim.compute_flow_velocity(free_stream_velocity=fs)
        else:
            sim.compute_velocity_from_vorticity()
            sim.update_velocity_with_free_stream(fs)
    for sim in forced:
        sim.eul_grid_forcing_field[...] = f0
    dt = 0.5 * forced[0].compute_stable_timestep()
    forced[0].time_step(dt=dt, free_stream_velocity=fs)
    forced[1].navier_stokes_with_forcing_timestep(dt=dt, free_stream_velocity=fs)
    assert forced[0].time == dt and forced[1].time == 0.0          # only time_step moves the clock
    assert np.array_equal(forced[0].vorticity_field.get(), forced[1].vorticity_field.get())
    assert np.array_equal(forced[0].velocity_field.get(), forced[1].velocity_field.get())
    assert not forced[1].eul_grid_forcing_field.get().any()
    # the plain step of a simulator with forcing leaves the forcing alone and equals a simulator without forcing
    forced[2].navier_stokes_timestep(dt=dt, free_stream_velocity=fs)
    plain.flow_time_step(dt=dt, free_stream_velocity=fs)
    assert np.array_equal(forced[2].eul_grid_forcing_field.get(), f0)
    assert_close(forced[2].vorticity_field.get(), plain.vorticity_field.get(), real_t, factor=1.0,
                 what="plain step of a forced simulator")
    if dim == 3:
        assert forced[2].rotational_form_navier_stokes_timestep is not None
    transport = cls(flow_type="passive_scalar", **dict(kw, with_free_stream_flow=False))
    name = "scalar_advection_and_diffusion_timestep" if dim == 3 else "advection_and_diffusion_timestep"
    assert callable(getattr(transport, name))
