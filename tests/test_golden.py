"""Frozen vectors (tests/golden/*.npz): the oracle must keep reproducing them on the CPU, and
the HIP library must match them on the GPU.  The vectors are oracle outputs on seeded inputs,
not reference outputs (the reference is not importable where they were made)."""
import os

import numpy as np
import pytest

import golden_cases as gc
from helpers import assert_close, make_ctx

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", sorted(gc.CASES))
def test_oracle_reproduces_golden(name):
    want = _load(name)
    got = gc.CASES[name]()
    assert sorted(got) == sorted(want)
    for k in want:
        if want[k].dtype.kind == "i":
            assert np.array_equal(got[k], want[k]), (name, k)
        else:
            # same code on the same seeds: only FFT library / BLAS differences may show
            assert_close(got[k], want[k], np.float64, factor=25, what=f"{name}:{k}")


@pytest.mark.gpu
@pytest.mark.parametrize("real_t", [np.float32, np.float64])
def test_gpu_stencils_match_golden(smi, real_t):
    want = _load("stencils_f64")
    f, h = gc.stencil_inputs(real_t)
    ctx = make_ctx(smi, gc.SHAPE, gc.G, real_t)
    call = smi._lib.call
    df, dh, out = ctx.from_numpy(f), ctx.from_numpy(h), ctx.field(3)
    fac = 1.0  # inputs rounded to f32 move the f64 golden by a few ulp of the inputs
    call("smi_diffusion_flux", ctx.handle, out.ptr, df.ptr, 0.137, 3)
    assert_close(out.get(), want["diffusion_flux"], real_t, factor=fac, what="diffusion flux")
    call("smi_curl", ctx.handle, out.ptr, df.ptr, 0.5)
    assert_close(out.get(), want["curl"], real_t, factor=fac, what="curl")
    dw = ctx.from_numpy(h)
    call("smi_update_vorticity_from_velocity_forcing", ctx.handle, dw.ptr, df.ptr, 0.25)
    assert_close(dw.get(), want["update_vorticity"], real_t, factor=fac, what="update vorticity")
    call("smi_cross_product", ctx.handle, out.ptr, df.ptr, dh.ptr)
    assert_close(out.get(), want["cross"], real_t, factor=fac, what="cross")
    da, scratch = ctx.from_numpy(f), ctx.field()
    call("smi_advection_timestep_eno3", ctx.handle, da.ptr, scratch.ptr, dh.ptr, 0.05, 3)
    assert_close(da.get(), want["advection_timestep"], real_t, factor=fac, what="advection timestep")
    dfl, b1, b2 = ctx.from_numpy(f), ctx.field(), ctx.field()
    call("smi_laplacian_filter", ctx.handle, dfl.ptr, 3, b1.ptr, b2.ptr, 2, 0)
    assert_close(dfl.get(), want["filter_multiplicative_2"], real_t, factor=fac, what="filter")


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["rocfft", "pruned"])
@pytest.mark.parametrize("real_t", [np.float32, np.float64])
def test_gpu_poisson_matches_golden(smi, real_t, backend):
    from sopht_mpi_amd.numeric.eulerian_grid_ops import UnboundedPoissonSolverMPI3D
    from sopht_mpi_amd.utils import MPIConstruct3D

    want = _load("poisson_f64")["solution"]
    n, rhs = gc.poisson_inputs(real_t)
    construct = MPIConstruct3D(*n, real_t=real_t)
    solver = UnboundedPoissonSolverMPI3D(*n, construct, gc.G, x_range=1.0, real_t=real_t, backend=backend)
    ctx = solver.ctx
    drhs, dsol = ctx.from_numpy(rhs), ctx.field(3)
    solver.vector_field_solve(dsol, drhs)
    g = gc.G
    inner = (slice(None),) + (slice(g, -g),) * 3
    assert_close(dsol.get()[inner], want[inner], real_t, factor=2.0, what="Poisson solution")


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("real_t", [np.float32, np.float64])
def test_gpu_flow_steps_match_golden(smi, real_t, fused):
    from sopht_mpi_amd.simulator import UnboundedFlowSimulator3D
    from oracle.flow_3d import FlowSimulator3D

    want = _load("flow_steps_f64")
    sim = UnboundedFlowSimulator3D(real_t=real_t, fused_step=fused, **gc.FLOW_KW)
    shape_src = FlowSimulator3D(real_t=np.float64, **gc.FLOW_KW)  # host coordinates for the seeded input
    sim.vorticity_field[...] = gc.flow_inputs(shape_src, real_t)
    sim.compute_flow_velocity(free_stream_velocity=gc.FLOW_FREE_STREAM)
    for _ in range(2):
        sim.time_step(dt=gc.FLOW_DT, free_stream_velocity=gc.FLOW_FREE_STREAM)
    g = gc.G
    inner = (slice(None),) + (slice(g, -g),) * 3
    assert_close(sim.vorticity_field.get()[inner], want["vorticity"][inner], real_t, factor=1.0, what="vorticity")
    assert_close(sim.velocity_field.get()[inner], want["velocity"][inner], real_t, factor=1.0, what="velocity")


@pytest.mark.gpu
def test_gpu_immersed_boundary_matches_golden(smi):
    from sopht_mpi_amd.numeric.immersed_boundary_ops import EulerianLagrangianGridCommunicatorMPI3D
    from sopht_mpi_amd.utils import MPIConstruct3D

    want = _load("immersed_boundary_f64")
    n, dx, pos, eul, lag = gc.ib_inputs()
    m = pos.shape[1]
    construct = MPIConstruct3D(*n, real_t=np.float64)
    comm = EulerianLagrangianGridCommunicatorMPI3D(dx=dx, eul_grid_coord_shift=dx / 2, interp_kernel_width=2,
                                                   real_t=np.float64, mpi_construct=construct, ghost_size=gc.G,
                                                   n_components=3, interp_kernel_type="cosine")
    ctx = comm.ctx
    dpos = ctx.from_numpy(pos)
    dsup, didx = ctx.zeros((3, 4, 4, 4, m), np.float64), ctx.zeros((3, m), np.int64)
    dwts = ctx.zeros((4, 4, 4, m), np.float64)
    comm.local_eulerian_grid_support_of_lagrangian_grid_kernel(dsup, didx, dpos)
    assert np.array_equal(didx.get(), want["nearest_index"])
    comm.interpolation_weights_kernel(dwts, dsup)
    assert_close(dwts.get(), want["weights"], np.float64, factor=1, what="weights")
    dlag = ctx.zeros((3, m), np.float64)
    comm.eulerian_to_lagrangian_grid_interpolation_kernel(dlag, ctx.from_numpy(eul), dwts, didx)
    assert_close(dlag.get(), want["interpolated"], np.float64, factor=1, what="E->L")
    dfor = ctx.field(3)
    comm.lagrangian_to_eulerian_grid_interpolation_kernel(dfor, ctx.from_numpy(lag), dwts, didx)
    assert_close(dfor.get(), want["spread"], np.float64, factor=1, what="L->E")
