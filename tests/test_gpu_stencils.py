"""GPU parity: every 3D stencil / elementwise operator of libsophtmi against the oracle on
the same seeded ghosted arrays (whole array compared, ghost cells included)."""
import ctypes as C

import numpy as np
import pytest

from helpers import assert_close, make_ctx, rand_field
from oracle import stencils_3d as st

pytestmark = pytest.mark.gpu

CASES = [
    # (nz, ny, nx), ghost
    ((8, 8, 8), 2),
    ((8, 12, 16), 2),   # aspect ratio (1, 1.5, 2) of the reference tests
    ((8, 8, 8), 1),
    ((10, 9, 11), 2),   # px not a multiple of 4: scalar path
    ((12, 20, 36), 3),
    ((36, 20, 300), 2),  # several LDS tiles along x (the last one partial), along y and three z chunks
]


def _random_shapes():
    """Seeded shapes that move the tile edges of the LDS-tiled kernels around: padded rows of 4 to
    ~330 cells (one to three x tiles, partial last tiles and lanes), 1 to 6 tile rows, one to
    three z chunks, ghost sizes 1 to 3; every padded row length is a multiple of 4 so that the
    16-byte path runs."""
    rng = np.random.default_rng(2024)
    out = []
    for _ in range(14):
        g = int(rng.integers(1, 4))
        px = 4 * int(rng.integers(2, 84))
        nx = px - 2 * g
        ny = int(rng.integers(2 * g, 44))
        nz = int(rng.integers(2 * g, 40))
        if nx >= 2 * g:
            out.append(((nz, ny, nx), g))
    return out


RANDOM_CASES = _random_shapes()


def _call(smi, name, *a):
    smi._lib.call(name, *a)


@pytest.mark.parametrize("real_t", [np.float32, np.float64])
@pytest.mark.parametrize("shape,g", CASES)
def test_elementwise_ops(smi, shape, g, real_t):
    rng = np.random.default_rng(1)
    ctx = make_ctx(smi, shape, g, real_t)
    pshape = ctx.padded_shape
    a = rand_field(rng, (3,) + pshape, real_t)
    b = rand_field(rng, (3,) + pshape, real_t)
    da, db, dr = ctx.from_numpy(a), ctx.from_numpy(b), ctx.field(3)
    _call(smi, "smi_cross_product", ctx.handle, dr.ptr, da.ptr, db.ptr)
    want = np.zeros_like(a)
    st.elementwise_cross_product(want, a, b)
    assert_close(dr.get(), want, real_t, what="cross")
    _call(smi, "smi_elementwise_sum", ctx.handle, dr.ptr, da.ptr, db.ptr, 3)
    assert_close(dr.get(), a + b, real_t, what="sum")
    _call(smi, "smi_elementwise_saxpby", ctx.handle, dr.ptr, da.ptr, 1.0, db.ptr, -1.0, 3)
    assert_close(dr.get(), a - b, real_t, what="saxpby")
    _call(smi, "smi_elementwise_copy", ctx.handle, dr.ptr, da.ptr, 3)
    assert np.array_equal(dr.get(), a)
    vals = smi._lib.darray([1.0, -2.0, 0.5])
    _call(smi, "smi_add_fixed_val", ctx.handle, dr.ptr, da.ptr, 3, vals)
    assert_close(dr.get(), a + np.array([1.0, -2.0, 0.5], dtype=real_t).reshape(3, 1, 1, 1), real_t, what="add")
    _call(smi, "smi_set_fixed_val", ctx.handle, dr.ptr, 3, vals)
    assert np.array_equal(dr.get(), np.broadcast_to(np.array([1.0, -2.0, 0.5], dtype=real_t).reshape(3, 1, 1, 1),
                                                    a.shape))
    # a component view (odd alignment on some shapes) must work as well
    _call(smi, "smi_elementwise_sum", ctx.handle, dr[1].ptr, da[1].ptr, db[2].ptr, 1)
    assert_close(dr[1].get(), a[1] + b[2], real_t, what="component sum")


@pytest.mark.parametrize("real_t", [np.float32, np.float64])
@pytest.mark.parametrize("shape,g", CASES)
def test_diffusion_flux_and_timestep(smi, shape, g, real_t):
    rng = np.random.default_rng(2)
    ctx = make_ctx(smi, shape, g, real_t)
    f = rand_field(rng, (3,) + ctx.padded_shape, real_t)
    c = real_t(0.137)
    # flux: stale garbage in the output must not survive
    dflux = ctx.from_numpy(rand_field(rng, (3,) + ctx.padded_shape, real_t))
    df = ctx.from_numpy(f)
    _call(smi, "smi_diffusion_flux", ctx.handle, dflux.ptr, df.ptr, float(c), 3)
    want = np.zeros_like(f)
    for k in range(3):
        st.diffusion_flux(want[k], f[k], c, g)
    assert_close(dflux.get(), want, real_t, what="diffusion flux")
    # timestep with one scalar scratch
    dscr = ctx.field()
    _call(smi, "smi_diffusion_timestep", ctx.handle, df.ptr, dscr.ptr, float(c), 3)
    wf, wscr = f.copy(), np.zeros(ctx.padded_shape, dtype=real_t)
    st.vector_diffusion_timestep(wf, wscr, c, g)
    assert_close(df.get(), wf, real_t, factor=2.0, what="diffusion timestep")
    assert_close(dscr.get(), wscr, real_t, scale=np.abs(wf).max(), factor=2.0, what="flux scratch")


@pytest.mark.parametrize("real_t", [np.float32, np.float64])
@pytest.mark.parametrize("shape,g", CASES)
def test_penalisation_method_operators(smi, shape, g, real_t):
    """gen_update_vorticity_from_penalised_velocity_pyst_mpi_kernel_3d, gen_brinkmann_penalise_pyst_mpi_kernel_3d and
    gen_char_func_from_level_set_via_sine_heaviside_pyst_mpi_kernel_3d through the package's generators, as the
    reference's tests call them (test_update_vorticity_from_velocity_forcing_mpi_3d.py:142-260,
    test_brinkmann_penalise_mpi_3d.py, test_char_func_from_level_set_mpi_3d.py)."""
    from sopht_mpi_amd.numeric import eulerian_grid_ops as ops
    from sopht_mpi_amd.utils import MPIConstruct3D, MPIGhostCommunicator3D
    from sopht_mpi_amd.world import SingleWorld

    rng = np.random.default_rng(17)
    construct = MPIConstruct3D(*shape, real_t=real_t, world=SingleWorld())
    ghost_comm = MPIGhostCommunicator3D(ghost_size=g, mpi_construct=construct)
    ctx = ghost_comm.ctx
    full = (3,) + ctx.padded_shape
    u, up, w = (rand_field(rng, full, real_t) for _ in range(3))
    p = real_t(0.41)
    update = ops.gen_update_vorticity_from_penalised_velocity_pyst_mpi_kernel_3d(real_t, construct, ghost_comm)
    assert update.kernel_support == 1
    du, dup, dw = ctx.from_numpy(u), ctx.from_numpy(up), ctx.from_numpy(w)
    update(vorticity_field=dw, penalised_velocity_field=dup, velocity_field=du, prefactor=p)
    want = w.copy()
    st.update_vorticity_from_penalised_velocity(want, up, u, p, g)
    assert_close(dw.get(), want, real_t, what="update from penalised velocity")
    untouched = want == w
    assert np.array_equal(dw.get()[untouched], w[untouched])

    chi = rng.random(ctx.padded_shape).astype(real_t)
    for field_type, shape_f in (("scalar", ctx.padded_shape), ("vector", full)):
        brinkmann = ops.gen_brinkmann_penalise_pyst_mpi_kernel_3d(real_t=real_t, field_type=field_type)
        assert brinkmann.kernel_support == 0
        f, pen = rand_field(rng, shape_f, real_t), rand_field(rng, shape_f, real_t)
        out = ctx.from_numpy(np.ones(shape_f, dtype=real_t))
        names = ("penalised_field", "penalty_field", "field") if field_type == "scalar" else (
            "penalised_vector_field", "penalty_vector_field", "vector_field")
        brinkmann(**{names[0]: out, "penalty_factor": real_t(0.1), "char_field": ctx.from_numpy(chi),
                     names[1]: ctx.from_numpy(pen), names[2]: ctx.from_numpy(f)})
        want = np.ones(shape_f, dtype=real_t)
        st.brinkmann_penalise(want, real_t(0.1), chi, pen, f)
        assert_close(out.get(), want, real_t, what=f"brinkmann penalisation ({field_type})")
    with pytest.raises(ValueError):
        ops.gen_brinkmann_penalise_pyst_mpi_kernel_3d(real_t=real_t, field_type="tensor")

    heaviside = ops.gen_char_func_from_level_set_via_sine_heaviside_pyst_mpi_kernel_3d(blend_width=real_t(0.2),
                                                                                        real_t=real_t)
    assert heaviside.kernel_support == 0
    phi = (rng.random(ctx.padded_shape) - 0.5).astype(real_t)   # both plateaus and the blend zone
    out = ctx.field()
    heaviside(char_func_field=out, level_set_field=ctx.from_numpy(phi))
    want = np.zeros_like(phi)
    st.char_func_from_level_set_via_sine_heaviside(want, phi, real_t(0.2))
    assert_close(out.get(), want, real_t, scale=1.0, factor=4.0, what="sine Heaviside characteristic function")
    got = out.get()
    # (rounding at the ends of the blend zone may leave the plateau values by a few units in the last place)
    assert np.all(got[phi < -0.2] == 0) and np.all(got[phi > 0.2] == 1) and got.min() >= -1e-6 and got.max() <= 1 + 1e-6


@pytest.mark.parametrize("real_t", [np.float32, np.float64])
@pytest.mark.parametrize("shape,g", CASES)
def test_curl_update_divergence(smi, shape, g, real_t):
    rng = np.random.default_rng(3)
    ctx = make_ctx(smi, shape, g, real_t)
    f = rand_field(rng, (3,) + ctx.padded_shape, real_t)
    w = rand_field(rng, (3,) + ctx.padded_shape, real_t)
    p = real_t(0.73)
    df, dw = ctx.from_numpy(f), ctx.from_numpy(w)
    _call(smi, "smi_update_vorticity_from_velocity_forcing", ctx.handle, dw.ptr, df.ptr, float(p))
    want = w.copy()
    st.update_vorticity_from_velocity_forcing(want, f, p, g)
    assert_close(dw.get(), want, real_t, what="update vorticity")
    # cells outside the reference's seven boxes must be bit-identical to the input
    untouched = want == w
    assert np.array_equal(dw.get()[untouched], w[untouched])

    dc = ctx.from_numpy(rand_field(rng, (3,) + ctx.padded_shape, real_t))
    _call(smi, "smi_curl", ctx.handle, dc.ptr, df.ptr, float(p))
    wc = np.zeros_like(f)
    st.curl(wc, f, p, g)
    assert_close(dc.get(), wc, real_t, what="curl")

    dd = ctx.from_numpy(rand_field(rng, ctx.padded_shape, real_t))
    inv_dx = real_t(8.0)
    _call(smi, "smi_divergence", ctx.handle, dd.ptr, df.ptr, float(inv_dx))
    wd = np.zeros(ctx.padded_shape, dtype=real_t)
    st.divergence(wd, f, inv_dx, g)
    assert_close(dd.get(), wd, real_t, what="divergence")


@pytest.mark.parametrize("real_t", [np.float32, np.float64])
@pytest.mark.parametrize("width", [0, 1, 2, 3])
@pytest.mark.parametrize("shape,g", [((12, 12, 12), 2), ((12, 14, 16), 1), ((16, 12, 20), 2)])
def test_penalise_boundary(smi, shape, g, width, real_t):
    from oracle.flow_3d import local_coordinates
    from sopht_mpi_amd.numeric.eulerian_grid_ops import penalty_factor_tables

    rng = np.random.default_rng(4)
    ctx = make_ctx(smi, shape, g, real_t)
    dx = real_t(1.0 / shape[2])
    zc, yc, xc = (local_coordinates(n, 0, dx, g, real_t) for n in shape)
    f = rand_field(rng, (3,) + ctx.padded_shape, real_t)
    df = ctx.from_numpy(f)
    if width == 0:
        _call(smi, "smi_penalise_boundary", ctx.handle, df.ptr, 3, 0, None, None)
        assert np.array_equal(df.get(), f)
        return
    pos = np.flipud(np.array(np.meshgrid(zc, yc, xc, indexing="ij")))
    lo, hi = penalty_factor_tables(width, dx, pos[0], pos[1], pos[2], g, real_t)
    _call(smi, "smi_penalise_boundary", ctx.handle, df.ptr, 3, width, smi._lib.darray(lo), smi._lib.darray(hi))
    want = f.copy()
    st.penalise_vector_field_boundary(want, width, dx, xc, yc, zc, g)
    assert_close(df.get(), want, real_t, what="penalise")


@pytest.mark.parametrize("real_t", [np.float32, np.float64])
@pytest.mark.parametrize("filter_type", ["multiplicative", "convolution"])
@pytest.mark.parametrize("order", [1, 2, 3])
@pytest.mark.parametrize("shape,g", [((8, 8, 8), 2), ((8, 12, 16), 1)])
def test_laplacian_filter(smi, shape, g, order, filter_type, real_t):
    rng = np.random.default_rng(5)
    ctx = make_ctx(smi, shape, g, real_t)
    f = rand_field(rng, (3,) + ctx.padded_shape, real_t)
    df = ctx.from_numpy(f)
    dflux = ctx.from_numpy(rand_field(rng, ctx.padded_shape, real_t))
    dbuf = ctx.from_numpy(rand_field(rng, ctx.padded_shape, real_t))
    code = {"multiplicative": 0, "convolution": 1}[filter_type]
    _call(smi, "smi_laplacian_filter", ctx.handle, df.ptr, 3, dflux.ptr, dbuf.ptr, order, code)
    want = f.copy()
    st.vector_laplacian_filter(want, np.zeros(ctx.padded_shape, real_t), np.zeros(ctx.padded_shape, real_t), order, g,
                               filter_type)
    assert_close(df.get(), want, real_t, what="filter")


@pytest.mark.parametrize("filter_type", ["multiplicative", "convolution"])
@pytest.mark.parametrize("order,width", [(1, 2), (2, 2), (1, 3), (0, 2)])
def test_laplacian_filter_boundary_width(smi, order, width, filter_type):
    """filter_flux_buffer_boundary_width other than the default 1 (laplacian_filter_mpi_3d.py:110-143), through the
    generator a simulation script would call."""
    from sopht_mpi_amd.numeric.eulerian_grid_ops import gen_laplacian_filter_mpi_kernel_3d
    from sopht_mpi_amd.utils import MPIConstruct3D, MPIGhostCommunicator3D

    real_t, shape, g = np.float32, (12, 16, 20), 2
    rng = np.random.default_rng(15)
    construct = MPIConstruct3D(*shape, real_t=real_t)
    ghost = MPIGhostCommunicator3D(ghost_size=g, mpi_construct=construct)
    ctx = ghost.ctx
    f = rand_field(rng, (3,) + ctx.padded_shape, real_t)
    flux0 = rand_field(rng, ctx.padded_shape, real_t)
    df, dflux, dbuf = ctx.from_numpy(f), ctx.from_numpy(flux0), ctx.from_numpy(rand_field(rng, ctx.padded_shape, real_t))
    apply = gen_laplacian_filter_mpi_kernel_3d(construct, ghost, order, dflux, dbuf, real_t, field_type="vector",
                                               filter_type=filter_type, filter_flux_buffer_boundary_width=width)
    apply(vector_field=df)
    want = f.copy()
    st.vector_laplacian_filter(want, flux0.copy(), np.zeros(ctx.padded_shape, real_t), order, g, filter_type,
                               boundary_width=width)
    assert_close(df.get(), want, real_t, what="filter with boundary width")
    with pytest.raises(ValueError):
        gen_laplacian_filter_mpi_kernel_3d(construct, ghost, 1, dflux, dbuf, real_t, filter_flux_buffer_boundary_width=0)


@pytest.mark.parametrize("real_t", [np.float32, np.float64])
def test_reductions(smi, real_t):
    rng = np.random.default_rng(6)
    shape, g = (8, 12, 16), 2
    ctx = make_ctx(smi, shape, g, real_t)
    u = rand_field(rng, (3,) + ctx.padded_shape, real_t)
    u[:, 0, 0, 0] = 100.0  # ghost cells must not count
    du = ctx.from_numpy(u)
    inner = (slice(None),) + (slice(g, -g),) * 3
    out = C.c_double()
    _call(smi, "smi_reduce_max_abs_sum", ctx.handle, du.ptr, C.byref(out))
    assert real_t(out.value) == np.amax(np.sum(np.fabs(u), axis=0)[inner[1:]])
    _call(smi, "smi_reduce_max", ctx.handle, du.ptr, 3, C.byref(out))
    assert real_t(out.value) == np.amax(u[inner])
    _call(smi, "smi_reduce_sum_sq", ctx.handle, du[0].ptr, 1, C.byref(out))
    want = np.sum(u[0][inner[1:]].astype(np.float64) ** 2)
    assert abs(out.value - want) <= 1e-12 * want


def test_argument_errors(smi):
    """Error behaviour of the boundary: bad arguments come back as exceptions, not faults."""
    from sopht_mpi_amd._lib import SophtMIError
    from sopht_mpi_amd.device import DeviceContext

    with pytest.raises(SophtMIError):
        DeviceContext((8, 8, 8), 0, np.float32)
    ctx = make_ctx(smi, (8, 8, 8), 2, np.float32)
    f = ctx.field(3)
    with pytest.raises(SophtMIError):
        smi._lib.call("smi_curl", ctx.handle, f.ptr, f.ptr, 1.0)  # aliasing
    with pytest.raises(SophtMIError):
        smi._lib.call("smi_laplacian_filter", ctx.handle, f.ptr, 3, f[0].ptr, f[1].ptr, 1, 7)


@pytest.mark.parametrize("real_t", [np.float32, np.float64])
@pytest.mark.parametrize("ncomp", [1, 3])
@pytest.mark.parametrize("shape,g", [c for c in CASES if c[1] >= 2])
def test_advection_timestep(smi, shape, g, ncomp, real_t):
    """Conservative third-order upwind advection (kernel support 2) against the oracle's
    restatement; the reference body lives outside its tree, so this parity is unpinned."""
    rng = np.random.default_rng(7)
    ctx = make_ctx(smi, shape, g, real_t)
    f = rand_field(rng, (ncomp,) + ctx.padded_shape, real_t)
    vel = rand_field(rng, (3,) + ctx.padded_shape, real_t)
    df, dvel = ctx.from_numpy(f), ctx.from_numpy(vel)
    dflux = ctx.from_numpy(rand_field(rng, ctx.padded_shape, real_t))  # stale garbage must not survive
    c = real_t(0.05)
    _call(smi, "smi_advection_timestep_eno3", ctx.handle, df.ptr, dflux.ptr, dvel.ptr, float(c), ncomp)
    want, wflux = f.copy(), np.zeros(ctx.padded_shape, dtype=real_t)
    for k in range(ncomp):
        st.advection_timestep_eno3(want[k], wflux, vel, c, g)
    assert_close(df.get(), want, real_t, factor=1.0, what="advection timestep")
    assert_close(dflux.get(), wflux, real_t, scale=np.abs(want).max(), factor=1.0, what="advection flux scratch")


@pytest.mark.parametrize("shape,g", RANDOM_CASES)
def test_tiled_kernels_on_random_shapes(smi, shape, g):
    """Diffusion flux, curl and the one-pass rotational advection update on shapes chosen to hit
    partial tiles (f32 and f64 alternate)."""
    real_t = np.float32 if (shape[0] + shape[1]) % 2 == 0 else np.float64
    rng = np.random.default_rng(sum(shape) + g)
    ctx = make_ctx(smi, shape, g, real_t)
    assert ctx.padded_shape[2] % 4 == 0
    f = rand_field(rng, (3,) + ctx.padded_shape, real_t)
    c = real_t(0.211)
    df = ctx.from_numpy(f)
    dflux = ctx.from_numpy(rand_field(rng, (3,) + ctx.padded_shape, real_t))
    _call(smi, "smi_diffusion_flux", ctx.handle, dflux.ptr, df.ptr, float(c), 3)
    want = np.zeros_like(f)
    for k in range(3):
        st.diffusion_flux(want[k], f[k], c, g)
    assert_close(dflux.get(), want, real_t, what="diffusion flux")
    dc = ctx.from_numpy(rand_field(rng, (3,) + ctx.padded_shape, real_t))
    _call(smi, "smi_curl", ctx.handle, dc.ptr, df.ptr, float(c))
    wc = np.zeros_like(f)
    st.curl(wc, f, c, g)
    assert_close(dc.get(), wc, real_t, what="curl")
    # out = w + p curl(u x w): cross product, then the vorticity update, as the reference sequences them
    u = rand_field(rng, (3,) + ctx.padded_shape, real_t)
    du, dout = ctx.from_numpy(u), ctx.from_numpy(rand_field(rng, (3,) + ctx.padded_shape, real_t))
    _call(smi, "smi_rotational_advection_update", ctx.handle, dout.ptr, du.ptr, df.ptr, float(c))
    cross = np.zeros_like(f)
    st.elementwise_cross_product(cross, u, f)
    want = f.copy()
    st.update_vorticity_from_velocity_forcing(want, cross, c, g)
    assert_close(dout.get(), want, real_t, factor=1.0, what="rotational advection update")
    # ... followed by the diffusion timestep, as one operator (one pass in single precision)
    dscr = ctx.from_numpy(rand_field(rng, (3,) + ctx.padded_shape, real_t))
    dout2 = ctx.from_numpy(rand_field(rng, (3,) + ctx.padded_shape, real_t))
    c2 = real_t(0.093)
    _call(smi, "smi_advection_diffusion_update", ctx.handle, dout2.ptr, du.ptr, df.ptr, dscr.ptr, float(c), float(c2))
    st.vector_diffusion_timestep(want, np.zeros(ctx.padded_shape, dtype=real_t), c2, g)
    assert_close(dout2.get(), want, real_t, factor=2.0, what="advection + diffusion update")
    assert np.array_equal(du.get(), u) and np.array_equal(df.get(), f), "inputs were modified"


@pytest.mark.parametrize("real_t", [np.float32, np.float64])
@pytest.mark.parametrize("zchunk", [None, "8", "13"])
@pytest.mark.parametrize("shape,g", [((40, 70, 260), 2), ((24, 100, 60), 2), ((8, 8, 8), 2), ((33, 45, 124), 2),
                                     ((16, 16, 34), 3), ((50, 21, 512), 2)])
def test_fused_advection_diffusion_shapes(smi, shape, g, zchunk, real_t):
    """The one-pass advection + diffusion kernel (1024 lanes in single, 512 in double precision) on grids with one to several tiles per plane,
    partial tiles in both directions and, through the fused_zchunk option, one to several z chunks with partial last
    chunks: against the reference's operator sequence (cross product, vorticity update, diffusion timestep)
    on the whole array, ghost cells included, and against the two-pass path of the library bit for bit."""
    rng = np.random.default_rng(sum(shape))
    ctx = make_ctx(smi, shape, g, real_t)
    if zchunk is not None:
        _call(smi, "smi_ctx_set_option", ctx.handle, b"fused_zchunk", int(zchunk))
    assert ctx.padded_shape[2] % 4 == 0
    u = rand_field(rng, (3,) + ctx.padded_shape, real_t)
    w = rand_field(rng, (3,) + ctx.padded_shape, real_t)
    p, c = real_t(0.211), real_t(0.093)
    du, dw = ctx.from_numpy(u), ctx.from_numpy(w)
    dscr = ctx.from_numpy(rand_field(rng, (3,) + ctx.padded_shape, real_t))
    dout = ctx.from_numpy(rand_field(rng, (3,) + ctx.padded_shape, real_t))
    smi._lib.call("smi_prof_enable", ctx.handle, 1)
    smi._lib.call("smi_prof_reset", ctx.handle)
    _call(smi, "smi_advection_diffusion_update", ctx.handle, dout.ptr, du.ptr, dw.ptr, dscr.ptr, float(p), float(c))
    assert smi._lib.prof_collect(ctx).get("fused_advect_diffuse", {}).get("launches", 0) >= 1, "the one-pass kernel did not run"
    smi._lib.call("smi_prof_enable", ctx.handle, 0)
    cross = np.zeros_like(w)
    st.elementwise_cross_product(cross, u, w)
    want = w.copy()
    st.update_vorticity_from_velocity_forcing(want, cross, p, g)
    st.vector_diffusion_timestep(want, np.zeros(ctx.padded_shape, dtype=real_t), c, g)
    assert_close(dout.get(), want, real_t, factor=2.0, what="advection + diffusion, one pass")
    # the two-pass path of the library: same arithmetic in the same order
    _call(smi, "smi_ctx_set_option", ctx.handle, b"no_fused_advect_diffuse", 1)
    dout2 = ctx.from_numpy(rand_field(rng, (3,) + ctx.padded_shape, real_t))
    _call(smi, "smi_advection_diffusion_update", ctx.handle, dout2.ptr, du.ptr, dw.ptr, dscr.ptr, float(p), float(c))
    assert np.array_equal(dout.get(), dout2.get()), "one pass and two passes must agree bit for bit"
    assert np.array_equal(du.get(), u) and np.array_equal(dw.get(), w), "inputs were modified"
    # the launch schedule of the overlapped multi-rank path (interior chunks first, then the two outer chunks)
    _call(smi, "smi_ctx_set_option", ctx.handle, b"no_fused_advect_diffuse", 0)
    _call(smi, "smi_ctx_set_option", ctx.handle, b"fused_split_launch", 1)
    dout3 = ctx.from_numpy(rand_field(rng, (3,) + ctx.padded_shape, real_t))
    _call(smi, "smi_advection_diffusion_update", ctx.handle, dout3.ptr, du.ptr, dw.ptr, dscr.ptr, float(p), float(c))
    assert np.array_equal(dout.get(), dout3.get()), "split launches must cover the array like the single launch"
