"""GPU test of smi_poisson_solve_noalloc: the scalar unbounded Poisson solve on the library's fixed per-GPU
scratch array (for callers that cannot allocate device memory) against smi_poisson_solve, through the C ABI.

Both entry points launch the same kernels with the same plans, so the results must be bit-identical.  The scratch
array keeps the contents of the previous call: the calls below follow each other on one GPU with shrinking and
growing grids and both precisions, so stale contents of a larger solve lie under every smaller one."""
import ctypes as C

import numpy as np
import pytest

from helpers import make_ctx, rand_field

pytestmark = pytest.mark.gpu


def _solver(ctx, shape, x_range, real_t, backend=1):
    from sopht_mpi_amd._lib import call

    h = C.c_void_p()
    dx = real_t(x_range / shape[2])
    call("smi_poisson_create", C.byref(h), ctx.handle, shape[0], shape[1], shape[2], float(dx), float(x_range), backend)
    return h


def _both(smi, shape, real_t, seed, g=2):
    """(solution of smi_poisson_solve, solution of smi_poisson_solve_noalloc, untouched start values)"""
    call = smi._lib.call
    rng = np.random.default_rng(seed)
    ctx = make_ctx(smi, shape, g, real_t)
    h = _solver(ctx, shape, 1.3, real_t)
    rhs = rand_field(rng, ctx.padded_shape, real_t)
    start = rand_field(rng, ctx.padded_shape, real_t)
    drhs = ctx.from_numpy(rhs)
    want, got = ctx.from_numpy(start), ctx.from_numpy(start)
    call("smi_poisson_solve", h, want.ptr, drhs.ptr)
    call("smi_poisson_solve_noalloc", h, got.ptr, drhs.ptr)
    out = want.get(), got.get(), start
    assert np.array_equal(drhs.get(), rhs), "rhs was modified"
    call("smi_poisson_destroy", h)
    return out


def test_noalloc_solve_is_bit_identical_on_consecutive_calls(smi):
    # larger first, then smaller (stale contents of the first call lie beyond and under the second), then other
    # aspect ratios, the other precision, and the largest grids that fit
    cases = [((64, 64, 64), np.float32), ((32, 32, 32), np.float32), ((16, 64, 32), np.float64),
             ((8, 8, 16), np.float32), ((64, 64, 64), np.float64), ((128, 128, 128), np.float32),
             ((8, 192, 16), np.float32), ((32, 32, 32), np.float64)]
    for n, (shape, real_t) in enumerate(cases):
        print("case", shape, np.dtype(real_t).name, flush=True)
        want, got, start = _both(smi, shape, real_t, seed=100 + n)
        g = 2
        inner = (slice(g, -g),) * 3
        assert np.all(np.isfinite(got))
        assert np.array_equal(got, want), f"{shape} {np.dtype(real_t).name}: differs from smi_poisson_solve"
        mask = np.ones(got.shape, dtype=bool)
        mask[inner] = False
        assert np.array_equal(got[mask], start[mask]), "ghost cells of the solution were modified"
        assert np.max(np.abs(got[inner])) > 0


def test_noalloc_solve_repeats_itself(smi):
    """Same input twice in a row with a smaller solve of the other precision in between: the second result equals the first."""
    call = smi._lib.call
    shape, real_t, g = (32, 64, 32), np.float32, 2
    ctx = make_ctx(smi, shape, g, real_t)
    h = _solver(ctx, shape, 1.0, real_t)
    rng = np.random.default_rng(5)
    drhs = ctx.from_numpy(rand_field(rng, ctx.padded_shape, real_t))
    a, b = ctx.field(), ctx.field()
    call("smi_poisson_solve_noalloc", h, a.ptr, drhs.ptr)
    _both(smi, (16, 16, 16), np.float64, seed=6)
    call("smi_poisson_solve_noalloc", h, b.ptr, drhs.ptr)
    assert np.array_equal(a.get(), b.get())
    call("smi_poisson_destroy", h)


def test_noalloc_solve_rejects_what_does_not_fit(smi):
    from sopht_mpi_amd._lib import SophtMIError, call, lib

    assert lib.smi_poisson_noalloc_capacity() == 64 << 20
    # 128^3 in double precision needs 113 MB of work space
    shape, real_t = (128, 128, 128), np.float64
    ctx = make_ctx(smi, shape, 2, real_t)
    h = _solver(ctx, shape, 1.0, real_t)
    f, r = ctx.field(), ctx.field()
    with pytest.raises(SophtMIError, match="fixed scratch"):
        call("smi_poisson_solve_noalloc", h, f.ptr, r.ptr)
    call("smi_poisson_destroy", h)
    # the rocFFT back end may allocate inside its transforms
    shape, real_t = (16, 16, 16), np.float32
    ctx = make_ctx(smi, shape, 2, real_t)
    h = _solver(ctx, shape, 1.0, real_t, backend=0)
    f, r = ctx.field(), ctx.field()
    with pytest.raises(SophtMIError, match="SMI_POISSON_PRUNED"):
        call("smi_poisson_solve_noalloc", h, f.ptr, r.ptr)
    call("smi_poisson_destroy", h)
