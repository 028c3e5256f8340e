"""Shared helpers of the GPU parity tests: tolerances and small factories."""
import json
import os

import numpy as np

# Floating-point parity bar.  north_star: "matching reference vorticity to rtol 1e-5" (f32).  A pointwise rtol is
# meaningless where a field crosses zero, so the bound is relative to the largest magnitude of the expected array:
# |got - want| <= factor * TOL * max|want|.  TOL is about 17 units of round-off (eps = 1.2e-7 / 2.2e-16), five times
# below the north-star bar: single operators and whole flow steps measure 1 to 3 eps against the oracle on the GPU
# (run the suite with SMI_TEST_ERRLOG=<file> to log every comparison), so a factor of 1 leaves a margin of about
# ten and a regression of a few eps per operator fails.  Comparisons that legitimately carry more round-off pass
# a `factor` and say why.
TOL = {np.dtype(np.float32): 2e-6, np.dtype(np.float64): 4e-15}
NORTH_STAR_RTOL = 1e-5


_ERRLOG = os.environ.get("SMI_TEST_ERRLOG")


def _record(what, real_t, rel_err, factor):
    """With SMI_TEST_ERRLOG=<file>: append the measured relative error of every comparison (used to set the bounds)."""
    if _ERRLOG:
        with open(_ERRLOG, "a") as fh:
            fh.write(json.dumps({"test": os.environ.get("PYTEST_CURRENT_TEST", ""), "what": what,
                                 "dtype": np.dtype(real_t).name, "rel_err": rel_err, "factor": factor}) + "\n")


def assert_close(got, want, real_t, scale=None, factor=1.0, what=""):
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    ref = float(np.max(np.abs(want))) if scale is None else float(scale)
    tol = factor * TOL[np.dtype(real_t)] * max(ref, np.finfo(np.float64).tiny)
    err = float(np.max(np.abs(got.astype(np.float64) - want.astype(np.float64)))) if got.size else 0.0
    _record(what, real_t, err / max(ref, np.finfo(np.float64).tiny), factor)
    assert err <= tol, f"{what}: max abs err {err:.3e} > {tol:.3e} (scale {ref:.3e})"


def make_ctx(smi, shape, ghost, real_t, rank=0, nranks=1):
    from sopht_mpi_amd.device import DeviceContext

    return DeviceContext(shape, ghost, real_t, rank=rank, nranks=nranks)


def rand_field(rng, shape, real_t):
    return rng.random(shape).astype(real_t) - real_t(0.5)


def assert_pointwise(got, want, real_t, what=""):
    """Pointwise form of the north-star bar for a field: |got - want| <= 1e-5 |want| + 1e-6 max|want| in single
    precision (1e-12 / 1e-13 in double)."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    rtol, arel = (NORTH_STAR_RTOL, 1e-6) if np.dtype(real_t) == np.float32 else (1e-12, 1e-13)
    bound = rtol * np.abs(want) + arel * float(np.max(np.abs(want)))
    worst = float(np.max(np.abs(got - want) - bound)) if got.size else 0.0
    assert worst <= 0.0, f"{what}: pointwise bound exceeded by {worst:.3e}"
