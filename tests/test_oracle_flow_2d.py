"""CPU: the 2D flow path of the oracle (the reference's own CPU-runnable case, the 256^2 Lamb-Oseen vortex, runs on it).

The reference's 2D case is the Lamb-Oseen vortex carried by a free stream
(examples/2d_examples/LambOseenVortexCase/lamb_oseen_vortex.py); it prints its errors
against the analytic solution without a threshold, so the checks here are: the errors are
small, they shrink with resolution (the scheme is first order in time), circulation is
conserved and the vortex arrives where the free stream takes it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from oracle import stencils_2d as st2  # noqa: E402
from oracle.flow_2d import FlowSimulator2D  # noqa: E402


def test_lamb_oseen_errors_shrink_with_resolution():
    import lamb_oseen_2d as lo

    coarse = lo.run(48, "double", t_end=1.2)
    fine = lo.run(96, "double", t_end=1.2)
    assert fine["vorticity_l2"] < 0.6 * coarse["vorticity_l2"]
    assert fine["velocity_l2"] < 0.6 * coarse["velocity_l2"]
    assert fine["velocity_linf"] < 2e-2
    # analytic peak at t = 1.2 is gamma / (4 pi nu t) = 1 / 1.2
    assert abs(fine["max_vorticity"] - 1 / 1.2) < 0.05


def test_step_conserves_circulation_and_moves_with_free_stream():
    n, nu = 64, 1e-3
    sim = FlowSimulator2D((n, n), 1.0, nu, flow_type="navier_stokes", with_free_stream_flow=True,
                          real_t=np.float64)
    x, y = sim.position_field
    sim.vorticity_field[...] = np.exp(-((x - 0.4) ** 2 + (y - 0.5) ** 2) / (2 * 0.05**2))
    g = sim.ghost_size
    inner = (slice(g, -g), slice(g, -g))
    circ0 = sim.vorticity_field[inner].sum() * sim.dx**2
    sim.compute_velocity_from_vorticity()
    fs = np.array([1.0, 0.0])
    sim.velocity_field[...] += fs.reshape(2, 1, 1)
    t_end = 0.1
    while sim.time < t_end:
        sim.time_step(sim.compute_stable_timestep(precision="double"), free_stream_velocity=fs)
    w = sim.vorticity_field[inner]
    assert abs(w.sum() * sim.dx**2 - circ0) < 1e-6 * abs(circ0)
    xc = (w * x[inner]).sum() / w.sum()
    yc = (w * y[inner]).sum() / w.sum()
    assert abs(xc - (0.4 + sim.time)) < 2e-3 and abs(yc - 0.5) < 2e-3


def test_2d_operators_single_cells():
    """Hand-checkable values: a unit spike through the 5-point flux, the psi -> velocity curl
    and the forcing update; band zeroing next to the physical boundaries."""
    g, n = 2, 8
    p = n + 2 * g
    f = np.zeros((p, p))
    f[6, 5] = 1.0
    flux = np.full((p, p), 7.0)
    st2.diffusion_flux(flux, f, 0.5, g)
    assert flux[6, 5] == -2.0 and flux[6, 4] == flux[6, 6] == flux[5, 5] == flux[7, 5] == 0.5
    assert np.count_nonzero(flux) == 5
    assert not flux[: g + 1].any() and not flux[:, -g - 1 :].any()
    vel = np.full((2, p, p), 7.0)
    st2.outplane_field_curl(vel, f, 0.5, g)
    assert vel[0, 5, 5] == 0.5 and vel[0, 7, 5] == -0.5      # u = p d(psi)/dy
    assert vel[1, 6, 4] == -0.5 and vel[1, 6, 6] == 0.5      # v = -p d(psi)/dx
    w = np.zeros((p, p))
    forcing = np.zeros((2, p, p))
    forcing[1, 6, 5] = 1.0                                   # f_y spike: w += p d(f_y)/dx
    st2.update_vorticity_from_velocity_forcing(w, forcing, 0.25, g)
    assert w[6, 4] == 0.25 and w[6, 6] == -0.25 and np.count_nonzero(w) == 2
