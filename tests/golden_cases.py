"""Seeded inputs and oracle evaluations of the golden cases (shared by the generator script
in tests/golden/ and by the tests that compare against the stored vectors)."""
import numpy as np

from oracle import immersed_boundary as ib
from oracle import stencils_3d as st
from oracle.flow_3d import FlowSimulator3D
from oracle.poisson import UnboundedPoissonSolver3D

G = 2
SHAPE = (12, 16, 20)  # interior (z, y, x)
PSHAPE = tuple(n + 2 * G for n in SHAPE)


def stencil_inputs(real_t=np.float64):
    rng = np.random.default_rng(2024)
    f = rng.random((3,) + PSHAPE).astype(real_t) - real_t(0.5)
    h = rng.random((3,) + PSHAPE).astype(real_t) - real_t(0.5)
    return f, h


def stencils():
    f, h = stencil_inputs()
    out = {}
    flux = np.zeros_like(f)
    for k in range(3):
        st.diffusion_flux(flux[k], f[k], 0.137, G)
    out["diffusion_flux"] = flux
    c = np.zeros_like(f)
    st.curl(c, f, 0.5, G)
    out["curl"] = c
    w = h.copy()
    st.update_vorticity_from_velocity_forcing(w, f, 0.25, G)
    out["update_vorticity"] = w
    x = np.zeros_like(f)
    st.elementwise_cross_product(x, f, h)
    out["cross"] = x
    a = f.copy()
    scratch = np.zeros(PSHAPE)
    for k in range(3):
        st.advection_timestep_eno3(a[k], scratch, h, 0.05, G)
    out["advection_timestep"] = a
    fl = f.copy()
    st.vector_laplacian_filter(fl, np.zeros(PSHAPE), np.zeros(PSHAPE), 2, G, filter_type="multiplicative")
    out["filter_multiplicative_2"] = fl
    return out


def poisson_inputs(real_t=np.float64):
    rng = np.random.default_rng(77)
    n = (16, 16, 32)
    rhs = np.zeros((3,) + tuple(m + 2 * G for m in n), dtype=real_t)
    rhs[:, G:-G, G:-G, G:-G] = rng.random((3,) + n).astype(real_t) - real_t(0.5)
    return n, rhs


def poisson():
    n, rhs = poisson_inputs()
    solver = UnboundedPoissonSolver3D(*n, x_range=1.0, real_t=np.float64)
    sol = np.zeros_like(rhs)
    solver.vector_field_solve(sol, rhs, G)
    return {"solution": sol}


def flow_inputs(sim, real_t=np.float64):
    rng = np.random.default_rng(11)
    x, y, z = sim.position_field
    w = np.zeros(sim.vorticity_field.shape)
    for _ in range(3):
        c = rng.uniform(0.35, 0.65, 3)
        amp = rng.uniform(-1, 1, 3)
        r2 = (x - c[0]) ** 2 + (y - c[1] * y.max()) ** 2 + (z - c[2] * z.max()) ** 2
        for k in range(3):
            w[k] += amp[k] * np.exp(-r2 / (2 * 0.07**2))
    return w.astype(real_t)


FLOW_KW = dict(grid_size=(16, 16, 32), x_range=1.0, kinematic_viscosity=1e-2, flow_type="navier_stokes",
               ghost_size=G, with_free_stream_flow=True)
FLOW_FREE_STREAM = np.array([1.0, 0.5, -0.25])
FLOW_DT = 2.5e-3


def flow_steps():
    sim = FlowSimulator3D(real_t=np.float64, **FLOW_KW)
    sim.vorticity_field[...] = flow_inputs(sim)
    sim.compute_flow_velocity(free_stream_velocity=FLOW_FREE_STREAM)
    for _ in range(2):
        sim.time_step(FLOW_DT, free_stream_velocity=FLOW_FREE_STREAM)
    return {"vorticity": sim.vorticity_field.copy(), "velocity": sim.velocity_field.copy(),
            "stream_func": sim.stream_func_field.copy()}


def ib_inputs():
    rng = np.random.default_rng(5)
    n = (16, 16, 16)
    dx = 1.0 / 16
    m = 37
    pos = rng.uniform(0.25, 0.75, (3, m))
    eul = rng.random((3,) + tuple(k + 2 * G for k in n)) - 0.5
    lag = rng.random((3, m)) - 0.5
    return n, dx, pos, eul, lag


def immersed_boundary():
    n, dx, pos, eul, lag = ib_inputs()
    m = pos.shape[1]
    support = np.zeros((3, 4, 4, 4, m))
    nearest = np.zeros((3, m), dtype=np.int64)
    ib.local_eulerian_grid_support_of_lagrangian_grid(support, nearest, pos, dx, dx / 2, 2, np.array([-G, -G, -G]))
    weights = np.zeros((4, 4, 4, m))
    ib.cosine_interpolation_weights(weights, support, dx, np.float64)
    out_lag = np.zeros((3, m))
    ib.eulerian_to_lagrangian_grid_interpolation(out_lag, eul, weights, nearest, dx)
    out_eul = np.zeros_like(eul)
    ib.lagrangian_to_eulerian_grid_interpolation(out_eul, lag, weights, nearest)
    return {"nearest_index": nearest, "weights": weights, "interpolated": out_lag, "spread": out_eul}


CASES = {"stencils_f64": stencils, "poisson_f64": poisson, "flow_steps_f64": flow_steps,
         "immersed_boundary_f64": immersed_boundary}
