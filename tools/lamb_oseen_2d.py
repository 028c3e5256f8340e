"""The reference's 2D 256^2 case (Lamb-Oseen vortex advected by
a free stream while it diffuses, examples/2d_examples/LambOseenVortexCase/lamb_oseen_vortex.py
:21-66, :122-180) on the CPU restatement, one process.  Prints the errors the reference prints.

  python tools/lamb_oseen_2d.py [grid] [precision] [cpu|gpu]

"gpu" runs the same case through UnboundedFlowSimulator2D of the HIP library instead.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.flow_2d import FlowSimulator2D  # noqa: E402


def lamb_oseen_vorticity(x, y, x_cm, y_cm, nu, gamma, t, real_t):
    return (gamma / (4 * real_t(np.pi) * nu * t) * np.exp(-((x - x_cm) ** 2 + (y - y_cm) ** 2) / (4 * nu * t))
            ).astype(real_t)


def lamb_oseen_velocity(x, y, x_cm, y_cm, nu, gamma, t, real_t):
    r = np.sqrt((x - x_cm) ** 2 + (y - y_cm) ** 2).astype(real_t)
    v_theta = (gamma / real_t(2 * np.pi * r) * (1 - np.exp(-(r**2) / (4 * nu * t)))).astype(real_t)
    vel = np.zeros((2,) + x.shape, dtype=real_t)
    vel[0] = v_theta * (-(y - y_cm) / r)
    vel[1] = v_theta * ((x - x_cm) / r)
    return vel


def run(grid=256, precision="double", t_end=1.4, device="cpu"):
    real_t = np.float64 if precision == "double" else np.float32
    nu, x_cm, y_cm, t_start = 1e-3, 0.3, 0.3, 1.0
    gamma = 4 * np.pi * nu * t_start
    if device == "gpu":
        sys.path.insert(0, os.path.join(ROOT, "sopht-mpi_amd"))
        from sopht_mpi_amd.simulator import UnboundedFlowSimulator2D as Sim
    else:
        Sim = FlowSimulator2D
    sim = Sim((grid, grid), 1.0, nu, flow_type="navier_stokes", with_free_stream_flow=True,
              real_t=real_t, time=t_start)
    x, y = sim.position_field
    sim.vorticity_field[...] = lamb_oseen_vorticity(x, y, x_cm, y_cm, nu, gamma, t_start, real_t)
    free_stream = np.ones(2, dtype=real_t)
    # the example initialises the velocity with the analytic field (of t_end) plus nothing else
    sim.velocity_field[...] = lamb_oseen_velocity(x, y, x_cm, y_cm, nu, gamma, t_end, real_t)
    steps, t0 = 0, time.perf_counter()
    while sim.time < t_end:
        dt = sim.compute_stable_timestep(precision=precision)
        sim.time_step(dt, free_stream_velocity=free_stream)
        steps += 1
    elapsed = time.perf_counter() - t0
    g = sim.ghost_size
    inner = (slice(g, -g), slice(g, -g))
    xf, yf = x_cm + free_stream[0] * (sim.time - t_start), y_cm + free_stream[1] * (sim.time - t_start)
    err_w = np.fabs(np.asarray(sim.vorticity_field) - lamb_oseen_vorticity(x, y, xf, yf, nu, gamma, sim.time, real_t))[inner]
    sim.compute_velocity_from_vorticity()
    err_u = np.fabs(np.asarray(sim.velocity_field) - lamb_oseen_velocity(x, y, xf, yf, nu, gamma, sim.time, real_t))[
        (slice(None),) + inner]
    return {"grid": grid, "device": device, "steps": steps, "seconds": elapsed,
            "grid_point_updates_per_s": grid * grid * steps / elapsed,
            "vorticity_l2": float(np.linalg.norm(err_w) * sim.dx), "vorticity_linf": float(err_w.max()),
            "velocity_l2": float(np.linalg.norm(err_u) * sim.dx), "velocity_linf": float(err_u.max()),
            "max_vorticity": float(sim.get_max_vorticity())}


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    prec = sys.argv[2] if len(sys.argv) > 2 else "double"
    dev = sys.argv[3] if len(sys.argv) > 3 else "cpu"
    print(run(n, prec, device=dev))
