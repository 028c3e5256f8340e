"""Oracle (test infrastructure): single-rank restatement of UnboundedFlowSimulator2D
(sopht_mpi/simulator/flow/flow_simulators_mpi_2d.py: domain :110-145, fields :147-164, steps
:260-298, time-step rule :300-321).  This is the CPU path the reference's own single-rank case
(2D 256^2 vorticity step, the Lamb-Oseen vortex) runs on.
"""
import numpy as np

from . import stencils_2d as st
from .flow_3d import get_test_tol, local_coordinates
from .poisson import UnboundedPoissonSolver2D


class FlowSimulator2D:
    def __init__(self, grid_size, x_range, kinematic_viscosity, time=0.0, CFL=0.1, flow_type="passive_scalar",
                 with_free_stream_flow=False, real_t=np.float32, ghost_size=2, **kwargs):
        if flow_type not in ("passive_scalar", "navier_stokes", "navier_stokes_with_forcing"):
            raise ValueError("Invalid flow type given")
        if flow_type == "passive_scalar" and with_free_stream_flow:
            raise ValueError("Free stream flow not defined for passive advection diffusion!")
        self.grid_dim = 2
        self.grid_size = grid_size
        self.grid_size_y, self.grid_size_x = grid_size
        self.x_range, self.real_t, self.flow_type = x_range, real_t, flow_type
        self.with_free_stream_flow = with_free_stream_flow
        self.kinematic_viscosity, self.CFL, self.time = kinematic_viscosity, CFL, time
        self.ghost_size = g = ghost_size
        self.y_range = x_range * self.grid_size_y / self.grid_size_x
        self.dx = real_t(x_range / self.grid_size_x)
        self.x_coords = local_coordinates(self.grid_size_x, 0, self.dx, g, real_t)
        self.y_coords = local_coordinates(self.grid_size_y, 0, self.dx, g, real_t)
        self.position_field = np.flipud(np.array(np.meshgrid(self.y_coords, self.x_coords, indexing="ij")))
        shape = tuple(n + 2 * g for n in grid_size)
        self.primary_scalar_field = np.zeros(shape, dtype=real_t)
        self.velocity_field = np.zeros((2,) + shape, dtype=real_t)
        self.buffer_scalar_field = np.zeros(shape, dtype=real_t)
        if flow_type in ("navier_stokes", "navier_stokes_with_forcing"):
            self.vorticity_field = self.primary_scalar_field.view()
            self.stream_func_field = np.zeros_like(self.vorticity_field)
            self.penalty_zone_width = kwargs.get("penalty_zone_width", 2)
            self.unbounded_poisson_solver = UnboundedPoissonSolver2D(self.grid_size_y, self.grid_size_x,
                                                                     x_range=x_range, real_t=real_t)
        if flow_type == "navier_stokes_with_forcing":
            self.eul_grid_forcing_field = np.zeros_like(self.velocity_field)

    def time_step(self, dt, free_stream_velocity=(0.0, 0.0)):
        if self.flow_type == "passive_scalar":
            self.advection_and_diffusion_timestep(dt)
        elif self.flow_type == "navier_stokes":
            self.navier_stokes_timestep(dt, free_stream_velocity)
        else:
            self.navier_stokes_with_forcing_timestep(dt, free_stream_velocity)
        self.time += dt

    def advection_and_diffusion_timestep(self, dt):
        g, t = self.ghost_size, self.real_t
        st.advection_timestep_eno3(self.primary_scalar_field, self.buffer_scalar_field, self.velocity_field,
                                   t(dt / self.dx), g)
        st.diffusion_timestep(self.primary_scalar_field, self.buffer_scalar_field,
                              t(self.kinematic_viscosity * dt / self.dx / self.dx), g)

    def compute_velocity_from_vorticity(self):
        g = self.ghost_size
        st.penalise_field_boundary(self.vorticity_field, self.penalty_zone_width, self.dx, self.x_coords,
                                   self.y_coords, g)
        self.unbounded_poisson_solver.solve(self.stream_func_field, self.vorticity_field, g)
        st.outplane_field_curl(self.velocity_field, self.stream_func_field, self.real_t(0.5 / self.dx), g)

    def navier_stokes_timestep(self, dt, free_stream_velocity=(0.0, 0.0)):
        self.advection_and_diffusion_timestep(dt)
        self.compute_velocity_from_vorticity()
        if self.with_free_stream_flow:
            fs = np.asarray(free_stream_velocity, dtype=self.real_t).reshape(2, 1, 1)
            self.velocity_field[...] = self.velocity_field + fs

    def navier_stokes_with_forcing_timestep(self, dt, free_stream_velocity=(0.0, 0.0)):
        st.update_vorticity_from_velocity_forcing(self.vorticity_field, self.eul_grid_forcing_field,
                                                  self.real_t(dt / (2 * self.dx)), self.ghost_size)
        self.navier_stokes_timestep(dt, free_stream_velocity)
        self.eul_grid_forcing_field[...] = 0

    def compute_stable_timestep(self, dt_prefac=1, precision="single"):
        g = self.ghost_size
        mag = self.buffer_scalar_field
        mag[...] = np.sum(np.fabs(self.velocity_field), axis=0)
        dt = min(self.CFL * self.dx / (np.amax(mag[g:-g, g:-g]) + get_test_tol(precision)),
                 0.9 * self.dx**2 / (2 * self.grid_dim) / self.kinematic_viscosity)
        return dt * dt_prefac

    def get_max_vorticity(self):
        g = self.ghost_size
        return np.amax(self.vorticity_field[g:-g, g:-g])
