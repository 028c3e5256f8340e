"""Oracle (test infrastructure): single-rank restatement of UnboundedFlowSimulator3D.

Follows sopht_mpi/simulator/flow/flow_simulators_mpi_3d.py: domain :124-158, fields
:170-196, step sequences :356-424, diagnostics :426-476; the sequence is cross-checked
against the reference's own hand re-sequencing in
tests/test_simulator/test_flow/test_flow_simulators_3d.py:266-318.
"""
import numpy as np

from . import stencils_3d as st
from .poisson import UnboundedPoissonSolver3D


def get_test_tol(precision="single"):
    """``sopht.utils.precision.get_test_tol``: constants are not visible in the reference
    tree (parity unpinned); it only enters compute_stable_timestep through ``+ tol``."""
    return 1e-4 if precision == "single" else 1e-11


def local_coordinates(n_local, substart, dx, g, real_t):
    """flow_simulators_mpi_3d.py:129-153: cell centres (i + 1/2) dx continued into ghosts."""
    shift = dx / 2.0
    ghost_shift = g * dx
    start = substart * dx
    end = (substart + n_local) * dx
    return np.linspace(shift + start - ghost_shift, end - shift + ghost_shift, n_local + 2 * g).astype(real_t)


class FlowSimulator3D:
    def __init__(self, grid_size, x_range, kinematic_viscosity, time=0.0, CFL=0.1,
                 flow_type="passive_scalar", filter_vorticity=False, real_t=np.float32,
                 ghost_size=2, **kwargs):
        if flow_type not in ("passive_scalar", "passive_vector", "navier_stokes", "navier_stokes_with_forcing"):
            raise ValueError("Invalid flow type given")
        self.grid_dim = 3
        self.grid_size = grid_size
        self.grid_size_z, self.grid_size_y, self.grid_size_x = grid_size
        self.x_range = x_range
        self.real_t = real_t
        self.flow_type = flow_type
        self.kinematic_viscosity = kinematic_viscosity
        self.CFL = CFL
        self.time = time
        self.filter_vorticity = filter_vorticity
        self.ghost_size = g = ghost_size
        self.y_range = x_range * self.grid_size_y / self.grid_size_x
        self.z_range = x_range * self.grid_size_z / self.grid_size_x
        self.dx = real_t(x_range / self.grid_size_x)
        self.x_coords = local_coordinates(self.grid_size_x, 0, self.dx, g, real_t)
        self.y_coords = local_coordinates(self.grid_size_y, 0, self.dx, g, real_t)
        self.z_coords = local_coordinates(self.grid_size_z, 0, self.dx, g, real_t)
        self.position_field = np.flipud(
            np.array(np.meshgrid(self.z_coords, self.y_coords, self.x_coords, indexing="ij"))
        )
        shape = tuple(n + 2 * g for n in grid_size)
        self.velocity_field = np.zeros((3,) + shape, dtype=real_t)
        self.buffer_scalar_field = np.zeros(shape, dtype=real_t)
        if flow_type == "passive_scalar":
            self.primary_scalar_field = np.zeros(shape, dtype=real_t)
        else:
            self.primary_vector_field = np.zeros((3,) + shape, dtype=real_t)
        if flow_type in ("navier_stokes", "navier_stokes_with_forcing"):
            self.vorticity_field = self.primary_vector_field.view()
            self.stream_func_field = np.zeros_like(self.vorticity_field)
            self.buffer_vector_field = np.zeros_like(self.vorticity_field)
            self.penalty_zone_width = kwargs.get("penalty_zone_width", 2)
            self.with_free_stream_flow = kwargs.get("with_free_stream_flow", False)
            self.filter_setting_dict = kwargs.get("filter_setting_dict") or {"order": 2, "type": "multiplicative"}
            self.unbounded_poisson_solver = UnboundedPoissonSolver3D(
                self.grid_size_z, self.grid_size_y, self.grid_size_x, x_range=x_range, real_t=real_t
            )
        if flow_type == "navier_stokes_with_forcing":
            self.eul_grid_forcing_field = np.zeros_like(self.velocity_field)

    # ---------------------------------------------------------------- steps
    def time_step(self, dt, free_stream_velocity=None):
        {
            "passive_scalar": self.scalar_advection_and_diffusion_timestep,
            "passive_vector": self.vector_advection_and_diffusion_timestep,
            "navier_stokes": self.rotational_form_navier_stokes_timestep,
            "navier_stokes_with_forcing": self.navier_stokes_with_forcing_timestep,
        }[self.flow_type](dt, free_stream_velocity)
        self.time += dt

    def scalar_advection_and_diffusion_timestep(self, dt, _=None):
        t, g = self.real_t, self.ghost_size
        st.advection_timestep_eno3(self.primary_scalar_field, self.buffer_scalar_field,
                                   self.velocity_field, t(dt / self.dx), g)
        st.diffusion_timestep(self.primary_scalar_field, self.buffer_scalar_field,
                              t(self.kinematic_viscosity * dt / self.dx / self.dx), g)

    def vector_advection_and_diffusion_timestep(self, dt, _=None):
        t, g = self.real_t, self.ghost_size
        for c in range(3):
            st.advection_timestep_eno3(self.primary_vector_field[c], self.buffer_scalar_field,
                                       self.velocity_field, t(dt / self.dx), g)
        st.vector_diffusion_timestep(self.primary_vector_field, self.buffer_scalar_field,
                                     t(self.kinematic_viscosity * dt / self.dx / self.dx), g)

    def compute_flow_velocity(self, free_stream_velocity=None):
        """:382-393."""
        t, g = self.real_t, self.ghost_size
        st.penalise_vector_field_boundary(self.vorticity_field, self.penalty_zone_width, self.dx,
                                          self.x_coords, self.y_coords, self.z_coords, g)
        self.unbounded_poisson_solver.vector_field_solve(self.stream_func_field, self.vorticity_field, g)
        st.curl(self.velocity_field, self.stream_func_field, t(0.5 / self.dx), g)
        if self.with_free_stream_flow:
            fs = np.asarray(free_stream_velocity, dtype=t).reshape(3, 1, 1, 1)
            self.velocity_field[...] = self.velocity_field + fs

    def rotational_form_navier_stokes_timestep(self, dt, free_stream_velocity=None):
        """:395-413."""
        t, g = self.real_t, self.ghost_size
        st.elementwise_cross_product(self.buffer_vector_field, self.velocity_field, self.vorticity_field)
        st.update_vorticity_from_velocity_forcing(self.vorticity_field, self.buffer_vector_field,
                                                  t(dt / (2 * self.dx)), g)
        st.vector_diffusion_timestep(self.vorticity_field, self.buffer_scalar_field,
                                     t(self.kinematic_viscosity * dt / self.dx / self.dx), g)
        if self.filter_vorticity:
            st.vector_laplacian_filter(self.vorticity_field, self.buffer_vector_field[0],
                                       self.buffer_vector_field[1], self.filter_setting_dict["order"], g,
                                       self.filter_setting_dict["type"])
        self.compute_flow_velocity(free_stream_velocity)

    def navier_stokes_with_forcing_timestep(self, dt, free_stream_velocity=None):
        """:415-424."""
        t, g = self.real_t, self.ghost_size
        st.update_vorticity_from_velocity_forcing(self.vorticity_field, self.eul_grid_forcing_field,
                                                  t(dt / (2 * self.dx)), g)
        self.rotational_form_navier_stokes_timestep(dt, free_stream_velocity)
        self.eul_grid_forcing_field[...] = 0

    # ---------------------------------------------------------------- diagnostics
    def compute_stable_timestep(self, dt_prefac=1, precision="single"):
        """:426-449."""
        g = self.ghost_size
        tol = get_test_tol(precision)
        mag = self.buffer_scalar_field
        mag[...] = np.sum(np.fabs(self.velocity_field), axis=0)
        dt = min(
            self.CFL * self.dx / (np.amax(mag[g:-g, g:-g, g:-g]) + tol),
            0.9 * self.dx**2 / (2 * self.grid_dim) / (self.kinematic_viscosity + tol),
        )
        return dt * dt_prefac

    def get_vorticity_divergence_l2_norm(self):
        """:451-469."""
        g = self.ghost_size
        div = self.buffer_scalar_field
        st.divergence(div, self.vorticity_field, 1.0 / self.dx, g)
        return np.sqrt(np.linalg.norm(div[g:-g, g:-g, g:-g]) ** 2) * self.dx**1.5

    def get_max_vorticity(self):
        """:471-476."""
        g = self.ghost_size
        return np.amax(self.vorticity_field[:, g:-g, g:-g, g:-g])
