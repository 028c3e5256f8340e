"""UnboundedFlowSimulator2D with the reference's surface
(sopht_mpi/simulator/flow/flow_simulators_mpi_2d.py:20-328) on one MI355X."""
import ctypes as C
import logging

import numpy as np

from ..._lib import call, darray
from ...device import as_ptr
from ...numeric.eulerian_grid_ops import (
    UnboundedPoissonSolverMPI2D,
    gen_add_fixed_val_pyst_kernel_2d,
    gen_advection_timestep_euler_forward_conservative_eno3_pyst_mpi_kernel_2d,
    gen_diffusion_timestep_euler_forward_pyst_mpi_kernel_2d,
    gen_outplane_field_curl_pyst_mpi_kernel_2d,
    gen_penalise_field_boundary_pyst_mpi_kernel_2d,
    gen_set_fixed_val_pyst_kernel_2d,
    gen_update_vorticity_from_velocity_forcing_pyst_mpi_kernel_2d,
)
from ...sopht_compat import get_test_tol
from ...utils import MPIConstruct2D, MPIGhostCommunicator2D

logger = logging.getLogger("sopht_mpi_amd")


class UnboundedFlowSimulator2D:
    """Class for 2D unbounded flow simulator (single GPU)."""

    def __init__(self, grid_size, x_range, kinematic_viscosity, time=0.0, CFL=0.1, flow_type="passive_scalar",
                 with_free_stream_flow=False, real_t=np.float32, rank_distribution=None, ghost_size=2, **kwargs):
        self.grid_dim = 2
        self.grid_size = grid_size
        self.grid_size_y, self.grid_size_x = self.grid_size
        self.x_range = x_range
        self.real_t = real_t
        self.flow_type = flow_type
        self.with_free_stream_flow = with_free_stream_flow
        self.kinematic_viscosity = kinematic_viscosity
        self.CFL = CFL
        self.time = time
        if self.flow_type not in ["passive_scalar", "navier_stokes", "navier_stokes_with_forcing"]:
            raise ValueError("Invalid flow type given")
        if self.flow_type == "passive_scalar" and self.with_free_stream_flow:
            raise ValueError("Free stream flow not defined for passive advection diffusion!")
        self.rank_distribution = rank_distribution
        self.ghost_size = ghost_size
        self._world = kwargs.get("world", None)
        # True: one library call per Navier-Stokes step (smi2d_flow_step_ns); False: the reference's
        # operator-by-operator sequence
        self.fused_step = kwargs.get("fused_step", True)
        self.init_mpi()
        self.init_domain()
        self.init_fields()
        if self.flow_type in ["navier_stokes", "navier_stokes_with_forcing"]:
            self.penalty_zone_width = kwargs.get("penalty_zone_width", 2)
        self.compile_kernels()
        self.finalise_flow_timestep()

    def init_mpi(self):
        """:95-108."""
        self.mpi_construct = MPIConstruct2D(grid_size_y=self.grid_size_y, grid_size_x=self.grid_size_x,
                                            real_t=self.real_t, rank_distribution=self.rank_distribution,
                                            world=self._world)
        self.mpi_ghost_exchange_communicator = MPIGhostCommunicator2D(
            ghost_size=self.ghost_size, mpi_construct=self.mpi_construct, full_exchange=False)
        self.ctx = self.mpi_ghost_exchange_communicator.ctx

    def init_domain(self):
        """:110-145 (host arrays)."""
        self.y_range = self.x_range * self.grid_size_y / self.grid_size_x
        self.dx = self.real_t(self.x_range / self.grid_size_x)
        eul_grid_shift = self.dx / 2.0
        ghost_grid_shift = self.ghost_size * self.dx
        local_grid_size = self.mpi_construct.local_grid_size
        substart_idx = self.mpi_construct.grid.coords * local_grid_size
        subend_idx = substart_idx + local_grid_size
        substart_y, substart_x = substart_idx * self.dx
        subend_y, subend_x = subend_idx * self.dx
        ny, nx = local_grid_size
        g = self.ghost_size
        local_x = np.linspace(eul_grid_shift + substart_x - ghost_grid_shift,
                              subend_x - eul_grid_shift + ghost_grid_shift, nx + 2 * g).astype(self.real_t)
        local_y = np.linspace(eul_grid_shift + substart_y - ghost_grid_shift,
                              subend_y - eul_grid_shift + ghost_grid_shift, ny + 2 * g).astype(self.real_t)
        self.position_field = np.flipud(np.array(np.meshgrid(local_y, local_x, indexing="ij")))
        self.local_grid_size_with_ghost = local_grid_size + 2 * g

    def init_fields(self):
        """:147-164 (device arrays)."""
        ctx, shape = self.ctx, tuple(int(n) for n in self.local_grid_size_with_ghost)
        self.primary_scalar_field = ctx.zeros(shape)
        self.velocity_field = ctx.zeros((2,) + shape)
        self.buffer_scalar_field = ctx.zeros(shape)
        if self.flow_type in ["navier_stokes", "navier_stokes_with_forcing"]:
            self.vorticity_field = self.primary_scalar_field.view()
            self.stream_func_field = ctx.zeros(shape)
        if self.flow_type == "navier_stokes_with_forcing":
            self.eul_grid_forcing_field = ctx.zeros((2,) + shape)

    def compile_kernels(self):
        """:166-240."""
        common = dict(real_t=self.real_t, mpi_construct=self.mpi_construct,
                      ghost_exchange_communicator=self.mpi_ghost_exchange_communicator)
        self.diffusion_timestep = gen_diffusion_timestep_euler_forward_pyst_mpi_kernel_2d(**common)
        self.advection_timestep = gen_advection_timestep_euler_forward_conservative_eno3_pyst_mpi_kernel_2d(**common)
        if self.flow_type in ["navier_stokes", "navier_stokes_with_forcing"]:
            self.unbounded_poisson_solver = UnboundedPoissonSolverMPI2D(
                grid_size_y=self.grid_size_y, grid_size_x=self.grid_size_x, x_range=self.x_range,
                real_t=self.real_t, mpi_construct=self.mpi_construct, ghost_size=self.ghost_size)
            self.curl = gen_outplane_field_curl_pyst_mpi_kernel_2d(**common)
            self.penalise_field_towards_boundary = gen_penalise_field_boundary_pyst_mpi_kernel_2d(
                width=self.penalty_zone_width, dx=self.dx, x_grid_field=self.position_field[0],
                y_grid_field=self.position_field[1], **common)
        if self.flow_type == "navier_stokes_with_forcing":
            self.update_vorticity_from_velocity_forcing = (
                gen_update_vorticity_from_velocity_forcing_pyst_mpi_kernel_2d(**common))
            self.set_field = gen_set_fixed_val_pyst_kernel_2d(real_t=self.real_t, field_type="vector")
        if self.with_free_stream_flow:
            add_fixed_val = gen_add_fixed_val_pyst_kernel_2d(real_t=self.real_t, field_type="vector")

            def update_velocity_with_free_stream(free_stream_velocity):
                add_fixed_val(sum_field=self.velocity_field, vector_field=self.velocity_field,
                              fixed_vals=free_stream_velocity)
        else:

            def update_velocity_with_free_stream(free_stream_velocity):
                ...

        self.update_velocity_with_free_stream = update_velocity_with_free_stream

    def finalise_flow_timestep(self):
        self.flow_time_step = self.advection_and_diffusion_timestep
        if self.flow_type == "navier_stokes":
            self.flow_time_step = self.navier_stokes_timestep
        elif self.flow_type == "navier_stokes_with_forcing":
            self.flow_time_step = self.navier_stokes_with_forcing_timestep
        if self.fused_step and self.flow_type in ["navier_stokes", "navier_stokes_with_forcing"]:
            self.flow_time_step = self._fused_navier_stokes_timestep

    def _fused_navier_stokes_timestep(self, dt, free_stream_velocity=(0.0, 0.0)):
        """navier_stokes(_with_forcing)_timestep (:284-298) as one library call."""
        lo = hi = None
        if self.penalty_zone_width > 0:
            lo, hi = self.penalise_field_towards_boundary.factor_tables
        fs = darray(np.asarray(free_stream_velocity, dtype=self.real_t)) if self.with_free_stream_flow else None
        forcing = as_ptr(self.eul_grid_forcing_field) if self.flow_type == "navier_stokes_with_forcing" else None
        call("smi2d_flow_step_ns", self.ctx.handle, self.unbounded_poisson_solver.handle,
             as_ptr(self.vorticity_field), as_ptr(self.velocity_field), as_ptr(self.stream_func_field),
             as_ptr(self.buffer_scalar_field), forcing, float(dt), float(self.kinematic_viscosity), float(self.dx),
             int(self.penalty_zone_width), lo, hi, fs)

    def update_simulator_time(self, dt):
        self.time += dt

    def time_step(self, dt, **kwargs):
        self.flow_time_step(dt=dt, **kwargs)
        self.update_simulator_time(dt=dt)

    def advection_and_diffusion_timestep(self, dt, **kwargs):
        """:260-271."""
        self.advection_timestep(field=self.primary_scalar_field, advection_flux=self.buffer_scalar_field,
                                velocity=self.velocity_field, dt_by_dx=self.real_t(dt / self.dx))
        self.diffusion_timestep(field=self.primary_scalar_field, diffusion_flux=self.buffer_scalar_field,
                                nu_dt_by_dx2=self.real_t(self.kinematic_viscosity * dt / self.dx / self.dx))

    def compute_velocity_from_vorticity(self):
        """:273-282."""
        self.penalise_field_towards_boundary(field=self.vorticity_field)
        self.unbounded_poisson_solver.solve(solution_field=self.stream_func_field, rhs_field=self.vorticity_field)
        self.curl(curl=self.velocity_field, field=self.stream_func_field, prefactor=self.real_t(0.5 / self.dx))

    def navier_stokes_timestep(self, dt, free_stream_velocity=(0.0, 0.0)):
        """:284-287."""
        self.advection_and_diffusion_timestep(dt=dt)
        self.compute_velocity_from_vorticity()
        self.update_velocity_with_free_stream(free_stream_velocity=free_stream_velocity)

    def navier_stokes_with_forcing_timestep(self, dt, free_stream_velocity=(0.0, 0.0)):
        """:289-298."""
        self.update_vorticity_from_velocity_forcing(
            vorticity_field=self.vorticity_field, velocity_forcing_field=self.eul_grid_forcing_field,
            prefactor=self.real_t(dt / (2 * self.dx)))
        self.navier_stokes_timestep(dt=dt, free_stream_velocity=free_stream_velocity)
        self.set_field(vector_field=self.eul_grid_forcing_field, fixed_vals=[0.0] * self.grid_dim)

    def compute_stable_timestep(self, dt_prefac=1, precision="single"):
        """:300-321 with the |u|_1 maximum reduced on the device (no tolerance on the viscosity in 2D)."""
        out = C.c_double(0.0)
        call("smi2d_reduce_max_abs_sum", self.ctx.handle, as_ptr(self.velocity_field), C.byref(out))
        vel_max = self.real_t(out.value)
        dt = min(self.CFL * self.dx / (vel_max + get_test_tol(precision)),
                 0.9 * self.dx**2 / (2 * self.grid_dim) / self.kinematic_viscosity)
        return dt * dt_prefac

    def get_max_vorticity(self):
        """:323-328."""
        out = C.c_double(0.0)
        call("smi2d_reduce_max", self.ctx.handle, as_ptr(self.vorticity_field), C.byref(out))
        return self.real_t(out.value)
