"""UnboundedFlowSimulator3D on MI355X, same surface as
sopht_mpi/simulator/flow/flow_simulators_mpi_3d.py:24-476.

Fields are DeviceField objects (HBM resident, ghost cells included); ``position_field`` is a
host numpy array since it is only used to set initial conditions.  ``time_step`` runs
either operator by operator (``fused_step=False``, the reference's call sequence through
the generator API) or as one library call (default).
"""
import ctypes as C
import logging

import numpy as np

from ..._lib import call, darray, lib
from ...device import EMPTY_SUPPORT, as_ptr
from ...numeric.eulerian_grid_ops import (
    UnboundedPoissonSolverMPI3D,
    gen_advection_timestep_euler_forward_conservative_eno3_pyst_mpi_kernel_3d,
    gen_curl_pyst_mpi_kernel_3d,
    gen_diffusion_timestep_euler_forward_pyst_mpi_kernel_3d,
    gen_divergence_pyst_mpi_kernel_3d,
    gen_laplacian_filter_mpi_kernel_3d,
    gen_penalise_field_boundary_pyst_mpi_kernel_3d,
    gen_update_vorticity_from_velocity_forcing_pyst_mpi_kernel_3d,
)
from ...sopht_compat import (
    VectorField,
    gen_add_fixed_val_pyst_kernel_3d,
    gen_elementwise_cross_product_pyst_kernel_3d,
    gen_set_fixed_val_pyst_kernel_3d,
    get_test_tol,
)
from ...utils import MPIConstruct3D, MPIGhostCommunicator3D

logger = logging.getLogger("sopht_mpi_amd")


class UnboundedFlowSimulator3D:
    """Class for 3D unbounded flow simulator, one z-slab per GPU."""

    def __init__(self, grid_size, x_range, kinematic_viscosity, time=0.0, CFL=0.1, flow_type="passive_scalar",
                 filter_vorticity=False, real_t=np.float32, rank_distribution=None, ghost_size=2, **kwargs):
        self.grid_dim = 3
        self.grid_size = grid_size
        self.grid_size_z, self.grid_size_y, self.grid_size_x = self.grid_size
        self.x_range = x_range
        self.real_t = real_t
        self.flow_type = flow_type
        self.kinematic_viscosity = kinematic_viscosity
        self.CFL = CFL
        self.time = time
        self.filter_vorticity = filter_vorticity
        if self.flow_type not in ["passive_scalar", "passive_vector", "navier_stokes", "navier_stokes_with_forcing"]:
            raise ValueError("Invalid flow type given")
        self.rank_distribution = rank_distribution
        self.ghost_size = ghost_size
        self.fused_step = kwargs.get("fused_step", True)
        self.poisson_backend = kwargs.get("poisson_backend", None)
        self._world = kwargs.get("world", None)

        self.init_mpi()
        self.init_domain()
        self.init_fields()
        if self.flow_type in ["navier_stokes", "navier_stokes_with_forcing"]:
            self.penalty_zone_width = kwargs.get("penalty_zone_width", 2)
            self.with_free_stream_flow = kwargs.get("with_free_stream_flow", False)
            self.filter_setting_dict = None
            if self.filter_vorticity:
                logger.warning("Vorticity filtering is turned on.")
                self.filter_setting_dict = kwargs.get("filter_setting_dict")
                if self.filter_setting_dict is None:
                    self.filter_setting_dict = {"order": 2, "type": "multiplicative"}
                    logger.warning(
                        "Since a dict named filter_setting with keys 'order' and 'type' is not provided, setting "
                        f"default filter order = {self.filter_setting_dict['order']} "
                        f"and type: {self.filter_setting_dict['type']}"
                    )
        self.compile_kernels()
        self.finalise_flow_timestep()

    # ------------------------------------------------------------------ set-up
    def init_mpi(self):
        """flow_simulators_mpi_3d.py:109-122."""
        self.mpi_construct = MPIConstruct3D(
            grid_size_z=self.grid_size_z, grid_size_y=self.grid_size_y, grid_size_x=self.grid_size_x,
            real_t=self.real_t, rank_distribution=self.rank_distribution, world=self._world,
        )
        need_full_exchange = self.flow_type == "navier_stokes_with_forcing"
        self.mpi_ghost_exchange_communicator = MPIGhostCommunicator3D(
            ghost_size=self.ghost_size, mpi_construct=self.mpi_construct, full_exchange=need_full_exchange
        )
        self.ctx = self.mpi_ghost_exchange_communicator.ctx

    def init_domain(self):
        """flow_simulators_mpi_3d.py:124-168 (host arrays)."""
        self.y_range = self.x_range * self.grid_size_y / self.grid_size_x
        self.z_range = self.x_range * self.grid_size_z / self.grid_size_x
        self.dx = self.real_t(self.x_range / self.grid_size_x)
        eul_grid_shift = self.dx / 2.0
        ghost_grid_shift = self.ghost_size * self.dx
        local_grid_size = self.mpi_construct.local_grid_size
        substart_idx = self.mpi_construct.grid.coords * local_grid_size
        subend_idx = substart_idx + local_grid_size
        substart_z, substart_y, substart_x = substart_idx * self.dx
        subend_z, subend_y, subend_x = subend_idx * self.dx
        nz, ny, nx = local_grid_size
        g = self.ghost_size
        local_x = np.linspace(eul_grid_shift + substart_x - ghost_grid_shift,
                              subend_x - eul_grid_shift + ghost_grid_shift, nx + 2 * g).astype(self.real_t)
        local_y = np.linspace(eul_grid_shift + substart_y - ghost_grid_shift,
                              subend_y - eul_grid_shift + ghost_grid_shift, ny + 2 * g).astype(self.real_t)
        local_z = np.linspace(eul_grid_shift + substart_z - ghost_grid_shift,
                              subend_z - eul_grid_shift + ghost_grid_shift, nz + 2 * g).astype(self.real_t)
        self.position_field = np.flipud(np.array(np.meshgrid(local_z, local_y, local_x, indexing="ij")))
        self.local_grid_size_with_ghost = local_grid_size + 2 * g
        logger.info(
            f"{self.grid_dim}D flow domain initialized with: X axis from 0.0 to {self.x_range}, "
            f"Y axis from 0.0 to {self.y_range}, Z axis from 0.0 to {self.z_range}"
        )

    def init_fields(self):
        """flow_simulators_mpi_3d.py:170-196 (device arrays)."""
        ctx = self.ctx
        self.velocity_field = ctx.field(3)
        self.buffer_scalar_field = ctx.field()
        if self.flow_type == "passive_scalar":
            self.primary_scalar_field = ctx.field()
        else:
            self.primary_vector_field = ctx.field(3)
        if self.flow_type in ["navier_stokes", "navier_stokes_with_forcing"]:
            self.vorticity_field = self.primary_vector_field.view()
            self.stream_func_field = ctx.field(3)
            self.buffer_vector_field = ctx.field(3)
        if self.flow_type == "navier_stokes_with_forcing":
            self.eul_grid_forcing_field = ctx.field(3)

    def compile_kernels(self):
        """flow_simulators_mpi_3d.py:198-325."""
        common = dict(mpi_construct=self.mpi_construct, ghost_exchange_communicator=self.mpi_ghost_exchange_communicator,
                      real_t=self.real_t)
        if self.flow_type in ("passive_scalar", "passive_vector"):
            ft = "scalar" if self.flow_type == "passive_scalar" else "vector"
            self.diffusion_timestep = gen_diffusion_timestep_euler_forward_pyst_mpi_kernel_3d(field_type=ft, **common)
            self.advection_timestep = gen_advection_timestep_euler_forward_conservative_eno3_pyst_mpi_kernel_3d(
                field_type=ft, **common
            )
        if self.flow_type in ["navier_stokes", "navier_stokes_with_forcing"]:
            self.diffusion_timestep = gen_diffusion_timestep_euler_forward_pyst_mpi_kernel_3d(
                field_type="vector", **common
            )
            self.unbounded_poisson_solver = UnboundedPoissonSolverMPI3D(
                grid_size_z=self.grid_size_z, grid_size_y=self.grid_size_y, grid_size_x=self.grid_size_x,
                x_range=self.x_range, real_t=self.real_t, mpi_construct=self.mpi_construct,
                ghost_size=self.ghost_size, backend=self.poisson_backend,
            )
            self.curl = gen_curl_pyst_mpi_kernel_3d(**common)
            self.penalise_field_towards_boundary = gen_penalise_field_boundary_pyst_mpi_kernel_3d(
                width=self.penalty_zone_width, dx=self.dx,
                x_grid_field=self.position_field[VectorField.x_axis_idx()],
                y_grid_field=self.position_field[VectorField.y_axis_idx()],
                z_grid_field=self.position_field[VectorField.z_axis_idx()],
                field_type="vector", **common,
            )
            self.elementwise_cross_product = gen_elementwise_cross_product_pyst_kernel_3d(real_t=self.real_t)
            self.update_vorticity_from_velocity_forcing = (
                gen_update_vorticity_from_velocity_forcing_pyst_mpi_kernel_3d(**common)
            )
            self.compute_divergence = gen_divergence_pyst_mpi_kernel_3d(**common)

            def filter_vector_field(vector_field):
                ...

            self.filter_vector_field = filter_vector_field
            if self.filter_vorticity and self.filter_setting_dict is not None:
                self.filter_vector_field = gen_laplacian_filter_mpi_kernel_3d(
                    mpi_construct=self.mpi_construct,
                    ghost_exchange_communicator=self.mpi_ghost_exchange_communicator,
                    filter_order=self.filter_setting_dict["order"],
                    filter_flux_buffer=self.buffer_vector_field[0],
                    field_buffer=self.buffer_vector_field[1],
                    real_t=self.real_t, field_type="vector", filter_type=self.filter_setting_dict["type"],
                )
        if self.flow_type == "navier_stokes_with_forcing":
            self.set_field = gen_set_fixed_val_pyst_kernel_3d(real_t=self.real_t, field_type="vector")
        if self.flow_type in ["navier_stokes", "navier_stokes_with_forcing"]:
            if self.with_free_stream_flow:
                add_fixed_val = gen_add_fixed_val_pyst_kernel_3d(real_t=self.real_t, field_type="vector")

                def update_velocity_with_free_stream(free_stream_velocity):
                    add_fixed_val(sum_field=self.velocity_field, vector_field=self.velocity_field,
                                  fixed_vals=free_stream_velocity)

            else:

                def update_velocity_with_free_stream(free_stream_velocity):
                    ...

            self.update_velocity_with_free_stream = update_velocity_with_free_stream

    def finalise_flow_timestep(self):
        """flow_simulators_mpi_3d.py:327-345."""
        self.navier_stokes_timestep = self.rotational_form_navier_stokes_timestep
        self.flow_time_step = self.scalar_advection_and_diffusion_timestep
        if self.flow_type == "passive_vector":
            self.flow_time_step = self.vector_advection_and_diffusion_timestep
        elif self.flow_type == "navier_stokes":
            self.flow_time_step = self.navier_stokes_timestep
        elif self.flow_type == "navier_stokes_with_forcing":
            self.flow_time_step = self.navier_stokes_with_forcing_timestep
        if self.fused_step and self.flow_type in ["navier_stokes", "navier_stokes_with_forcing"]:
            self.flow_time_step = self._fused_navier_stokes_timestep

    # ------------------------------------------------------------------ stepping
    def update_simulator_time(self, dt):
        self.time += dt

    def time_step(self, dt, **kwargs):
        """Final simulator time step."""
        self.flow_time_step(dt=dt, **kwargs)
        self.update_simulator_time(dt=dt)

    def scalar_advection_and_diffusion_timestep(self, dt, **kwargs):
        self.advection_timestep(field=self.primary_scalar_field, advection_flux=self.buffer_scalar_field,
                                velocity=self.velocity_field, dt_by_dx=self.real_t(dt / self.dx))
        self.diffusion_timestep(field=self.primary_scalar_field, diffusion_flux=self.buffer_scalar_field,
                                nu_dt_by_dx2=self.real_t(self.kinematic_viscosity * dt / self.dx / self.dx))

    def vector_advection_and_diffusion_timestep(self, dt, **kwargs):
        self.advection_timestep(vector_field=self.primary_vector_field, advection_flux=self.buffer_scalar_field,
                                velocity=self.velocity_field, dt_by_dx=self.real_t(dt / self.dx))
        self.diffusion_timestep(vector_field=self.primary_vector_field, diffusion_flux=self.buffer_scalar_field,
                                nu_dt_by_dx2=self.real_t(self.kinematic_viscosity * dt / self.dx / self.dx))

    def compute_flow_velocity(self, free_stream_velocity):
        """flow_simulators_mpi_3d.py:382-393."""
        self.penalise_field_towards_boundary(vector_field=self.vorticity_field)
        self.unbounded_poisson_solver.vector_field_solve(
            solution_vector_field=self.stream_func_field, rhs_vector_field=self.vorticity_field
        )
        self.curl(curl=self.velocity_field, field=self.stream_func_field, prefactor=self.real_t(0.5 / self.dx))
        self.update_velocity_with_free_stream(free_stream_velocity=free_stream_velocity)

    def rotational_form_navier_stokes_timestep(self, dt, free_stream_velocity=None):
        """flow_simulators_mpi_3d.py:395-413."""
        velocity_cross_vorticity = self.buffer_vector_field.view()
        self.elementwise_cross_product(result_field=velocity_cross_vorticity, field_1=self.velocity_field,
                                       field_2=self.vorticity_field)
        self.update_vorticity_from_velocity_forcing(
            vorticity_field=self.vorticity_field, velocity_forcing_field=velocity_cross_vorticity,
            prefactor=self.real_t(dt / (2 * self.dx)),
        )
        self.diffusion_timestep(vector_field=self.vorticity_field, diffusion_flux=self.buffer_scalar_field,
                                nu_dt_by_dx2=self.real_t(self.kinematic_viscosity * dt / self.dx / self.dx))
        self.filter_vector_field(vector_field=self.vorticity_field)
        self.compute_flow_velocity(free_stream_velocity=free_stream_velocity)

    def navier_stokes_with_forcing_timestep(self, dt, free_stream_velocity=None):
        """flow_simulators_mpi_3d.py:415-424."""
        self.update_vorticity_from_velocity_forcing(
            vorticity_field=self.vorticity_field, velocity_forcing_field=self.eul_grid_forcing_field,
            prefactor=self.real_t(dt / (2 * self.dx)),
        )
        self.rotational_form_navier_stokes_timestep(dt=dt, free_stream_velocity=free_stream_velocity)
        self.set_field(vector_field=self.eul_grid_forcing_field, fixed_vals=[0.0] * self.grid_dim)

    def _fused_navier_stokes_timestep(self, dt, free_stream_velocity=None):
        """The same sequence as one library call (smi_flow_step_ns3d)."""
        lo = hi = None
        if self.penalty_zone_width > 0:
            lo_np, hi_np = self.penalise_field_towards_boundary.factor_tables
            lo, hi = darray(lo_np), darray(hi_np)
        fs = None
        if self.with_free_stream_flow:
            fs = darray(np.asarray(free_stream_velocity, dtype=self.real_t))
        forcing = None
        if self.flow_type == "navier_stokes_with_forcing":
            field = self.eul_grid_forcing_field
            support = field.support
            forcing = field.tracked_ptr()
            if support is not None and self.mpi_construct.size == 1:
                # the forcing is known to vanish outside a box (only immersed-boundary spreads have written to
                # it since the last reset): the step sweeps that box instead of the grid
                box_lo, box_hi = ((1, 1, 1), (0, 0, 0)) if support == EMPTY_SUPPORT else support
                call("smi_set_forcing_support", self.ctx.handle, (C.c_int * 3)(*box_lo), (C.c_int * 3)(*box_hi))
            field._alloc.support = EMPTY_SUPPORT  # the step ends with the reset
        order, ftype = 0, 0
        if self.filter_vorticity and self.filter_setting_dict is not None:
            order = int(self.filter_setting_dict["order"])
            ftype = {"multiplicative": 0, "convolution": 1}[self.filter_setting_dict["type"]]
        call("smi_flow_step_ns3d", self.ctx.handle, self.unbounded_poisson_solver.handle,
             as_ptr(self.vorticity_field), as_ptr(self.velocity_field), as_ptr(self.stream_func_field),
             as_ptr(self.buffer_vector_field), as_ptr(self.buffer_scalar_field), forcing, float(dt),
             float(self.kinematic_viscosity), float(self.dx), int(self.penalty_zone_width), lo, hi, fs, order, ftype)
        # the step's curl kernel has gathered max |u|_1 of the velocity it wrote (compute_stable_timestep)
        self.velocity_field._alloc.derived.add("step_max_abs_sum")

    # ------------------------------------------------------------------ diagnostics
    def compute_stable_timestep(self, dt_prefac=1, precision="single"):
        """flow_simulators_mpi_3d.py:426-449 with the |u|_1 maximum reduced on the device."""
        tol = get_test_tol(precision)
        out = C.c_double(0.0)
        fresh = "step_max_abs_sum" in self.velocity_field._alloc.derived
        if not fresh or lib.smi_step_velocity_max_abs_sum(self.ctx.handle, C.byref(out)) != 0:
            # the velocity has been written since the last fused step (or there was none): reduce the field
            call("smi_reduce_max_abs_sum", self.ctx.handle, self.velocity_field.tracked_ptr(), C.byref(out))
        vel_max = self.real_t(out.value)
        dt = min(
            self.CFL * self.dx / (vel_max + tol),
            0.9 * self.dx**2 / (2 * self.grid_dim) / (self.kinematic_viscosity + tol),
        )
        dt = self.ctx.allreduce(float(dt), "min")
        return dt * dt_prefac

    def get_vorticity_divergence_l2_norm(self):
        """flow_simulators_mpi_3d.py:451-469."""
        self.compute_divergence(divergence=self.buffer_scalar_field, field=self.vorticity_field,
                                inv_dx=(1.0 / self.dx))
        out = C.c_double(0.0)
        call("smi_reduce_sum_sq", self.ctx.handle, as_ptr(self.buffer_scalar_field), 1, C.byref(out))
        total = self.ctx.allreduce(out.value, "sum")
        return np.sqrt(total) * self.dx**1.5

    def get_max_vorticity(self):
        """flow_simulators_mpi_3d.py:471-476."""
        out = C.c_double(0.0)
        call("smi_reduce_max", self.ctx.handle, as_ptr(self.vorticity_field), 3, C.byref(out))
        return self.real_t(self.ctx.allreduce(out.value, "max"))
