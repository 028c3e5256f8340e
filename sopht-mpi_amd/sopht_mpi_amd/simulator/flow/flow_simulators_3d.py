"""UnboundedFlowSimulator3D on MI355X: the constructor, attributes and methods users of
sopht_mpi/simulator/flow/flow_simulators_mpi_3d.py:24-476 rely on, over device-resident fields.

Fields are DeviceField objects (HBM resident, ghost cells included); ``position_field`` is a host numpy array
since it only serves to set initial conditions.  A step is a list of stages: with ``fused_step=True`` (default)
the Navier-Stokes flow types have a single stage, one library call (smi_flow_step_ns3d); with ``fused_step=False``
the stages are the reference's operators in the reference's order, each through the generator API.
"""
import ctypes as C
import logging

import numpy as np

from ..._lib import call, darray, lib
from ...device import EMPTY_SUPPORT, as_ptr
from ...numeric import eulerian_grid_ops as ops
from ... import sopht_compat as compat
from ...utils import MPIConstruct3D, MPIGhostCommunicator3D
from ._flow_base import FieldSpec, UnboundedFlowSimulatorBase

logger = logging.getLogger("sopht_mpi_amd")

_ALL = ("passive_scalar", "passive_vector", "navier_stokes", "navier_stokes_with_forcing")
_NS = UnboundedFlowSimulatorBase.VORTICITY_FLOW_TYPES
_FILTER_CODES = {"multiplicative": 0, "convolution": 1}


class UnboundedFlowSimulator3D(UnboundedFlowSimulatorBase):
    """3D unbounded flow simulator, one z-slab of the grid per GPU."""

    grid_dim = 3
    FLOW_TYPES = _ALL
    # the arrays of flow_simulators_mpi_3d.py:170-196
    FIELDS = (
        FieldSpec("velocity_field", 3, _ALL),
        FieldSpec("buffer_scalar_field", None, _ALL),
        FieldSpec("primary_scalar_field", None, ("passive_scalar",)),
        FieldSpec("primary_vector_field", 3, _ALL[1:]),
        FieldSpec("vorticity_field", 3, _NS, view_of="primary_vector_field"),
        FieldSpec("stream_func_field", 3, _NS),
        FieldSpec("buffer_vector_field", 3, _NS),
        FieldSpec("eul_grid_forcing_field", 3, ("navier_stokes_with_forcing",)),
    )

    def __init__(self, grid_size, x_range, kinematic_viscosity, time=0.0, CFL=0.1, flow_type="passive_scalar",
                 filter_vorticity=False, real_t=np.float32, rank_distribution=None, ghost_size=2, **kwargs):
        self._configure(grid_size, x_range, kinematic_viscosity, time, CFL, flow_type, real_t, rank_distribution,
                        ghost_size, kwargs)
        self.filter_vorticity = filter_vorticity
        self.poisson_backend = kwargs.get("poisson_backend", None)
        if self.solves_vorticity:
            self.penalty_zone_width = kwargs.get("penalty_zone_width", 2)
            self.with_free_stream_flow = kwargs.get("with_free_stream_flow", False)
            self.filter_setting_dict = self._filter_settings(kwargs.get("filter_setting_dict"))
        self._assemble()

    def _filter_settings(self, given):
        """The reference's defaults (order 2, multiplicative) when filtering is on and no settings are given."""
        if not self.filter_vorticity:
            return None
        logger.warning("Vorticity filtering is turned on.")
        if given is None:
            given = {"order": 2, "type": "multiplicative"}
            logger.warning(f"No filter_setting_dict with keys 'order' and 'type' given: using order "
                           f"{given['order']}, type {given['type']}")
        return given

    def _make_topology(self):
        construct = MPIConstruct3D(grid_size_z=self.grid_size_z, grid_size_y=self.grid_size_y,
                                   grid_size_x=self.grid_size_x, real_t=self.real_t,
                                   rank_distribution=self.rank_distribution, world=self._world)
        # diagonal neighbours matter to the reference's pencils only when bodies spread forcing across them
        halo = MPIGhostCommunicator3D(ghost_size=self.ghost_size, mpi_construct=construct,
                                      full_exchange=self.has_forcing)
        return construct, halo

    # ------------------------------------------------------------------ operators
    def _build_operators(self, common):
        """The operator attributes of flow_simulators_mpi_3d.py:198-325, from one table per flow family:
        (attribute, generator, keyword arguments)."""
        table = []
        if not self.solves_vorticity:
            kind = "scalar" if self.flow_type == "passive_scalar" else "vector"
            table += [
                ("diffusion_timestep", ops.gen_diffusion_timestep_euler_forward_pyst_mpi_kernel_3d,
                 dict(common, field_type=kind)),
                ("advection_timestep", ops.gen_advection_timestep_euler_forward_conservative_eno3_pyst_mpi_kernel_3d,
                 dict(common, field_type=kind)),
            ]
        else:
            x, y, z = self.position_field
            table += [
                ("diffusion_timestep", ops.gen_diffusion_timestep_euler_forward_pyst_mpi_kernel_3d,
                 dict(common, field_type="vector")),
                ("curl", ops.gen_curl_pyst_mpi_kernel_3d, common),
                ("penalise_field_towards_boundary", ops.gen_penalise_field_boundary_pyst_mpi_kernel_3d,
                 dict(common, width=self.penalty_zone_width, dx=self.dx, x_grid_field=x, y_grid_field=y,
                      z_grid_field=z, field_type="vector")),
                ("elementwise_cross_product", compat.gen_elementwise_cross_product_pyst_kernel_3d,
                 dict(real_t=self.real_t)),
                ("update_vorticity_from_velocity_forcing",
                 ops.gen_update_vorticity_from_velocity_forcing_pyst_mpi_kernel_3d, common),
                ("compute_divergence", ops.gen_divergence_pyst_mpi_kernel_3d, common),
            ]
            if self.filter_setting_dict is not None:
                table.append(("filter_vector_field", ops.gen_laplacian_filter_mpi_kernel_3d, dict(
                    common, filter_order=self.filter_setting_dict["order"],
                    filter_type=self.filter_setting_dict["type"], field_type="vector",
                    filter_flux_buffer=self.buffer_vector_field[0], field_buffer=self.buffer_vector_field[1])))
            if self.has_forcing:
                table.append(("set_field", compat.gen_set_fixed_val_pyst_kernel_3d,
                              dict(real_t=self.real_t, field_type="vector")))
            if self.with_free_stream_flow:
                table.append(("_add_fixed_val", compat.gen_add_fixed_val_pyst_kernel_3d,
                              dict(real_t=self.real_t, field_type="vector")))
        for name, generator, kwargs in table:
            setattr(self, name, generator(**kwargs))
        if self.solves_vorticity:
            self.unbounded_poisson_solver = ops.UnboundedPoissonSolverMPI3D(
                grid_size_z=self.grid_size_z, grid_size_y=self.grid_size_y, grid_size_x=self.grid_size_x,
                x_range=self.x_range, real_t=self.real_t, mpi_construct=self.mpi_construct,
                ghost_size=self.ghost_size, backend=self.poisson_backend)

    def update_velocity_with_free_stream(self, free_stream_velocity):
        if self.with_free_stream_flow:
            self._add_fixed_val(sum_field=self.velocity_field, vector_field=self.velocity_field,
                                fixed_vals=free_stream_velocity)

    # ------------------------------------------------------------------ stages of a step
    def _stages(self):
        if not self.solves_vorticity:
            return [self._advect_primary_field, self._diffuse_primary_field]
        if self.fused_step:
            return [self._fused_navier_stokes_step]
        # flow_simulators_mpi_3d.py:395-424 in order
        stages = [self._apply_forcing] if self.has_forcing else []
        stages += [self._advect_vorticity_rotational_form, self._diffuse_vorticity]
        if self.filter_setting_dict is not None:
            stages.append(lambda dt, **_: self.filter_vector_field(vector_field=self.vorticity_field))
        stages.append(lambda dt, free_stream_velocity=None: self.compute_flow_velocity(free_stream_velocity))
        if self.has_forcing:
            stages.append(lambda dt, **_: self.set_field(vector_field=self.eul_grid_forcing_field,
                                                         fixed_vals=[0.0] * self.grid_dim))
        return stages

    def _named_stage_lists(self):
        """flow_simulators_mpi_3d.py:356-424 by name.  The plain Navier-Stokes step of a simulator with forcing is
        the operator sequence without the forcing update and reset (the one-call step always includes them)."""
        if not self.solves_vorticity:
            name = "scalar" if self.flow_type == "passive_scalar" else "vector"
            return {f"{name}_advection_and_diffusion_timestep": self._step_stages}
        if not self.has_forcing:
            plain = self._step_stages
        else:
            plain = [self._advect_vorticity_rotational_form, self._diffuse_vorticity]
            if self.filter_setting_dict is not None:
                plain.append(lambda dt, **_: self.filter_vector_field(vector_field=self.vorticity_field))
            plain.append(lambda dt, free_stream_velocity=None: self.compute_flow_velocity(free_stream_velocity))
        names = {"rotational_form_navier_stokes_timestep": plain, "navier_stokes_timestep": plain}
        if self.has_forcing:
            names["navier_stokes_with_forcing_timestep"] = self._step_stages
        return names

    def _primary(self):
        if self.flow_type == "passive_scalar":
            return {"field": self.primary_scalar_field}
        return {"vector_field": self.primary_vector_field}

    def _advect_primary_field(self, dt, **_):
        """ENO3 advection of the transported field (:356-380, first half)."""
        self.advection_timestep(advection_flux=self.buffer_scalar_field, velocity=self.velocity_field,
                                dt_by_dx=self._scaled(dt / self.dx), **self._primary())

    def _diffuse_primary_field(self, dt, **_):
        self.diffusion_timestep(diffusion_flux=self.buffer_scalar_field,
                                nu_dt_by_dx2=self._scaled(self.kinematic_viscosity * dt / self.dx / self.dx),
                                **self._primary())

    def _apply_forcing(self, dt, **_):
        """w += dt / (2 dx) curl(f) (:415-421)."""
        self.update_vorticity_from_velocity_forcing(
            vorticity_field=self.vorticity_field, velocity_forcing_field=self.eul_grid_forcing_field,
            prefactor=self._scaled(dt / (2 * self.dx)))

    def _advect_vorticity_rotational_form(self, dt, **_):
        """w += dt / (2 dx) curl(u x w) (:395-406)."""
        u_cross_w = self.buffer_vector_field.view()
        self.elementwise_cross_product(result_field=u_cross_w, field_1=self.velocity_field,
                                       field_2=self.vorticity_field)
        self.update_vorticity_from_velocity_forcing(vorticity_field=self.vorticity_field,
                                                    velocity_forcing_field=u_cross_w,
                                                    prefactor=self._scaled(dt / (2 * self.dx)))

    def _diffuse_vorticity(self, dt, **_):
        self.diffusion_timestep(vector_field=self.vorticity_field, diffusion_flux=self.buffer_scalar_field,
                                nu_dt_by_dx2=self._scaled(self.kinematic_viscosity * dt / self.dx / self.dx))

    def compute_flow_velocity(self, free_stream_velocity):
        """Vorticity -> velocity: penalise towards the boundary, three Poisson solves, curl, free stream
        (flow_simulators_mpi_3d.py:382-393)."""
        self.penalise_field_towards_boundary(vector_field=self.vorticity_field)
        self.unbounded_poisson_solver.vector_field_solve(solution_vector_field=self.stream_func_field,
                                                         rhs_vector_field=self.vorticity_field)
        self.curl(curl=self.velocity_field, field=self.stream_func_field, prefactor=self._scaled(0.5 / self.dx))
        self.update_velocity_with_free_stream(free_stream_velocity=free_stream_velocity)

    def _fused_navier_stokes_step(self, dt, free_stream_velocity=None):
        """The whole Navier-Stokes sequence as one library call (smi_flow_step_ns3d)."""
        lo = hi = None
        if self.penalty_zone_width > 0:
            lo, hi = (darray(t) for t in self.penalise_field_towards_boundary.factor_tables)
        free_stream = None
        if self.with_free_stream_flow:
            free_stream = darray(np.asarray(free_stream_velocity, dtype=self.real_t))
        forcing, forcing_field = None, None
        if self.has_forcing:
            forcing_field = self.eul_grid_forcing_field
            forcing = forcing_field.tracked_ptr()
            support = forcing_field.support
            if support is not None and self.mpi_construct.size == 1:
                # the forcing is known to vanish outside a box (only immersed-boundary spreads have written to
                # it since the last reset): the step sweeps that box instead of the grid
                box_lo, box_hi = ((1, 1, 1), (0, 0, 0)) if support == EMPTY_SUPPORT else support
                call("smi_set_forcing_support", self.ctx.handle, (C.c_int * 3)(*box_lo), (C.c_int * 3)(*box_hi))
        order, ftype = 0, 0
        if self.filter_setting_dict is not None:
            order, ftype = int(self.filter_setting_dict["order"]), _FILTER_CODES[self.filter_setting_dict["type"]]
        result_in_buffer = C.c_int(0)
        try:
            call("smi_flow_step_ns3d_swap", self.ctx.handle, self.unbounded_poisson_solver.handle,
                 as_ptr(self.vorticity_field), as_ptr(self.velocity_field), as_ptr(self.stream_func_field),
                 as_ptr(self.buffer_vector_field), as_ptr(self.buffer_scalar_field), forcing, float(dt),
                 float(self.kinematic_viscosity), float(self.dx), int(self.penalty_zone_width), lo, hi, free_stream,
                 order, ftype, C.byref(result_in_buffer))
        except Exception:
            # a step that failed part-way says nothing about where the forcing is non-zero
            if forcing_field is not None:
                forcing_field._alloc.support = None
            raise
        if result_in_buffer.value:
            # the step left the new vorticity in the buffer array: the two allocations trade places (every view
            # of either field shares its allocation object and follows)
            self.vorticity_field.exchange_storage_with(self.buffer_vector_field)
        if forcing_field is not None:
            forcing_field._alloc.support = EMPTY_SUPPORT   # the step ended with the reset
        # the step's curl kernel has gathered max |u|_1 of the velocity it wrote (compute_stable_timestep)
        self.velocity_field._alloc.derived.add("step_max_abs_sum")

    # ------------------------------------------------------------------ diagnostics
    def compute_stable_timestep(self, dt_prefac=1, precision="single"):
        """The CFL / diffusion limit of flow_simulators_mpi_3d.py:426-449, with max |u|_1 taken from the last
        fused step's curl kernel when the velocity has not been written since, else reduced on the device."""
        out = C.c_double(0.0)
        fresh = "step_max_abs_sum" in self.velocity_field._alloc.derived
        if fresh and lib.smi_step_velocity_max_abs_sum(self.ctx.handle, self.velocity_field.tracked_ptr(),
                                                       C.byref(out)) == 0:
            vel_max = out.value
        else:
            vel_max = self._reduce("smi_reduce_max_abs_sum", self.velocity_field.tracked_ptr())
        return self._stable_timestep(vel_max, dt_prefac, precision, viscosity_tol=True)

    def get_vorticity_divergence_l2_norm(self):
        """sqrt(sum div(w)^2) dx^1.5 over all interiors (flow_simulators_mpi_3d.py:451-469)."""
        self.compute_divergence(divergence=self.buffer_scalar_field, field=self.vorticity_field,
                                inv_dx=(1.0 / self.dx))
        local = self._reduce("smi_reduce_sum_sq", as_ptr(self.buffer_scalar_field), 1)
        return np.sqrt(self.ctx.allreduce(local, "sum")) * self.dx**1.5

    def get_max_vorticity(self):
        """Largest vorticity component over all interiors (flow_simulators_mpi_3d.py:471-476)."""
        local = self._reduce("smi_reduce_max", as_ptr(self.vorticity_field), 3)
        return self.real_t(self.ctx.allreduce(local, "max"))
