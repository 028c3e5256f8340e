"""UnboundedFlowSimulator2D (one MI355X, or row slabs over several): the constructor, attributes and methods users of
sopht_mpi/simulator/flow/flow_simulators_mpi_2d.py:20-328 rely on.  Same construction as the 3D simulator
(tables of fields and operators, a step as a list of stages); the 2D Navier-Stokes step advects with the ENO3
flux instead of the rotational form and has no vorticity filter."""
import numpy as np

from ..._lib import call, darray
from ...device import as_ptr
from ...numeric import eulerian_grid_ops as ops
from ...utils import MPIConstruct2D, MPIGhostCommunicator2D
from ._flow_base import FieldSpec, UnboundedFlowSimulatorBase

_ALL = ("passive_scalar", "navier_stokes", "navier_stokes_with_forcing")
_NS = UnboundedFlowSimulatorBase.VORTICITY_FLOW_TYPES


class UnboundedFlowSimulator2D(UnboundedFlowSimulatorBase):
    """2D unbounded flow simulator."""

    grid_dim = 2
    FLOW_TYPES = _ALL
    # the arrays of flow_simulators_mpi_2d.py:147-164
    FIELDS = (
        FieldSpec("primary_scalar_field", None, _ALL),
        FieldSpec("velocity_field", 2, _ALL),
        FieldSpec("buffer_scalar_field", None, _ALL),
        FieldSpec("vorticity_field", None, _NS, view_of="primary_scalar_field"),
        FieldSpec("stream_func_field", None, _NS),
        FieldSpec("eul_grid_forcing_field", 2, ("navier_stokes_with_forcing",)),
    )

    def __init__(self, grid_size, x_range, kinematic_viscosity, time=0.0, CFL=0.1, flow_type="passive_scalar",
                 with_free_stream_flow=False, real_t=np.float32, rank_distribution=None, ghost_size=2, **kwargs):
        self._configure(grid_size, x_range, kinematic_viscosity, time, CFL, flow_type, real_t, rank_distribution,
                        ghost_size, kwargs)
        if flow_type == "passive_scalar" and with_free_stream_flow:
            raise ValueError("Free stream flow not defined for passive advection diffusion!")
        self.with_free_stream_flow = with_free_stream_flow
        if self.solves_vorticity:
            self.penalty_zone_width = kwargs.get("penalty_zone_width", 2)
        self._assemble()

    def _make_topology(self):
        construct = MPIConstruct2D(grid_size_y=self.grid_size_y, grid_size_x=self.grid_size_x, real_t=self.real_t,
                                   rank_distribution=self.rank_distribution, world=self._world)
        return construct, MPIGhostCommunicator2D(ghost_size=self.ghost_size, mpi_construct=construct,
                                                 full_exchange=False)

    # ------------------------------------------------------------------ operators
    def _build_operators(self, common):
        """The operator attributes of flow_simulators_mpi_2d.py:166-240: (attribute, generator, keyword arguments)."""
        table = [
            ("diffusion_timestep", ops.gen_diffusion_timestep_euler_forward_pyst_mpi_kernel_2d, common),
            ("advection_timestep", ops.gen_advection_timestep_euler_forward_conservative_eno3_pyst_mpi_kernel_2d,
             common),
        ]
        if self.solves_vorticity:
            x, y = self.position_field
            table += [
                ("curl", ops.gen_outplane_field_curl_pyst_mpi_kernel_2d, common),
                ("penalise_field_towards_boundary", ops.gen_penalise_field_boundary_pyst_mpi_kernel_2d,
                 dict(common, width=self.penalty_zone_width, dx=self.dx, x_grid_field=x, y_grid_field=y)),
            ]
        if self.has_forcing:
            table += [
                ("update_vorticity_from_velocity_forcing",
                 ops.gen_update_vorticity_from_velocity_forcing_pyst_mpi_kernel_2d, common),
                ("set_field", ops.gen_set_fixed_val_pyst_kernel_2d, dict(real_t=self.real_t, field_type="vector")),
            ]
        if self.with_free_stream_flow:
            table.append(("_add_fixed_val", ops.gen_add_fixed_val_pyst_kernel_2d,
                          dict(real_t=self.real_t, field_type="vector")))
        for name, generator, kwargs in table:
            setattr(self, name, generator(**kwargs))
        if self.solves_vorticity:
            self.unbounded_poisson_solver = ops.UnboundedPoissonSolverMPI2D(
                grid_size_y=self.grid_size_y, grid_size_x=self.grid_size_x, x_range=self.x_range, real_t=self.real_t,
                mpi_construct=self.mpi_construct, ghost_size=self.ghost_size)

    def update_velocity_with_free_stream(self, free_stream_velocity):
        if self.with_free_stream_flow:
            self._add_fixed_val(sum_field=self.velocity_field, vector_field=self.velocity_field,
                                fixed_vals=free_stream_velocity)

    # ------------------------------------------------------------------ stages of a step
    def _stages(self):
        if self.solves_vorticity and self.fused_step:
            return [self._fused_navier_stokes_step]
        # flow_simulators_mpi_2d.py:260-298 in order
        stages = [self._apply_forcing] if self.has_forcing else []
        stages += [self._advect, self._diffuse]
        if self.solves_vorticity:
            stages += [
                lambda dt, **_: self.compute_velocity_from_vorticity(),
                lambda dt, free_stream_velocity=(0.0, 0.0): self.update_velocity_with_free_stream(free_stream_velocity),
            ]
        if self.has_forcing:
            stages.append(lambda dt, **_: self.set_field(vector_field=self.eul_grid_forcing_field,
                                                         fixed_vals=[0.0] * self.grid_dim))
        return stages

    def _named_stage_lists(self):
        """flow_simulators_mpi_2d.py:260-298 by name."""
        transport = [self._advect, self._diffuse]
        names = {"advection_and_diffusion_timestep": transport}
        if self.solves_vorticity:
            plain = self._step_stages if not self.has_forcing else transport + [
                lambda dt, **_: self.compute_velocity_from_vorticity(),
                lambda dt, free_stream_velocity=(0.0, 0.0): self.update_velocity_with_free_stream(free_stream_velocity)]
            names["navier_stokes_timestep"] = plain
        if self.has_forcing:
            names["navier_stokes_with_forcing_timestep"] = self._step_stages
        return names

    def _apply_forcing(self, dt, **_):
        self.update_vorticity_from_velocity_forcing(
            vorticity_field=self.vorticity_field, velocity_forcing_field=self.eul_grid_forcing_field,
            prefactor=self._scaled(dt / (2 * self.dx)))

    def _advect(self, dt, **_):
        self.advection_timestep(field=self.primary_scalar_field, advection_flux=self.buffer_scalar_field,
                                velocity=self.velocity_field, dt_by_dx=self._scaled(dt / self.dx))

    def _diffuse(self, dt, **_):
        self.diffusion_timestep(field=self.primary_scalar_field, diffusion_flux=self.buffer_scalar_field,
                                nu_dt_by_dx2=self._scaled(self.kinematic_viscosity * dt / self.dx / self.dx))

    def compute_velocity_from_vorticity(self):
        """Vorticity -> velocity: penalise, Poisson solve, out-of-plane curl (flow_simulators_mpi_2d.py:273-282)."""
        self.penalise_field_towards_boundary(field=self.vorticity_field)
        self.unbounded_poisson_solver.solve(solution_field=self.stream_func_field, rhs_field=self.vorticity_field)
        self.curl(curl=self.velocity_field, field=self.stream_func_field, prefactor=self._scaled(0.5 / self.dx))

    def _fused_navier_stokes_step(self, dt, free_stream_velocity=(0.0, 0.0)):
        """The whole 2D Navier-Stokes sequence (:284-298) as one library call (smi2d_flow_step_ns)."""
        lo, hi = self.penalise_field_towards_boundary.factor_tables if self.penalty_zone_width > 0 else (None, None)
        free_stream = None
        if self.with_free_stream_flow:
            free_stream = darray(np.asarray(free_stream_velocity, dtype=self.real_t))
        forcing = as_ptr(self.eul_grid_forcing_field) if self.has_forcing else None
        call("smi2d_flow_step_ns", self.ctx.handle, self.unbounded_poisson_solver.handle,
             as_ptr(self.vorticity_field), as_ptr(self.velocity_field), as_ptr(self.stream_func_field),
             as_ptr(self.buffer_scalar_field), forcing, float(dt), float(self.kinematic_viscosity), float(self.dx),
             int(self.penalty_zone_width), lo, hi, free_stream)

    # ------------------------------------------------------------------ diagnostics
    def compute_stable_timestep(self, dt_prefac=1, precision="single"):
        """The CFL / diffusion limit of flow_simulators_mpi_2d.py:300-321, max |u|_1 reduced on the device."""
        vel_max = self._reduce("smi2d_reduce_max_abs_sum", as_ptr(self.velocity_field))
        return self._stable_timestep(vel_max, dt_prefac, precision, viscosity_tol=False)

    def get_max_vorticity(self):
        """flow_simulators_mpi_2d.py:323-328."""
        local_max = self._reduce("smi2d_reduce_max", as_ptr(self.vorticity_field))
        return self.real_t(self.ctx.allreduce(float(local_max), "max"))
