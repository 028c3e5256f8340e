"""What the 2D and the 3D unbounded flow simulators share: the grid geometry derived from the device context,
table-driven allocation of the fields and construction of the operators, and a step that is a list of stages.

The reference builds the same objects imperatively in ``init_domain`` / ``init_fields`` / ``compile_kernels``
(sopht_mpi/simulator/flow/flow_simulators_mpi_3d.py:124-325, flow_simulators_mpi_2d.py:110-240); what a drop-in
needs from those is only their outcome - attribute names, array shapes, coordinates - which is what the tables in
the concrete classes describe.
"""
import ctypes as C
import logging

import numpy as np

from ..._lib import call
from ...sopht_compat import get_test_tol

logger = logging.getLogger("sopht_mpi_amd")


def cell_centres(num_cells, first_cell, dx, ghost_size, real_t):
    """Coordinates of ``num_cells`` cell centres (i + 1/2) dx starting at global cell ``first_cell``, continued
    linearly over ``ghost_size`` ghost cells on either side, in ``real_t``.

    Bit for bit the values of flow_simulators_mpi_3d.py:128-153, which depend on where that code rounds: the
    half-cell shift and the ghost shift are formed in ``real_t`` (python scalars do not promote a numpy scalar),
    the block's first and last edge as int64 * real_t, i.e. in double, and the sum in double before
    ``linspace`` and the final cast."""
    first = np.int64(first_cell)
    block_start, block_end = first * dx, (first + np.int64(num_cells)) * dx
    half_cell, ghost_shift = dx / 2.0, ghost_size * dx
    return np.linspace(half_cell + block_start - ghost_shift, block_end - half_cell + ghost_shift,
                       num_cells + 2 * ghost_size).astype(real_t)


class FieldSpec:
    """One device field of a simulator: attribute name, number of vector components (None: scalar), the flow
    types that own it and, for a view, the attribute whose storage it shares."""

    def __init__(self, name, ncomp, flow_types, view_of=None):
        self.name, self.ncomp, self.flow_types, self.view_of = name, ncomp, flow_types, view_of


class UnboundedFlowSimulatorBase:
    """Constructor arguments, attributes and methods follow ``UnboundedFlowSimulator3D`` / ``2D`` of the
    reference; concrete classes set ``grid_dim``, ``FLOW_TYPES``, ``FIELDS`` and provide ``_build_operators``
    and ``_stages``."""

    grid_dim = None
    FLOW_TYPES = ()
    VORTICITY_FLOW_TYPES = ("navier_stokes", "navier_stokes_with_forcing")
    FIELDS = ()
    _AXES = "zyx"

    # ------------------------------------------------------------------ construction
    def _configure(self, grid_size, x_range, kinematic_viscosity, time, CFL, flow_type, real_t, rank_distribution,
                   ghost_size, options):
        if flow_type not in self.FLOW_TYPES:
            raise ValueError("Invalid flow type given")
        self.grid_size = grid_size
        axes = self._AXES[-self.grid_dim:]
        for axis, n in zip(axes, grid_size):
            setattr(self, f"grid_size_{axis}", n)
        self.x_range, self.real_t, self.flow_type = x_range, real_t, flow_type
        self.kinematic_viscosity, self.CFL, self.time = kinematic_viscosity, CFL, time
        self.rank_distribution, self.ghost_size = rank_distribution, ghost_size
        # True: one library call per Navier-Stokes step; False: the reference's operator-by-operator sequence
        self.fused_step = options.get("fused_step", True)
        self._world = options.get("world", None)
        self.solves_vorticity = flow_type in self.VORTICITY_FLOW_TYPES
        self.has_forcing = flow_type == "navier_stokes_with_forcing"

    def _assemble(self):
        """Topology -> geometry -> fields -> operators -> stage list."""
        self.mpi_construct, self.mpi_ghost_exchange_communicator = self._make_topology()
        self.ctx = self.mpi_ghost_exchange_communicator.ctx
        self._derive_geometry()
        self._allocate_fields()
        self._build_operators(dict(mpi_construct=self.mpi_construct, real_t=self.real_t,
                                   ghost_exchange_communicator=self.mpi_ghost_exchange_communicator))
        self._step_stages = self._stages()
        self._publish_steps()

    def _derive_geometry(self):
        """Grid spacing, axis ranges and the host ``position_field`` (component 0 = x) of this rank's block."""
        self.dx = self.real_t(self.x_range / self.grid_size_x)
        axes = self._AXES[-self.grid_dim:]
        for axis in axes[:-1]:
            setattr(self, f"{axis}_range", self.x_range * getattr(self, f"grid_size_{axis}") / self.grid_size_x)
        local = [int(n) for n in self.mpi_construct.local_grid_size]
        first = [int(c) * n for c, n in zip(self.mpi_construct.grid.coords, local)]
        self.axis_coordinates = {
            axis: cell_centres(n, f0, self.dx, self.ghost_size, self.real_t) for axis, n, f0 in zip(axes, local, first)
        }
        mesh = np.meshgrid(*(self.axis_coordinates[a] for a in axes), indexing="ij")
        self.position_field = np.array(mesh[::-1])   # ordered x, y(, z) like the reference's flipud
        self.local_grid_size_with_ghost = self.mpi_construct.local_grid_size + 2 * self.ghost_size
        logger.info(f"{self.grid_dim}D flow domain: " + ", ".join(
            f"{a.upper()} axis from 0.0 to {getattr(self, a + '_range')}" for a in reversed(axes)))

    def _allocate_fields(self):
        shape = tuple(int(n) for n in self.local_grid_size_with_ghost)
        for spec in self.FIELDS:
            if self.flow_type not in spec.flow_types:
                continue
            if spec.view_of is not None:
                field = getattr(self, spec.view_of).view()
            else:
                field = self.ctx.zeros(shape if spec.ncomp is None else (spec.ncomp,) + shape)
            setattr(self, spec.name, field)

    # ------------------------------------------------------------------ stepping
    def update_simulator_time(self, dt):
        self.time += dt

    def time_step(self, dt, **kwargs):
        """One flow step (``flow_time_step``: the stages of this flow type in order), then the clock."""
        self.flow_time_step(dt=dt, **kwargs)
        self.update_simulator_time(dt=dt)

    @staticmethod
    def _in_sequence(stages):
        def step(dt, **kwargs):
            for stage in stages:
                stage(dt=dt, **kwargs)
        return step

    def _publish_steps(self):
        """The callables users of the reference reach for besides ``time_step``: ``flow_time_step`` (what
        ``time_step`` runs; the reference picks it in ``finalise_flow_timestep``, flow_simulators_mpi_3d.py:335-345,
        _2d.py:242-249) and the step of each flow family under the reference's method name
        (``navier_stokes_with_forcing_timestep`` ...), each a stage list of this simulator run in order."""
        self.flow_time_step = self._in_sequence(self._step_stages)
        for name, stages in self._named_stage_lists().items():
            setattr(self, name, self._in_sequence(stages))

    def _named_stage_lists(self):
        return {}

    def _scaled(self, value):
        """A step coefficient rounded to the field precision, as the reference passes real_t(...)."""
        return self.real_t(value)

    # ------------------------------------------------------------------ diagnostics
    def _reduce(self, entry, field_ptr, *extra):
        out = C.c_double(0.0)
        call(entry, self.ctx.handle, field_ptr, *extra, C.byref(out))
        return out.value

    def _stable_timestep(self, max_abs_velocity_sum, dt_prefac, precision, viscosity_tol):
        """min(CFL dx / (max |u|_1 + tol), 0.9 dx^2 / (2 d) / (nu [+ tol])) dt_prefac
        (flow_simulators_mpi_3d.py:426-449; the 2D rule has no tolerance on the viscosity, _2d.py:305-321)."""
        tol = get_test_tol(precision)
        advective = self.CFL * self.dx / (self.real_t(max_abs_velocity_sum) + tol)
        diffusive = 0.9 * self.dx**2 / (2 * self.grid_dim) / (self.kinematic_viscosity + (tol if viscosity_tol else 0))
        return self.ctx.allreduce(float(min(advective, diffusive)), "min") * dt_prefac
