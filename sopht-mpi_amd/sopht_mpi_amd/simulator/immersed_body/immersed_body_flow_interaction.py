"""Immersed body <-> flow interactors: the objects a simulation script builds in place of
sopht_mpi/simulator/immersed_body/immersed_body_flow_interaction_mpi.py:10-202 and the rigid-body / Cosserat-rod
constructors next to it (rigid_body_flow_interaction_mpi.py:10-64, cosserat_rod_flow_interaction_mpi.py:10-68).

An interactor is the virtual-boundary forcing of one body: it owns the body's forcing grid (on the master rank; the
other ranks hold an empty stand-in), views of the simulator's device-resident forcing and velocity fields, and the
host arrays the body model reads its loads from.  Calling it runs ghosting -> node kinematics -> interpolation ->
penalty force -> spreading; ``compute_flow_forces_and_torques`` stops after the penalty force and hands the loads to
the body, which is what a rod's time integrator asks for on every sub-step.
"""
import logging
import os

import numpy as np

from ...numeric.immersed_boundary_ops import VirtualBoundaryForcingMPI
from .forcing_grids import FramedForcingGrid
from .immersed_body_forcing_grid import EmptyForcingGrid

logger = logging.getLogger("sopht_mpi_amd")


def lagrangian_resolution_note(grid_name, dx, max_lag_grid_dx):
    """(log level, message) about the Lagrangian spacing relative to the Eulerian one: coarser than 2 dx and finer
    than dx / 2 are both worth a warning (immersed_body_flow_interaction_mpi.py:52-80)."""
    head = f"For {grid_name}: Eulerian grid spacing (dx): {dx}, max Lagrangian grid spacing: {max_lag_grid_dx}"
    if max_lag_grid_dx > 2 * dx:
        return logging.WARNING, (head + " > 2 * dx. The Lagrangian grid of the body is too coarse relative to the "
                                 "Eulerian grid of the flow. Please make the Lagrangian grid finer.")
    if max_lag_grid_dx < 0.5 * dx:
        return logging.WARNING, (head + " < 0.5 * dx. The Lagrangian grid of the body is too fine relative to the "
                                 "Eulerian grid of the flow. Please make the Lagrangian grid coarser.")
    return logging.INFO, f"For {grid_name}: Lagrangian grid is resolved almost the same as the Eulerian grid."


class ImmersedBodyFlowInteractionMPI(VirtualBoundaryForcingMPI):
    """Base interactor.  A derived class sets ``forcing_grid``, ``master_rank``, ``body_flow_forces`` and
    ``body_flow_torques`` before it calls this initialiser (as in the reference)."""

    def __init__(self, mpi_construct, mpi_ghost_exchange_communicator, eul_grid_forcing_field,
                 eul_grid_velocity_field, virtual_boundary_stiffness_coeff, virtual_boundary_damping_coeff, dx,
                 grid_dim, eul_grid_coord_shift=None, interp_kernel_width=None, enable_eul_grid_forcing_reset=False,
                 start_time=0.0, assume_data_locality=False, auto_ghosting=True, interp_kernel_type="cosine"):
        self.mpi_ghost_exchange_communicator = mpi_ghost_exchange_communicator
        self.eul_grid_forcing_field = eul_grid_forcing_field.view()     # the simulator's device arrays, not copies
        self.eul_grid_velocity_field = eul_grid_velocity_field.view()
        # the master rank knows the body; everybody needs its spacing and its nodes to size their buffers
        spacing = mpi_construct.grid.bcast(self.forcing_grid.get_maximum_lagrangian_grid_spacing(),
                                           root=self.master_rank)
        logger.log(*lagrangian_resolution_note(type(self.forcing_grid).__name__, dx, spacing))
        nodes = self.forcing_grid.position_field
        if mpi_construct.size > 1:
            nodes = mpi_construct.grid.bcast(nodes, root=self.master_rank)
        # coefficients are given per unit area (length in 2D) of body surface: one node stands for spacing^(d-1)
        # of it (immersed_body_flow_interaction_mpi.py:82-85)
        node_measure = spacing ** (grid_dim - 1)
        super().__init__(
            mpi_construct=mpi_construct, ghost_size=mpi_ghost_exchange_communicator.ghost_size,
            virtual_boundary_stiffness_coeff=virtual_boundary_stiffness_coeff * node_measure,
            virtual_boundary_damping_coeff=virtual_boundary_damping_coeff * node_measure,
            grid_dim=grid_dim, dx=dx, eul_grid_coord_shift=eul_grid_coord_shift,
            interp_kernel_width=interp_kernel_width, enable_eul_grid_forcing_reset=enable_eul_grid_forcing_reset,
            start_time=start_time, master_rank=self.master_rank, global_lag_grid_position_field=nodes,
            assume_data_locality=assume_data_locality, interp_kernel_type=interp_kernel_type,
        )
        # On one rank a framed forcing grid (sphere, rod surface) forms its node arrays on the device, straight into
        # the arrays the interaction kernels read: a few doubles per frame cross PCIe instead of 6 M per call, and
        # the host does no per-node arithmetic.  SOPHT_MI_HOST_FORCING_GRID=1 keeps the host path.
        self.nodes_on_device = (mpi_construct.size == 1 and grid_dim == 3
                                and isinstance(self.forcing_grid, FramedForcingGrid)
                                and os.environ.get("SOPHT_MI_HOST_FORCING_GRID", "0") in ("", "0"))
        if self.nodes_on_device:
            self.forcing_grid.attach_device(self.ctx, self.local_lag_grid_position_field,
                                            self.local_lag_grid_velocity_field)
        self.auto_ghosting = bool(auto_ghosting)
        if not self.auto_ghosting:
            logger.warning("Auto ghosting of velocity field is disabled for interactor. "
                           "Please ensure ghosting is done before calling interactor functions.")

    # ------------------------------------------------------------------ one interaction
    def _body_state(self):
        """Bring the inputs of an interaction up to date - ghost cells of the velocity (unless the caller does
        that), node positions and velocities from the body - and return them as the keyword arguments the
        forcing methods take."""
        if self.auto_ghosting:
            self.mpi_ghost_exchange_communicator.exchange_vector_field_init(self.eul_grid_velocity_field)
            self.mpi_ghost_exchange_communicator.exchange_finalise()
        if self.nodes_on_device:
            nodes = self.forcing_grid.update_device_nodes()
            return dict(local_eul_grid_velocity_field=self.eul_grid_velocity_field,
                        global_lag_grid_position_field=nodes, global_lag_grid_velocity_field=nodes)
        self.forcing_grid.compute_lag_grid_position_field()
        self.forcing_grid.compute_lag_grid_velocity_field()
        return dict(local_eul_grid_velocity_field=self.eul_grid_velocity_field,
                    global_lag_grid_position_field=self.forcing_grid.position_field,
                    global_lag_grid_velocity_field=self.forcing_grid.velocity_field)

    def compute_full_interaction(self):
        """Forces on the Lagrangian nodes and their spread onto the Eulerian forcing field (:123-166)."""
        self.compute_interaction_forcing(local_eul_grid_forcing_field=self.eul_grid_forcing_field, **self._body_state())

    __call__ = compute_full_interaction

    def compute_interaction_on_lag_grid(self):
        """Forces on the Lagrangian nodes only (no spreading)."""
        self.compute_interaction_force_on_lag_grid(**self._body_state())

    def compute_flow_forces_and_torques(self):
        """Node forces -> loads on the body, in ``body_flow_forces`` / ``body_flow_torques`` (:168-175)."""
        self.compute_interaction_on_lag_grid()
        if self.nodes_on_device:
            # per-frame sums on the device: 6 doubles per frame come back instead of 3 per node
            self.forcing_grid.transfer_forcing_from_device(self.body_flow_forces, self.body_flow_torques,
                                                           self.local_lag_grid_forcing_field)
            return
        self.forcing_grid.transfer_forcing_from_grid_to_body(
            body_flow_forces=self.body_flow_forces, body_flow_torques=self.body_flow_torques,
            lag_grid_forcing_field=self.global_lag_grid_forcing_field)

    def get_grid_deviation_error_l2_norm(self, compute_global=True):
        """Root mean square of the node position mismatch, over this rank's nodes or over all (:177-202)."""
        sum_sq = float(np.sum(self.local_lag_grid_position_mismatch_field.get() ** 2))
        count = self.forcing_grid.num_lag_nodes
        if compute_global:
            sum_sq = self.mpi_construct.world.allreduce(sum_sq, "sum")
            count = self.mpi_construct.grid.bcast(count, root=self.master_rank)
        return np.sqrt(sum_sq) / np.sqrt(count)


def _attach_body(interactor, mpi_construct, master_rank, grid_dim, forcing_grid_cls, body_keyword, body,
                 force_shape, torque_shape, forcing_grid_kwargs):
    """What the body-specific interactors do before the common initialiser: host arrays for the loads and the
    forcing grid, built where the body lives (``master_rank``), an empty stand-in elsewhere."""
    interactor.body_flow_forces = np.zeros(force_shape)
    interactor.body_flow_torques = np.zeros(torque_shape)
    interactor.master_rank = master_rank
    if mpi_construct.rank == master_rank:
        interactor.forcing_grid = forcing_grid_cls(grid_dim=grid_dim, **{body_keyword: body}, **forcing_grid_kwargs)
    else:
        interactor.forcing_grid = EmptyForcingGrid(grid_dim=grid_dim)


class RigidBodyFlowInteractionMPI(ImmersedBodyFlowInteractionMPI):
    """Interactor of a rigid body: one force and one torque vector, shape (3, 1)
    (rigid_body_flow_interaction_mpi.py:10-64)."""

    def __init__(self, mpi_construct, mpi_ghost_exchange_communicator, rigid_body, eul_grid_forcing_field,
                 eul_grid_velocity_field, virtual_boundary_stiffness_coeff, virtual_boundary_damping_coeff, dx,
                 grid_dim, forcing_grid_cls, eul_grid_coord_shift=None, interp_kernel_width=None,
                 enable_eul_grid_forcing_reset=False, start_time=0.0, master_rank=0, assume_data_locality=False,
                 auto_ghosting=True, **forcing_grid_kwargs):
        _attach_body(self, mpi_construct, master_rank, grid_dim, forcing_grid_cls, "rigid_body", rigid_body,
                     (3, 1), (3, 1), forcing_grid_kwargs)
        super().__init__(mpi_construct, mpi_ghost_exchange_communicator, eul_grid_forcing_field,
                         eul_grid_velocity_field, virtual_boundary_stiffness_coeff, virtual_boundary_damping_coeff,
                         dx, grid_dim, eul_grid_coord_shift=eul_grid_coord_shift,
                         interp_kernel_width=interp_kernel_width,
                         enable_eul_grid_forcing_reset=enable_eul_grid_forcing_reset, start_time=start_time,
                         assume_data_locality=assume_data_locality, auto_ghosting=auto_ghosting)


class CosseratRodFlowInteraction(ImmersedBodyFlowInteractionMPI):
    """Interactor of a Cosserat rod: forces per node, (3, n_elems + 1), torques per element, (3, n_elems)
    (cosserat_rod_flow_interaction_mpi.py:10-68)."""

    def __init__(self, mpi_construct, mpi_ghost_exchange_communicator, cosserat_rod, eul_grid_forcing_field,
                 eul_grid_velocity_field, virtual_boundary_stiffness_coeff, virtual_boundary_damping_coeff, dx,
                 grid_dim, forcing_grid_cls, eul_grid_coord_shift=None, interp_kernel_width=None,
                 enable_eul_grid_forcing_reset=False, start_time=0.0, master_rank=0, assume_data_locality=False,
                 auto_ghosting=True, **forcing_grid_kwargs):
        n_elems = getattr(cosserat_rod, "n_elems", 1)
        _attach_body(self, mpi_construct, master_rank, grid_dim, forcing_grid_cls, "cosserat_rod", cosserat_rod,
                     (3, n_elems + 1), (3, n_elems), forcing_grid_kwargs)
        super().__init__(mpi_construct, mpi_ghost_exchange_communicator, eul_grid_forcing_field,
                         eul_grid_velocity_field, virtual_boundary_stiffness_coeff, virtual_boundary_damping_coeff,
                         dx, grid_dim, eul_grid_coord_shift=eul_grid_coord_shift,
                         interp_kernel_width=interp_kernel_width,
                         enable_eul_grid_forcing_reset=enable_eul_grid_forcing_reset, start_time=start_time,
                         assume_data_locality=assume_data_locality, auto_ghosting=auto_ghosting)
