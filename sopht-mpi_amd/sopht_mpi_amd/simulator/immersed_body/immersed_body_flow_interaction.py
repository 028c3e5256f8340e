"""Immersed body <-> flow interactors, same surface as
sopht_mpi/simulator/immersed_body/immersed_body_flow_interaction_mpi.py:10-202 and the
rigid-body / Cosserat-rod constructors next to it."""
import logging

import numpy as np

from ...numeric.immersed_boundary_ops import VirtualBoundaryForcingMPI
from .immersed_body_forcing_grid import EmptyForcingGrid

logger = logging.getLogger("sopht_mpi_amd")


class ImmersedBodyFlowInteractionMPI(VirtualBoundaryForcingMPI):
    """Base class for immersed body flow interaction.  Derived classes set ``forcing_grid``,
    ``master_rank``, ``body_flow_forces`` and ``body_flow_torques`` before calling this
    initialiser, as in the reference."""

    def __init__(self, mpi_construct, mpi_ghost_exchange_communicator, eul_grid_forcing_field,
                 eul_grid_velocity_field, virtual_boundary_stiffness_coeff, virtual_boundary_damping_coeff, dx,
                 grid_dim, eul_grid_coord_shift=None, interp_kernel_width=None, enable_eul_grid_forcing_reset=False,
                 start_time=0.0, assume_data_locality=False, auto_ghosting=True, interp_kernel_type="cosine"):
        self.mpi_ghost_exchange_communicator = mpi_ghost_exchange_communicator
        # views of the simulator's device fields
        self.eul_grid_forcing_field = eul_grid_forcing_field.view()
        self.eul_grid_velocity_field = eul_grid_velocity_field.view()
        max_lag_grid_dx = self.forcing_grid.get_maximum_lagrangian_grid_spacing()
        max_lag_grid_dx = mpi_construct.grid.bcast(max_lag_grid_dx, root=self.master_rank)
        grid_type = type(self.forcing_grid).__name__
        if max_lag_grid_dx > 2 * dx:
            logger.warning(
                f"For {grid_type}: Eulerian grid spacing (dx): {dx}, max Lagrangian grid spacing: "
                f"{max_lag_grid_dx} > 2 * dx. The Lagrangian grid of the body is too coarse relative to the "
                "Eulerian grid of the flow. Please make the Lagrangian grid finer."
            )
        elif max_lag_grid_dx < 0.5 * dx:
            logger.warning(
                f"For {grid_type}: Eulerian grid spacing (dx): {dx}, max Lagrangian grid spacing: "
                f"{max_lag_grid_dx} < 0.5 * dx. The Lagrangian grid of the body is too fine relative to the "
                "Eulerian grid of the flow. Please make the Lagrangian grid coarser."
            )
        else:
            logger.info(f"For {grid_type}: Lagrangian grid is resolved almost the same as the Eulerian grid.")
        # :82-85
        virtual_boundary_stiffness_coeff *= max_lag_grid_dx ** (grid_dim - 1)
        virtual_boundary_damping_coeff *= max_lag_grid_dx ** (grid_dim - 1)
        global_positions = self.forcing_grid.position_field
        if mpi_construct.size > 1:
            global_positions = mpi_construct.grid.bcast(global_positions, root=self.master_rank)
        super().__init__(
            mpi_construct=mpi_construct, ghost_size=self.mpi_ghost_exchange_communicator.ghost_size,
            virtual_boundary_stiffness_coeff=virtual_boundary_stiffness_coeff,
            virtual_boundary_damping_coeff=virtual_boundary_damping_coeff, grid_dim=grid_dim, dx=dx,
            eul_grid_coord_shift=eul_grid_coord_shift, interp_kernel_width=interp_kernel_width,
            enable_eul_grid_forcing_reset=enable_eul_grid_forcing_reset, start_time=start_time,
            master_rank=self.master_rank, global_lag_grid_position_field=global_positions,
            assume_data_locality=assume_data_locality, interp_kernel_type=interp_kernel_type,
        )
        if auto_ghosting:
            self.compute_full_interaction = self._compute_full_interaction_with_ghosting
            self.compute_interaction_on_lag_grid = self._compute_interaction_on_lag_grid_with_ghosting
        else:
            logger.warning("Auto ghosting of velocity field is disabled for interactor. "
                           "Please ensure ghosting is done before calling interactor functions.")
            self.compute_full_interaction = self._compute_full_interaction_without_ghosting
            self.compute_interaction_on_lag_grid = self._compute_interaction_on_lag_grid_without_ghosting

    def __call__(self):
        self.compute_full_interaction()

    def _ghost_velocity_field_for_interaction(self):
        self.mpi_ghost_exchange_communicator.exchange_vector_field_init(self.eul_grid_velocity_field)
        self.mpi_ghost_exchange_communicator.exchange_finalise()

    def _compute_interaction_on_lag_grid_without_ghosting(self):
        self.forcing_grid.compute_lag_grid_position_field()
        self.forcing_grid.compute_lag_grid_velocity_field()
        self.compute_interaction_force_on_lag_grid(
            local_eul_grid_velocity_field=self.eul_grid_velocity_field,
            global_lag_grid_position_field=self.forcing_grid.position_field,
            global_lag_grid_velocity_field=self.forcing_grid.velocity_field,
        )

    def _compute_interaction_on_lag_grid_with_ghosting(self):
        self._ghost_velocity_field_for_interaction()
        self._compute_interaction_on_lag_grid_without_ghosting()

    def _compute_full_interaction_without_ghosting(self):
        self.forcing_grid.compute_lag_grid_position_field()
        self.forcing_grid.compute_lag_grid_velocity_field()
        self.compute_interaction_forcing(
            local_eul_grid_forcing_field=self.eul_grid_forcing_field,
            local_eul_grid_velocity_field=self.eul_grid_velocity_field,
            global_lag_grid_position_field=self.forcing_grid.position_field,
            global_lag_grid_velocity_field=self.forcing_grid.velocity_field,
        )

    def _compute_full_interaction_with_ghosting(self):
        self._ghost_velocity_field_for_interaction()
        self._compute_full_interaction_without_ghosting()

    def compute_flow_forces_and_torques(self):
        """:168-175."""
        self.compute_interaction_on_lag_grid()
        self.forcing_grid.transfer_forcing_from_grid_to_body(
            body_flow_forces=self.body_flow_forces, body_flow_torques=self.body_flow_torques,
            lag_grid_forcing_field=self.global_lag_grid_forcing_field,
        )

    def get_grid_deviation_error_l2_norm(self, compute_global=True):
        """:177-202."""
        local = self.local_lag_grid_position_mismatch_field.get()
        if not compute_global:
            return np.linalg.norm(local) / np.sqrt(self.forcing_grid.num_lag_nodes)
        total = self.mpi_construct.world.allreduce(np.linalg.norm(local) ** 2, "sum")
        n = self.mpi_construct.grid.bcast(self.forcing_grid.num_lag_nodes, root=self.master_rank)
        return np.sqrt(total) / np.sqrt(n)


class RigidBodyFlowInteractionMPI(ImmersedBodyFlowInteractionMPI):
    """rigid_body/rigid_body_flow_interaction_mpi.py:10-64."""

    def __init__(self, mpi_construct, mpi_ghost_exchange_communicator, rigid_body, eul_grid_forcing_field,
                 eul_grid_velocity_field, virtual_boundary_stiffness_coeff, virtual_boundary_damping_coeff, dx,
                 grid_dim, forcing_grid_cls, eul_grid_coord_shift=None, interp_kernel_width=None,
                 enable_eul_grid_forcing_reset=False, start_time=0.0, master_rank=0, assume_data_locality=False,
                 auto_ghosting=True, **forcing_grid_kwargs):
        self.body_flow_forces = np.zeros((3, 1))
        self.body_flow_torques = np.zeros((3, 1))
        self.master_rank = master_rank
        if mpi_construct.rank == self.master_rank:
            self.forcing_grid = forcing_grid_cls(grid_dim=grid_dim, rigid_body=rigid_body, **forcing_grid_kwargs)
        else:
            self.forcing_grid = EmptyForcingGrid(grid_dim=grid_dim)
        super().__init__(
            mpi_construct=mpi_construct, mpi_ghost_exchange_communicator=mpi_ghost_exchange_communicator,
            eul_grid_forcing_field=eul_grid_forcing_field, eul_grid_velocity_field=eul_grid_velocity_field,
            virtual_boundary_stiffness_coeff=virtual_boundary_stiffness_coeff,
            virtual_boundary_damping_coeff=virtual_boundary_damping_coeff, dx=dx, grid_dim=grid_dim,
            eul_grid_coord_shift=eul_grid_coord_shift, interp_kernel_width=interp_kernel_width,
            enable_eul_grid_forcing_reset=enable_eul_grid_forcing_reset, start_time=start_time,
            assume_data_locality=assume_data_locality, auto_ghosting=auto_ghosting,
        )


class CosseratRodFlowInteraction(ImmersedBodyFlowInteractionMPI):
    """cosserat_rod/cosserat_rod_flow_interaction_mpi.py:10-68: forces and torques per rod
    element, shape (3, n_elems + 1) / (3, n_elems)."""

    def __init__(self, mpi_construct, mpi_ghost_exchange_communicator, cosserat_rod, eul_grid_forcing_field,
                 eul_grid_velocity_field, virtual_boundary_stiffness_coeff, virtual_boundary_damping_coeff, dx,
                 grid_dim, forcing_grid_cls, eul_grid_coord_shift=None, interp_kernel_width=None,
                 enable_eul_grid_forcing_reset=False, start_time=0.0, master_rank=0, assume_data_locality=False,
                 auto_ghosting=True, **forcing_grid_kwargs):
        n_elems = getattr(cosserat_rod, "n_elems", 1)
        self.body_flow_forces = np.zeros((3, n_elems + 1))
        self.body_flow_torques = np.zeros((3, n_elems))
        self.master_rank = master_rank
        if mpi_construct.rank == self.master_rank:
            self.forcing_grid = forcing_grid_cls(grid_dim=grid_dim, cosserat_rod=cosserat_rod, **forcing_grid_kwargs)
        else:
            self.forcing_grid = EmptyForcingGrid(grid_dim=grid_dim)
        super().__init__(
            mpi_construct=mpi_construct, mpi_ghost_exchange_communicator=mpi_ghost_exchange_communicator,
            eul_grid_forcing_field=eul_grid_forcing_field, eul_grid_velocity_field=eul_grid_velocity_field,
            virtual_boundary_stiffness_coeff=virtual_boundary_stiffness_coeff,
            virtual_boundary_damping_coeff=virtual_boundary_damping_coeff, dx=dx, grid_dim=grid_dim,
            eul_grid_coord_shift=eul_grid_coord_shift, interp_kernel_width=interp_kernel_width,
            enable_eul_grid_forcing_reset=enable_eul_grid_forcing_reset, start_time=start_time,
            assume_data_locality=assume_data_locality, auto_ghosting=auto_ghosting,
        )
