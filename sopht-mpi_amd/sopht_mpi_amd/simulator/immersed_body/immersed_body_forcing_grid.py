"""Forcing-grid operator API (the object an immersed-body interactor drives).

The reference's forcing grids live in the un-vendored ``sopht`` package; the interface
below is the one sopht_mpi uses (simulator/immersed_body/immersed_body_forcing_grid.py:4-26,
immersed_body_flow_interaction_mpi.py:47, 100, 138-144, 171-175, 185).
"""
import numpy as np


class ImmersedBodyForcingGrid:
    """Base class: Lagrangian node positions / velocities as host (grid_dim, M) float64 arrays."""

    def __init__(self, grid_dim, num_lag_nodes):
        self.grid_dim = grid_dim
        self.num_lag_nodes = num_lag_nodes
        self.position_field = np.zeros((grid_dim, num_lag_nodes))
        self.velocity_field = np.zeros_like(self.position_field)

    def compute_lag_grid_position_field(self):
        raise NotImplementedError

    def compute_lag_grid_velocity_field(self):
        raise NotImplementedError

    def transfer_forcing_from_grid_to_body(self, body_flow_forces, body_flow_torques, lag_grid_forcing_field):
        raise NotImplementedError

    def get_maximum_lagrangian_grid_spacing(self):
        raise NotImplementedError


class EmptyForcingGrid(ImmersedBodyForcingGrid):
    """immersed_body_forcing_grid.py:4-26: what non-master ranks hold."""

    def __init__(self, grid_dim):
        super().__init__(grid_dim=grid_dim, num_lag_nodes=0)

    def compute_lag_grid_position_field(self):
        pass

    def compute_lag_grid_velocity_field(self):
        pass

    def transfer_forcing_from_grid_to_body(self, body_flow_forces, body_flow_torques, lag_grid_forcing_field):
        pass

    def get_maximum_lagrangian_grid_spacing(self):
        pass


class PrescribedForcingGrid(ImmersedBodyForcingGrid):
    """A rigid set of Lagrangian nodes with prescribed kinematics.

    ``rigid_body`` may be any object; nodes come from ``positions`` (3, M), move with
    ``velocity`` (3,) added each ``compute_lag_grid_*`` call through ``body_position(time)`` /
    ``body_velocity(time)`` callables if given.  The net force and torque about ``centre``
    are accumulated by ``transfer_forcing_from_grid_to_body`` (force on the body = minus the
    sum of the forces applied to the fluid nodes).
    """

    def __init__(self, grid_dim, rigid_body=None, positions=None, max_spacing=None, centre=None, **_):
        positions = np.asarray(positions, dtype=np.float64)
        super().__init__(grid_dim=grid_dim, num_lag_nodes=positions.shape[-1])
        self.rigid_body = rigid_body
        self.reference_positions = positions.copy()
        self.position_field[...] = positions
        self.centre = np.mean(positions, axis=1) if centre is None else np.asarray(centre, dtype=np.float64)
        self._max_spacing = max_spacing

    def compute_lag_grid_position_field(self):
        self.position_field[...] = self.reference_positions

    def compute_lag_grid_velocity_field(self):
        self.velocity_field[...] = 0.0

    def transfer_forcing_from_grid_to_body(self, body_flow_forces, body_flow_torques, lag_grid_forcing_field):
        body_flow_forces[...] = -np.sum(lag_grid_forcing_field, axis=1).reshape(-1, 1)
        arm = self.position_field - self.centre.reshape(-1, 1)
        body_flow_torques[...] = -np.sum(np.cross(arm.T, lag_grid_forcing_field.T), axis=0).reshape(-1, 1)

    def get_maximum_lagrangian_grid_spacing(self):
        if self._max_spacing is not None:
            return self._max_spacing
        raise ValueError("PrescribedForcingGrid needs max_spacing")
