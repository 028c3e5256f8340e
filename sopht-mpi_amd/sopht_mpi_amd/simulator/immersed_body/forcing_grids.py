"""Forcing grids for spheres and Cosserat rods, and minimal body stand-ins.

The reference takes these classes from the un-vendored ``sopht`` package
(``SphereForcingGrid``: examples/3d_examples/FlowPastSphereCase/flow_past_sphere_case.py:11, 80-81;
``CosseratRodSurfaceForcingGrid``: examples/3d_examples/FlowPastRodCase/flow_past_rod_case.py:10,
131-132) and the bodies from PyElastica, neither of which is available here.  What is built
below keeps their constructor keywords and the forcing-grid operator interface
(immersed_body_forcing_grid.py:4-26); the node placement itself is this package's own
(parity with sopht's placement is not pinned by anything in the reference tree).

Body convention (PyElastica's): ``position_collection`` (3, n), ``velocity_collection`` (3, n),
``director_collection`` (3, 3, n_elems) with rows = material frame axes in lab coordinates
(so ``Q @ lab = material``), ``omega_collection`` (3, n_elems) in the material frame.
"""
import numpy as np

from .immersed_body_forcing_grid import ImmersedBodyForcingGrid


class RigidSphereBody:
    """Stand-in for ``elastica.Sphere``: one node, one frame."""

    def __init__(self, center, base_radius, density=1.0):
        self.radius = float(base_radius)
        self.density = float(density)
        self.position_collection = np.asarray(center, dtype=np.float64).reshape(3, 1).copy()
        self.velocity_collection = np.zeros((3, 1))
        self.omega_collection = np.zeros((3, 1))
        self.director_collection = np.eye(3).reshape(3, 3, 1).copy()


class StraightRodBody:
    """Stand-in for ``elastica.CosseratRod.straight_rod`` with prescribed (rigid) kinematics."""

    def __init__(self, n_elements, start, direction, normal, base_length, base_radius):
        self.n_elems = int(n_elements)
        t = np.asarray(direction, dtype=np.float64)
        t /= np.linalg.norm(t)
        d1 = np.asarray(normal, dtype=np.float64)
        d1 = d1 - t * (d1 @ t)
        d1 /= np.linalg.norm(d1)
        d2 = np.cross(t, d1)
        s = np.linspace(0.0, float(base_length), self.n_elems + 1)
        self.position_collection = np.asarray(start, dtype=np.float64).reshape(3, 1) + t.reshape(3, 1) * s
        self.velocity_collection = np.zeros((3, self.n_elems + 1))
        self.omega_collection = np.zeros((3, self.n_elems))
        self.director_collection = np.repeat(np.stack([d1, d2, t]).reshape(3, 3, 1), self.n_elems, axis=2)
        self.radius = np.full(self.n_elems, float(base_radius))
        self.lengths = np.full(self.n_elems, float(base_length) / self.n_elems)


def _lab_from_material(directors, k, vec):
    return directors[:, :, k].T @ vec


class SphereForcingGrid(ImmersedBodyForcingGrid):
    """Latitude rings on the sphere surface; ``num_forcing_points_along_equator`` points on the
    equator and the same arc spacing on every other ring and between rings."""

    def __init__(self, grid_dim, rigid_body, num_forcing_points_along_equator):
        if grid_dim != 3:
            raise ValueError("Invalid grid dimensions. Supported only for 3D")
        self.sphere = rigid_body
        n_eq = int(num_forcing_points_along_equator)
        n_polar = max(n_eq // 2, 2)
        local = []
        for i in range(n_polar):
            polar = np.pi * (i + 0.5) / n_polar
            n_ring = max(int(round(n_eq * np.sin(polar))), 1)
            az = 2 * np.pi * (np.arange(n_ring) + 0.5 * (i % 2)) / n_ring
            local.append(np.stack([np.sin(polar) * np.cos(az), np.sin(polar) * np.sin(az),
                                   np.full(n_ring, np.cos(polar))]))
        self.local_frame_relative_position_field = self.sphere.radius * np.concatenate(local, axis=1)
        super().__init__(grid_dim=grid_dim, num_lag_nodes=self.local_frame_relative_position_field.shape[1])
        self.global_frame_relative_position_field = np.zeros_like(self.position_field)
        self._spacing = 2 * np.pi * self.sphere.radius / n_eq
        self.compute_lag_grid_position_field()
        self.compute_lag_grid_velocity_field()

    def compute_lag_grid_position_field(self):
        q = self.sphere.director_collection[:, :, 0]
        self.global_frame_relative_position_field[...] = q.T @ self.local_frame_relative_position_field
        self.position_field[...] = self.sphere.position_collection + self.global_frame_relative_position_field

    def compute_lag_grid_velocity_field(self):
        omega_lab = _lab_from_material(self.sphere.director_collection, 0, self.sphere.omega_collection[:, 0])
        self.velocity_field[...] = self.sphere.velocity_collection + np.cross(
            omega_lab, self.global_frame_relative_position_field, axisa=0, axisb=0, axisc=0)

    def transfer_forcing_from_grid_to_body(self, body_flow_forces, body_flow_torques, lag_grid_forcing_field):
        """The Lagrangian forcing acts on the fluid; the body receives its negative."""
        body_flow_forces[...] = -np.sum(lag_grid_forcing_field, axis=1, keepdims=True)
        torque_lab = -np.sum(np.cross(self.global_frame_relative_position_field, lag_grid_forcing_field,
                                      axisa=0, axisb=0, axisc=0), axis=1)
        body_flow_torques[...] = (self.sphere.director_collection[:, :, 0] @ torque_lab).reshape(3, 1)

    def get_maximum_lagrangian_grid_spacing(self):
        return self._spacing


class CosseratRodElementCentricForcingGrid(ImmersedBodyForcingGrid):
    """One node at the centre of every rod element."""

    def __init__(self, grid_dim, cosserat_rod):
        self.cosserat_rod = cosserat_rod
        super().__init__(grid_dim=grid_dim, num_lag_nodes=cosserat_rod.n_elems)
        self.compute_lag_grid_position_field()
        self.compute_lag_grid_velocity_field()

    def compute_lag_grid_position_field(self):
        x = self.cosserat_rod.position_collection
        self.position_field[...] = 0.5 * (x[: self.grid_dim, 1:] + x[: self.grid_dim, :-1])

    def compute_lag_grid_velocity_field(self):
        v = self.cosserat_rod.velocity_collection
        self.velocity_field[...] = 0.5 * (v[: self.grid_dim, 1:] + v[: self.grid_dim, :-1])

    def transfer_forcing_from_grid_to_body(self, body_flow_forces, body_flow_torques, lag_grid_forcing_field):
        body_flow_forces[...] = 0.0
        body_flow_torques[...] = 0.0
        body_flow_forces[: self.grid_dim, 1:] -= 0.5 * lag_grid_forcing_field
        body_flow_forces[: self.grid_dim, :-1] -= 0.5 * lag_grid_forcing_field

    def get_maximum_lagrangian_grid_spacing(self):
        return float(np.amax(self.cosserat_rod.lengths))


class CosseratRodSurfaceForcingGrid(ImmersedBodyForcingGrid):
    """Rings of surface points around every rod element: ``surface_grid_density_for_largest_element``
    points around the thickest element, proportionally fewer around thinner ones."""

    def __init__(self, grid_dim, cosserat_rod, surface_grid_density_for_largest_element, with_cap=False):
        if grid_dim != 3:
            raise ValueError("Invalid grid dimensions. Cosserat rod surface forcing grid is only defined for 3D")
        self.cosserat_rod = cosserat_rod
        n = cosserat_rod.n_elems
        radius = np.asarray(cosserat_rod.radius, dtype=np.float64)
        self.surface_grid_points = np.maximum(
            np.rint(surface_grid_density_for_largest_element * radius / radius.max()).astype(int), 1)
        self.element_of_node = np.repeat(np.arange(n), self.surface_grid_points)
        angles = np.concatenate([2 * np.pi * np.arange(k) / k for k in self.surface_grid_points])
        # (3, M) in each element's material frame: radius * (cos, sin, 0)
        self.local_frame_surface_points = np.stack([np.cos(angles), np.sin(angles), np.zeros_like(angles)])
        super().__init__(grid_dim=grid_dim, num_lag_nodes=int(self.surface_grid_points.sum()))
        self.moment_arm = np.zeros_like(self.position_field)
        self.with_cap = with_cap
        self._density = surface_grid_density_for_largest_element
        self.compute_lag_grid_position_field()
        self.compute_lag_grid_velocity_field()

    def compute_lag_grid_position_field(self):
        rod, e = self.cosserat_rod, self.element_of_node
        centre = 0.5 * (rod.position_collection[:, 1:] + rod.position_collection[:, :-1])
        q = rod.director_collection[:, :, e]                       # (3, 3, M)
        local = self.local_frame_surface_points * np.asarray(rod.radius)[e]
        self.moment_arm[...] = np.einsum("jin,jn->in", q, local)    # Q^T local
        self.position_field[...] = centre[:, e] + self.moment_arm

    def compute_lag_grid_velocity_field(self):
        rod, e = self.cosserat_rod, self.element_of_node
        v_elem = 0.5 * (rod.velocity_collection[:, 1:] + rod.velocity_collection[:, :-1])
        omega_lab = np.einsum("jin,jn->in", rod.director_collection, rod.omega_collection)
        self.velocity_field[...] = v_elem[:, e] + np.cross(omega_lab[:, e], self.moment_arm, axisa=0, axisb=0, axisc=0)

    def transfer_forcing_from_grid_to_body(self, body_flow_forces, body_flow_torques, lag_grid_forcing_field):
        rod, e, n = self.cosserat_rod, self.element_of_node, self.cosserat_rod.n_elems
        body_flow_forces[...] = 0.0
        body_flow_torques[...] = 0.0
        f_elem = np.zeros((3, n))
        t_elem = np.zeros((3, n))
        np.add.at(f_elem.T, e, lag_grid_forcing_field.T)
        np.add.at(t_elem.T, e, np.cross(self.moment_arm, lag_grid_forcing_field, axisa=0, axisb=0, axisc=0).T)
        body_flow_forces[:, 1:] -= 0.5 * f_elem
        body_flow_forces[:, :-1] -= 0.5 * f_elem
        body_flow_torques[...] = -np.einsum("ijn,jn->in", rod.director_collection, t_elem)

    def get_maximum_lagrangian_grid_spacing(self):
        rod = self.cosserat_rod
        around = 2 * np.pi * np.asarray(rod.radius).max() / self._density
        return float(max(np.amax(rod.lengths), around))
