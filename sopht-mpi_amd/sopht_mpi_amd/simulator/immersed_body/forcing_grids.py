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

from ...numeric.immersed_boundary_ops import DeviceResidentNodes
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


class FramedForcingGrid(ImmersedBodyForcingGrid):
    """Forcing grid whose nodes are tied to frames of a body (the sphere; the elements of a rod):
    node = centre + Q^T (scale * local offset), velocity = v + (Q^T omega) x arm.

    A subclass supplies ``frames()`` and ``loads_to_body``.  Attached to a device context (the interactor does that
    on a single rank), the node arrays are formed by ``smi_ib_frame_nodes`` from the few doubles per frame that
    ``frames()`` returns and land where the interaction kernels read them; ``position_field`` / ``velocity_field``
    then fetch their host copies on demand.  The host methods ``compute_lag_grid_*`` keep working either way.
    """

    FRAME_DOUBLES = 19   # centre 3, directors 9 (row major), velocity 3, omega (material frame) 3, scale 1
    _dev = None            # device attachment (attach_device)
    _host_stale = False    # the host copies of the node arrays are older than the device arrays

    def _init_frames(self, frame_of_node, local_offsets, num_frames):
        frame_of_node = np.ascontiguousarray(frame_of_node, dtype=np.int32)
        if np.any(np.diff(frame_of_node) < 0):
            raise ValueError("the nodes of a frame must be contiguous and the frames in order")
        self.frame_of_node = frame_of_node
        self.local_offsets = np.ascontiguousarray(local_offsets, dtype=np.float64)
        self.num_frames = int(num_frames)
        self.frame_start = np.searchsorted(frame_of_node, np.arange(self.num_frames + 1)).astype(np.int32)
        # farthest node of each frame, in units of the frame's scale: bounds the body without touching the nodes
        norms = np.linalg.norm(self.local_offsets, axis=0)
        self.frame_reach = np.array([norms[a:b].max() if b > a else 0.0
                                     for a, b in zip(self.frame_start[:-1], self.frame_start[1:])])

    # ---- the arrays the reference's users read ---------------------------------
    def _host_field(self, name):
        """Host copy of a node array; while the device forms the nodes (update_device_nodes) the positions, the
        velocities AND the moment arms are brought up to date first, so that the host-side
        transfer_forcing_from_grid_to_body sees the arms of the body's current pose."""
        if self._host_stale and self._dev is not None:
            self._position_field[...] = self._dev["pos"].get()
            self._velocity_field[...] = self._dev["vel"].get()
            if getattr(self, "_arm_field", None) is not None:
                self._arm_field[...] = self._dev["arms"].get()
            self._host_stale = False
        return getattr(self, name)

    position_field = property(lambda self: self._host_field("_position_field"),
                              lambda self, v: setattr(self, "_position_field", v))
    velocity_field = property(lambda self: self._host_field("_velocity_field"),
                              lambda self, v: setattr(self, "_velocity_field", v))
    # lab-frame offset of every node from the centre of its frame; the subclasses expose it under the name their
    # sopht counterpart uses (global_frame_relative_position_field, moment_arm)
    _arm_field = None

    # ---- what a subclass states -------------------------------------------------
    def frames(self):
        """(num_frames, 19) float64: centre, directors, velocity, omega (material frame), scale of every frame."""
        raise NotImplementedError

    def loads_to_body(self, frame_sums, body_flow_forces, body_flow_torques):
        """Body loads from the per-frame sums (num_frames, 6) of (force, arm x force) over the nodes, lab frame."""
        raise NotImplementedError

    # ---- device path ------------------------------------------------------------
    def attach_device(self, ctx, positions, velocities):
        """``positions`` / ``velocities``: the (3, M) device arrays the interaction kernels read."""
        self._dev = dict(ctx=ctx, pos=positions, vel=velocities,
                         local=ctx.from_numpy(self.local_offsets), frame_of_node=ctx.from_numpy(self.frame_of_node),
                         frame_start=ctx.from_numpy(self.frame_start),
                         frames=ctx.zeros((self.num_frames, self.FRAME_DOUBLES), np.float64),
                         arms=ctx.zeros((3, self.num_lag_nodes), np.float64),
                         loads=ctx.zeros((self.num_frames, 6), np.float64))

    def detach_device(self):
        self._host_field("_position_field")
        self._dev = None

    def update_device_nodes(self):
        """Form positions, velocities and moment arms of every node on the device; returns the box that holds them."""
        from ..._lib import call
        from ...device import as_ptr

        d = self._dev
        frames = np.ascontiguousarray(self.frames(), dtype=np.float64)
        d["frames"].set(frames)
        call("smi_ib_frame_nodes", d["ctx"].handle, as_ptr(d["pos"]), as_ptr(d["vel"]), as_ptr(d["arms"]),
             as_ptr(d["local"]), as_ptr(d["frame_of_node"]), as_ptr(d["frames"]), self.num_frames, self.num_lag_nodes)
        self._host_stale = True
        reach = (self.frame_reach * np.abs(frames[:, 18])).reshape(-1, 1)
        return DeviceResidentNodes((frames[:, :3] - reach).min(axis=0), (frames[:, :3] + reach).max(axis=0),
                                   self.num_lag_nodes)

    def transfer_forcing_from_device(self, body_flow_forces, body_flow_torques, lag_grid_forcing_device):
        """transfer_forcing_from_grid_to_body with the node forces still on the device: 6 doubles per frame come back."""
        from ..._lib import call
        from ...device import as_ptr

        d = self._dev
        call("smi_ib_frame_loads", d["ctx"].handle, as_ptr(d["loads"]), as_ptr(lag_grid_forcing_device),
             as_ptr(d["arms"]), as_ptr(d["frame_start"]), self.num_frames, self.num_lag_nodes)
        self.loads_to_body(d["loads"].get(), body_flow_forces, body_flow_torques)


class SphereForcingGrid(FramedForcingGrid):
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
        self._init_frames(np.zeros(self.num_lag_nodes, dtype=np.int32), self.local_frame_relative_position_field, 1)
        self._arm_field = np.zeros_like(self.position_field)
        self._spacing = 2 * np.pi * self.sphere.radius / n_eq
        self.compute_lag_grid_position_field()
        self.compute_lag_grid_velocity_field()

    global_frame_relative_position_field = property(lambda self: self._host_field("_arm_field"))

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
        sums = np.concatenate([np.sum(lag_grid_forcing_field, axis=1),
                               np.sum(np.cross(self.global_frame_relative_position_field, lag_grid_forcing_field,
                                               axisa=0, axisb=0, axisc=0), axis=1)])
        self.loads_to_body(sums.reshape(1, 6), body_flow_forces, body_flow_torques)

    def frames(self):
        s = self.sphere
        return np.concatenate([s.position_collection[:, 0], s.director_collection[:, :, 0].reshape(9),
                               s.velocity_collection[:, 0], s.omega_collection[:, 0], [1.0]]).reshape(1, 19)

    def loads_to_body(self, frame_sums, body_flow_forces, body_flow_torques):
        body_flow_forces[...] = -frame_sums[0, :3].reshape(3, 1)
        body_flow_torques[...] = -(self.sphere.director_collection[:, :, 0] @ frame_sums[0, 3:]).reshape(3, 1)

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


class CosseratRodSurfaceForcingGrid(FramedForcingGrid):
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
        self._init_frames(self.element_of_node, self.local_frame_surface_points, n)
        self._arm_field = np.zeros_like(self.position_field)
        self.with_cap = with_cap
        self._density = surface_grid_density_for_largest_element
        self.compute_lag_grid_position_field()
        self.compute_lag_grid_velocity_field()

    moment_arm = property(lambda self: self._host_field("_arm_field"))

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
        e, n = self.element_of_node, self.cosserat_rod.n_elems
        sums = np.zeros((n, 6))
        np.add.at(sums[:, :3], e, lag_grid_forcing_field.T)
        np.add.at(sums[:, 3:], e, np.cross(self.moment_arm, lag_grid_forcing_field, axisa=0, axisb=0, axisc=0).T)
        self.loads_to_body(sums, body_flow_forces, body_flow_torques)

    def frames(self):
        rod, n = self.cosserat_rod, self.cosserat_rod.n_elems
        f = np.empty((n, 19))
        f[:, 0:3] = (0.5 * (rod.position_collection[:, 1:] + rod.position_collection[:, :-1])).T
        f[:, 3:12] = np.moveaxis(rod.director_collection, 2, 0).reshape(n, 9)
        f[:, 12:15] = (0.5 * (rod.velocity_collection[:, 1:] + rod.velocity_collection[:, :-1])).T
        f[:, 15:18] = np.asarray(rod.omega_collection).T
        f[:, 18] = np.asarray(rod.radius)
        return f

    def loads_to_body(self, frame_sums, body_flow_forces, body_flow_torques):
        """Half of an element's force to each of its two nodes; its torque into the element's material frame."""
        rod = self.cosserat_rod
        body_flow_forces[...] = 0.0
        body_flow_forces[:, 1:] -= 0.5 * frame_sums[:, :3].T
        body_flow_forces[:, :-1] -= 0.5 * frame_sums[:, :3].T
        body_flow_torques[...] = -np.einsum("ijn,jn->in", rod.director_collection, frame_sums[:, 3:].T)

    def get_maximum_lagrangian_grid_spacing(self):
        rod = self.cosserat_rod
        around = 2 * np.pi * np.asarray(rod.radius).max() / self._density
        return float(max(np.amax(rod.lengths), around))
