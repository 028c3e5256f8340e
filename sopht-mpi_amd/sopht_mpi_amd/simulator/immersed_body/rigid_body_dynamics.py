"""Two-way coupling for the stand-in bodies: an explicit rigid-body integrator driven by the flow loads.

The reference's examples integrate their bodies with PyElastica (absent here): a position-Verlet stepper whose
forcing calls ``interactor.compute_flow_forces_and_torques()`` on every sub-step, with the interactor's mismatch
state advanced by the same sub-step (examples/3d_examples/FlowPastRodCase/flow_past_rod_case.py:261-271,
FlowPastFreelyRotatingRodCase/flow_past_freely_rotating_rod_case.py:294-307), a constraint that pins the first node
and leaves some rotations free (``ea.GeneralConstraint``, :84-91), and ``ea.save_state`` / ``ea.load_state`` for
restarts (:212-227, 283-288).  ``RigidBodyIntegrator`` is the rigid counterpart of that loop for
``RigidSphereBody`` and ``StraightRodBody`` (forcing_grids.py): translation of the centre of mass and rotation of the
body frame under the net force and torque the forcing grid hands over, with the same position-Verlet order of
operations, an optional pinned node, and a state file that holds only arrays.

Conventions are the bodies' (PyElastica's): ``director_collection[:, :, e]`` has the material axes as ROWS in lab
coordinates (``Q @ lab = material``), ``omega_collection`` is in the material frame, node forces are in the lab
frame and element torques in the material frame.
"""
import numpy as np


def _rotation(rotvec):
    """Rodrigues matrix of the rotation vector (axis * angle)."""
    angle = float(np.linalg.norm(rotvec))
    if angle < 1e-300:
        return np.eye(3)
    k = np.asarray(rotvec, dtype=np.float64) / angle
    kx = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    return np.eye(3) + np.sin(angle) * kx + (1.0 - np.cos(angle)) * (kx @ kx)


class RigidBodyIntegrator:
    """Position-Verlet integration of one rigid body under ``interactor.body_flow_forces`` / ``body_flow_torques``.

    ``mass`` and ``inertia`` (three principal moments about the centre of mass, in the material frame of the first
    element) describe the body; ``pinned_node`` fixes that node in space (the body then turns about it);
    ``free_rotation`` says about which material axes the body may turn (the complement of
    ``rotational_constraint_selector``); ``external_force`` is a constant lab-frame force (gravity, a tether).
    """

    def __init__(self, body, interactor, mass, inertia, pinned_node=None, free_rotation=(True, True, True),
                 free_translation=(True, True, True), external_force=(0.0, 0.0, 0.0)):
        self.body = body
        self.interactor = interactor
        self.mass = float(mass)
        self.free_rotation = np.asarray(free_rotation, dtype=bool)
        self.free_translation = np.asarray(free_translation, dtype=bool)
        self.external_force = np.asarray(external_force, dtype=np.float64)
        self.pinned_node = pinned_node
        x = np.asarray(body.position_collection, dtype=np.float64)
        q0 = np.array(body.director_collection[:, :, 0], dtype=np.float64)
        self.centre = x.mean(axis=1) if x.shape[1] > 1 else x[:, 0].copy()
        # node offsets and element frames relative to the body frame: fixed for a rigid body
        self._node_offsets = q0 @ (x - self.centre.reshape(3, 1))
        self._element_frames = np.einsum("ijn,kj->ikn", body.director_collection, q0)   # Q_e Q_0^T
        self.frame = q0
        self.velocity = np.asarray(body.velocity_collection, dtype=np.float64).mean(axis=1)
        self.omega = np.array(body.omega_collection[:, 0], dtype=np.float64)
        inertia = np.asarray(inertia, dtype=np.float64).reshape(3)
        self._pivot_offset = np.zeros(3)
        if pinned_node is not None:
            # turn about the pinned node: parallel-axis theorem with the (material-frame) offset of the centre
            d = -self._node_offsets[:, pinned_node]
            self._pivot_offset = d
            self._pivot_position = x[:, pinned_node].copy()
            self.inertia_tensor = np.diag(inertia) + self.mass * (float(d @ d) * np.eye(3) - np.outer(d, d))
        else:
            self.inertia_tensor = np.diag(inertia)
        self._write_body()

    # ------------------------------------------------------------------ body arrays <- rigid state
    def _write_body(self):
        b, q = self.body, self.frame
        if self.pinned_node is not None:
            self.centre = self._pivot_position + q.T @ self._pivot_offset
            self.velocity = np.cross(q.T @ self.omega, self.centre - self._pivot_position)
        arms = q.T @ self._node_offsets
        b.position_collection[...] = self.centre.reshape(3, 1) + arms
        b.velocity_collection[...] = self.velocity.reshape(3, 1) + np.cross(q.T @ self.omega, arms, axisa=0, axisb=0,
                                                                            axisc=0)
        b.director_collection[...] = np.einsum("ikn,kj->ijn", self._element_frames, q)
        b.omega_collection[...] = np.einsum("ikn,k->in", self._element_frames, self.omega)

    def _kinematic_half_step(self, half_dt):
        if self.pinned_node is None:
            self.centre = self.centre + half_dt * np.where(self.free_translation, self.velocity, 0.0)
        # rows of the frame are the material axes: they turn with the body, Q <- Q R(omega_lab dt)^T = R(-omega_mat dt) Q
        self.frame = _rotation(-half_dt * self.omega) @ self.frame
        self._write_body()

    def _loads(self):
        """Net force (lab) and torque about the pivot or the centre of mass (material frame) from the interactor."""
        it = self.interactor
        it.compute_flow_forces_and_torques()
        forces = np.asarray(it.body_flow_forces, dtype=np.float64)
        torques = np.asarray(it.body_flow_torques, dtype=np.float64)
        q = self.frame
        about = self._pivot_position if self.pinned_node is not None else self.centre
        arms = np.asarray(self.body.position_collection) - about.reshape(3, 1)
        torque_lab = np.cross(arms, forces, axisa=0, axisb=0, axisc=0).sum(axis=1)
        torque_lab += np.einsum("jin,jn->i", self.body.director_collection, torques)      # sum of Q_e^T tau_e
        force = forces.sum(axis=1) + self.external_force
        torque_lab += np.cross(self.centre - about, self.external_force)
        return force, q @ torque_lab

    def do_step(self, time, dt):
        """One position-Verlet step (half kinematic step, loads and velocities, half kinematic step); returns the new
        time.  The caller advances the interactor's mismatch state by the same dt, as the reference's loop does."""
        self._kinematic_half_step(0.5 * dt)
        force, torque = self._loads()
        if self.pinned_node is None:
            self.velocity = self.velocity + dt * np.where(self.free_translation, force / self.mass, 0.0)
        j = self.inertia_tensor
        alpha = np.linalg.solve(j, torque - np.cross(self.omega, j @ self.omega))
        self.omega = np.where(self.free_rotation, self.omega + dt * alpha, 0.0)
        self._kinematic_half_step(0.5 * dt)
        return time + dt

    # ------------------------------------------------------------------ restart
    def save_state(self, path, time):
        """Rigid state and time as plain arrays (``numpy.savez``; nothing in the file is executable)."""
        np.savez(path, time=np.float64(time), centre=self.centre, velocity=self.velocity, frame=self.frame,
                 omega=self.omega)

    def load_state(self, path):
        with np.load(path, allow_pickle=False) as f:
            self.centre, self.velocity = f["centre"].copy(), f["velocity"].copy()
            self.frame, self.omega = f["frame"].copy(), f["omega"].copy()
            time = float(f["time"])
        self._write_body()
        return time


def sphere_mass_and_inertia(radius, density):
    mass = density * 4.0 / 3.0 * np.pi * radius**3
    return mass, np.full(3, 0.4 * mass * radius**2)


def rod_mass_and_inertia(length, radius, density):
    """Solid cylinder with its axis along the third material axis (the rod's tangent)."""
    mass = density * np.pi * radius**2 * length
    transverse = mass * (3.0 * radius**2 + length**2) / 12.0
    return mass, np.array([transverse, transverse, 0.5 * mass * radius**2])
