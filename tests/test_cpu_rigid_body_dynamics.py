"""Host logic of the rigid-body stand-in integrator (simulator/immersed_body/rigid_body_dynamics.py) against closed
forms: constant force, constant torque about a principal axis, a pinned rod, torque-free precession, state files."""
import numpy as np
import pytest

from sopht_mpi_amd.simulator.immersed_body.forcing_grids import RigidSphereBody, StraightRodBody
from sopht_mpi_amd.simulator.immersed_body.rigid_body_dynamics import (RigidBodyIntegrator, rod_mass_and_inertia,
                                                                      sphere_mass_and_inertia)


class ConstantLoads:
    """Stands in for an interactor: the same node forces (lab frame) and element torques (material frame) on every call."""

    def __init__(self, forces, torques):
        self.body_flow_forces, self.body_flow_torques = np.array(forces, float), np.array(torques, float)
        self.calls = 0

    def compute_flow_forces_and_torques(self):
        self.calls += 1


def test_sphere_under_constant_force_and_torque():
    body = RigidSphereBody(center=(0.1, 0.2, 0.3), base_radius=0.05)
    mass, inertia = sphere_mass_and_inertia(0.05, 2.0)
    loads = ConstantLoads([[0.3], [0.0], [-0.1]], [[0.0], [0.0], [2e-4]])
    stepper = RigidBodyIntegrator(body, loads, mass, inertia)
    t, dt, n = 0.0, 1e-3, 200
    for _ in range(n):
        t = stepper.do_step(t, dt)
    assert loads.calls == n and t == pytest.approx(n * dt)
    a = np.array([0.3, 0.0, -0.1]) / mass
    np.testing.assert_allclose(body.position_collection[:, 0], np.array([0.1, 0.2, 0.3]) + 0.5 * a * t**2, rtol=1e-12)
    np.testing.assert_allclose(body.velocity_collection[:, 0], a * t, rtol=1e-12)
    # spin about the third material axis: omega = tau t / J, the frame has turned by half of omega t
    np.testing.assert_allclose(body.omega_collection[:, 0], [0.0, 0.0, 2e-4 * t / inertia[2]], rtol=1e-12, atol=1e-15)
    angle = 0.5 * (2e-4 / inertia[2]) * t**2
    q = body.director_collection[:, :, 0]
    np.testing.assert_allclose(q[0], [np.cos(angle), np.sin(angle), 0.0], atol=1e-12)   # d1 turns counter-clockwise
    np.testing.assert_allclose(q @ q.T, np.eye(3), atol=1e-13)


def test_pinned_rod_turns_about_its_first_node():
    rod = StraightRodBody(n_elements=10, start=(0.2, 0.5, 0.5), direction=(1.0, 0.0, 0.0), normal=(0.0, 1.0, 0.0),
                          base_length=0.4, base_radius=0.01)
    mass, inertia = rod_mass_and_inertia(0.4, 0.01, 3.0)
    forces = np.zeros((3, 11))
    forces[1, -1] = 0.002         # a transverse force on the free end
    loads = ConstantLoads(forces, np.zeros((3, 10)))
    stepper = RigidBodyIntegrator(rod, loads, mass, inertia, pinned_node=0)
    t, dt = 0.0, 2e-4
    for _ in range(100):
        t = stepper.do_step(t, dt)
    np.testing.assert_allclose(rod.position_collection[:, 0], [0.2, 0.5, 0.5], atol=1e-15)   # the pin holds
    lengths = np.linalg.norm(np.diff(rod.position_collection, axis=1), axis=0)
    np.testing.assert_allclose(lengths, 0.04, rtol=1e-12)                                    # rigid
    j_pivot = inertia[0] + mass * 0.2**2
    # small angles: the lever arm stays 0.4 to second order
    want = 0.002 * 0.4 / j_pivot * t
    omega_lab = rod.director_collection[:, :, 0].T @ rod.omega_collection[:, 0]
    np.testing.assert_allclose(omega_lab, [0.0, 0.0, want], rtol=2e-4, atol=1e-12)
    tip_speed = rod.velocity_collection[:, -1]
    np.testing.assert_allclose(np.linalg.norm(tip_speed), want * 0.4, rtol=2e-4)


def test_torque_free_rotation_keeps_angular_momentum_and_constraints_hold():
    body = RigidSphereBody(center=(0.0, 0.0, 0.0), base_radius=0.1)
    loads = ConstantLoads(np.zeros((3, 1)), np.zeros((3, 1)))
    inertia = np.array([1.0, 2.0, 3.0]) * 1e-3
    body.omega_collection[:, 0] = (3.0, -1.0, 2.0)
    stepper = RigidBodyIntegrator(body, loads, 1.0, inertia)
    l0 = stepper.frame.T @ (inertia * stepper.omega)
    e0 = 0.5 * float(stepper.omega @ (inertia * stepper.omega))
    t = 0.0
    for _ in range(2000):
        t = stepper.do_step(t, 1e-4)
    l1 = stepper.frame.T @ (inertia * stepper.omega)
    np.testing.assert_allclose(l1, l0, rtol=2e-3)      # first-order scheme in omega: drift, no blow-up
    assert 0.5 * float(stepper.omega @ (inertia * stepper.omega)) == pytest.approx(e0, rel=2e-3)
    # constrained: only rotation about the third material axis, no translation along y
    body2 = RigidSphereBody(center=(0.0, 0.0, 0.0), base_radius=0.1)
    loads2 = ConstantLoads([[1.0], [1.0], [1.0]], [[1.0], [1.0], [1.0]])
    s2 = RigidBodyIntegrator(body2, loads2, 1.0, inertia, free_rotation=(False, False, True),
                             free_translation=(True, False, True))
    for _ in range(10):
        s2.do_step(0.0, 1e-3)
    assert body2.position_collection[1, 0] == 0.0 and body2.position_collection[0, 0] > 0
    assert np.all(body2.omega_collection[:2, 0] == 0.0) and body2.omega_collection[2, 0] > 0


def test_state_file_round_trip(tmp_path):
    rod = StraightRodBody(n_elements=6, start=(0.1, 0.1, 0.1), direction=(0.0, 1.0, 0.0), normal=(0.0, 0.0, 1.0),
                          base_length=0.3, base_radius=0.02)
    loads = ConstantLoads(np.full((3, 7), 0.01), np.full((3, 6), 1e-4))
    mass, inertia = rod_mass_and_inertia(0.3, 0.02, 1.5)
    a = RigidBodyIntegrator(rod, loads, mass, inertia)
    t = 0.0
    for _ in range(20):
        t = a.do_step(t, 1e-3)
    path = str(tmp_path / "body_state.npz")
    a.save_state(path, t)
    rod_b = StraightRodBody(n_elements=6, start=(0.1, 0.1, 0.1), direction=(0.0, 1.0, 0.0), normal=(0.0, 0.0, 1.0),
                            base_length=0.3, base_radius=0.02)
    b = RigidBodyIntegrator(rod_b, loads, mass, inertia)
    assert b.load_state(path) == t
    for name in ("position_collection", "velocity_collection", "director_collection", "omega_collection"):
        assert np.array_equal(getattr(rod_b, name), getattr(rod, name)), name
    for _ in range(5):
        a.do_step(t, 1e-3)
        b.do_step(t, 1e-3)
    assert np.array_equal(rod_b.position_collection, rod.position_collection)
