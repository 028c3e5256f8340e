"""CPU: geometry and force transfer of the sphere / rod forcing grids (host-side numpy).
The reference takes these classes from the un-vendored sopht package, so there is nothing to
compare node placement against; the checks are geometric and mechanical invariants."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def ib(smi):
    from sopht_mpi_amd.simulator import immersed_body

    return immersed_body


def _rotation(axis, angle):
    axis = np.asarray(axis, dtype=float) / np.linalg.norm(axis)
    k = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * k + (1 - np.cos(angle)) * (k @ k)


def test_sphere_grid_geometry_and_kinematics(ib):
    body = ib.RigidSphereBody(center=(0.25, 0.5, 0.5), base_radius=0.2)
    n_eq = 192
    grid = ib.SphereForcingGrid(grid_dim=3, rigid_body=body, num_forcing_points_along_equator=n_eq)
    r = np.linalg.norm(grid.position_field - body.position_collection, axis=0)
    np.testing.assert_allclose(r, 0.2, rtol=1e-12)
    spacing = grid.get_maximum_lagrangian_grid_spacing()
    assert abs(spacing - 2 * np.pi * 0.2 / n_eq) < 1e-15
    # surface area per node ~ spacing^2
    assert abs(4 * np.pi * 0.2**2 / grid.num_lag_nodes / spacing**2 - 1) < 0.05
    # nearly uniform: centroid at the centre
    assert np.linalg.norm(grid.position_field.mean(axis=1) - body.position_collection[:, 0]) < 1e-3
    # rigid motion: translate, rotate the frame, spin about the material z axis
    q = _rotation((1, 2, 3), 0.7)            # rows = material axes in lab coordinates
    body.director_collection[:, :, 0] = q
    body.velocity_collection[:, 0] = (0.1, -0.2, 0.3)
    body.omega_collection[:, 0] = (0.0, 0.0, 2.0)
    grid.compute_lag_grid_position_field()
    grid.compute_lag_grid_velocity_field()
    arm = grid.position_field - body.position_collection
    omega_lab = q.T @ body.omega_collection[:, 0]
    want = body.velocity_collection + np.cross(omega_lab, arm, axisa=0, axisb=0, axisc=0)
    np.testing.assert_allclose(grid.velocity_field, want, atol=1e-14)
    np.testing.assert_allclose(np.linalg.norm(arm, axis=0), 0.2, rtol=1e-12)


def test_sphere_force_and_torque_transfer(ib):
    body = ib.RigidSphereBody(center=(0.5, 0.5, 0.5), base_radius=0.1)
    grid = ib.SphereForcingGrid(grid_dim=3, rigid_body=body, num_forcing_points_along_equator=64)
    m = grid.num_lag_nodes
    forces, torques = np.zeros((3, 1)), np.zeros((3, 1))
    uniform = np.tile(np.array([[1.0], [0.5], [-2.0]]), (1, m))
    grid.transfer_forcing_from_grid_to_body(forces, torques, uniform)
    np.testing.assert_allclose(forces[:, 0], -m * np.array([1.0, 0.5, -2.0]))
    assert np.linalg.norm(torques) < 1e-3 * m * 0.1          # symmetric grid: no net moment
    # tangential forcing omega x r gives the torque -sum r x (omega x r) = -(2/3) M R^2 omega (sphere shell)
    omega = np.array([0.0, 0.0, 1.0])
    arm = grid.position_field - body.position_collection
    swirl = np.cross(omega, arm, axisa=0, axisb=0, axisc=0)
    grid.transfer_forcing_from_grid_to_body(forces, torques, swirl)
    assert abs(torques[2, 0] / (-(2.0 / 3.0) * m * 0.1**2) - 1) < 0.02
    assert np.linalg.norm(forces) < 1e-3 * m * 0.1


def test_rod_grids(ib):
    rod = ib.StraightRodBody(n_elements=20, start=(0.2, 0.5, 0.5), direction=(1, 0, 0), normal=(0, 1, 0),
                             base_length=0.5, base_radius=0.02)
    centric = ib.CosseratRodElementCentricForcingGrid(grid_dim=3, cosserat_rod=rod)
    assert centric.num_lag_nodes == 20
    np.testing.assert_allclose(centric.position_field[0], 0.2 + (np.arange(20) + 0.5) * 0.025)
    forces, torques = np.zeros((3, 21)), np.zeros((3, 20))
    f = np.random.default_rng(0).standard_normal((3, 20))
    centric.transfer_forcing_from_grid_to_body(forces, torques, f)
    np.testing.assert_allclose(forces.sum(axis=1), -f.sum(axis=1), atol=1e-13)
    assert not torques.any()
    surface = ib.CosseratRodSurfaceForcingGrid(grid_dim=3, cosserat_rod=rod,
                                               surface_grid_density_for_largest_element=16)
    assert surface.num_lag_nodes == 20 * 16
    axis_dist = np.linalg.norm(surface.position_field[1:] - np.array([[0.5], [0.5]]), axis=0)
    np.testing.assert_allclose(axis_dist, 0.02, rtol=1e-12)
    # spin about the rod axis (material d3): surface speed = omega * radius
    rod.omega_collection[2, :] = 3.0
    surface.compute_lag_grid_velocity_field()
    np.testing.assert_allclose(np.linalg.norm(surface.velocity_field, axis=0), 3.0 * 0.02, rtol=1e-12)
    assert np.abs(surface.velocity_field[0]).max() < 1e-15
    # tangential unit forcing on every node: torque about d3 per element = -16 * radius
    tang = surface.velocity_field / (3.0 * 0.02)
    forces, torques = np.zeros((3, 21)), np.zeros((3, 20))
    surface.transfer_forcing_from_grid_to_body(forces, torques, tang)
    np.testing.assert_allclose(torques[2], -16 * 0.02, rtol=1e-12)
    assert np.abs(forces).max() < 1e-13
    assert surface.get_maximum_lagrangian_grid_spacing() == pytest.approx(0.025)
