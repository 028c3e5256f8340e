"""Closed-form fields for the analytic known-answer tests (helper module, no tests of its own).

Every case leaves no freedom in sign, axis order or prefactor of the operator it drives: a swapped
component, a flipped sign or a factor of two gives an error of order one, the checks ask for a
second-order small one.  Coordinates follow flow_simulators_mpi_3d.py:124-158: arrays are
[z][y][x], cell centres (i + 1/2) dx continued into the ghost cells, vector component 0 = x.
"""
import numpy as np

TWO_PI = 2.0 * np.pi


def padded_coordinates(shape, g, dx):
    """(x, y, z) coordinate arrays of the ghosted [z][y][x] grid, float64."""
    nz, ny, nx = shape
    ax = [(np.arange(-g, n + g) + 0.5) * float(dx) for n in (nz, ny, nx)]
    z, y, x = np.meshgrid(*ax, indexing="ij")
    return x, y, z


# ------------------------------------------------------------------ trigonometric vector field
# f_c = A_c sin(2 pi (k_c . r) + phase_c) with distinct integer wave vectors: every partial derivative
# of every component is non-zero and different.
_K = np.array([[1.0, 2.0, -1.0], [-2.0, 1.0, 1.0], [1.0, -1.0, 2.0]])  # rows: component, columns: (kx, ky, kz)
_A = np.array([1.0, -0.7, 0.45])
_PH = np.array([0.3, 1.1, -0.4])


def trig_field(x, y, z):
    arg = [TWO_PI * (_K[c, 0] * x + _K[c, 1] * y + _K[c, 2] * z) + _PH[c] for c in range(3)]
    return np.stack([_A[c] * np.sin(arg[c]) for c in range(3)])


def _trig_grad(x, y, z):
    """d[c][a] = d f_c / d r_a, a = x, y, z."""
    arg = [TWO_PI * (_K[c, 0] * x + _K[c, 1] * y + _K[c, 2] * z) + _PH[c] for c in range(3)]
    return [[_A[c] * TWO_PI * _K[c, a] * np.cos(arg[c]) for a in range(3)] for c in range(3)]


def trig_curl(x, y, z):
    d = _trig_grad(x, y, z)
    return np.stack([d[2][1] - d[1][2], d[0][2] - d[2][0], d[1][0] - d[0][1]])


def trig_divergence(x, y, z):
    d = _trig_grad(x, y, z)
    return d[0][0] + d[1][1] + d[2][2]


def trig_laplacian(x, y, z):
    f = trig_field(x, y, z)
    return np.stack([-(TWO_PI**2) * np.sum(_K[c] ** 2) * f[c] for c in range(3)])


# ------------------------------------------------------------------ divergence-free Gaussian stream function
# psi = grad(phi) x A, phi = exp(-|r - c|^2 / (2 s^2)): div psi = 0, so u = curl psi has the vorticity
# w = curl u = -lap psi, and the free-space solution of -lap psi' = w that decays at infinity is psi itself.
BLOB_A = np.array([0.3, -0.5, 0.8])        # (x, y, z) components
BLOB_C = np.array([0.46, 0.55, 0.5])
BLOB_S = 0.07


def blob_fields(x, y, z):
    """(vorticity, stream function, velocity), each (3, z, y, x) with component 0 = x."""
    s2 = BLOB_S**2
    d = np.stack([x - BLOB_C[0], y - BLOB_C[1], z - BLOB_C[2]])
    r2 = np.sum(d * d, axis=0)
    phi = np.exp(-r2 / (2 * s2))
    a = BLOB_A.reshape(3, 1, 1, 1)
    grad_phi = -d / s2 * phi
    psi = np.cross(grad_phi, np.broadcast_to(a, d.shape), axis=0)
    lap_phi = phi * (r2 / s2**2 - 3.0 / s2)
    grad_lap_phi = phi * d * (5.0 / s2**2 - r2 / s2**3)
    vorticity = np.cross(np.broadcast_to(a, d.shape), grad_lap_phi, axis=0)   # -(grad lap phi) x A
    a_dot_d = np.sum(a * d, axis=0)
    velocity = -a * lap_phi + phi * (a_dot_d * d / s2**2 - a / s2)
    return vorticity, psi, velocity


# ------------------------------------------------------------------ straight Gaussian vortex tube
def lamb_oseen_tube(x, y, z, axis, centre, four_nu_t, gamma, half_length):
    """Vorticity (3, ...) of a straight tube along `axis` (0 = x, 1 = y, 2 = z) through `centre` with the
    Lamb-Oseen core of lamb_oseen_helpers.py:4-21, and its velocity by :24-50 (right-handed rotation about
    the tube axis) times the factor of a vortex segment of finite length, (cos t1 - cos t2) / 2 with the
    angles under which the two ends are seen (Biot-Savart for a straight segment): the unbounded solve sees
    only the part of the tube inside the domain.  `half_length`: distance from the tube's middle to its ends;
    the velocity is evaluated for points whose axial coordinate is `s` away from the middle."""
    r = [x, y, z]
    # the two in-plane axes in right-handed order: (a, b, axis) is a cyclic permutation of (x, y, z)
    a, b = [(1, 2), (2, 0), (0, 1)][axis]
    da, db = r[a] - centre[a], r[b] - centre[b]
    s = r[axis] - centre[axis]
    rho2 = da * da + db * db
    rho = np.sqrt(rho2)
    w = np.zeros((3,) + x.shape)
    w[axis] = gamma / (np.pi * four_nu_t) * np.exp(-rho2 / four_nu_t)
    with np.errstate(divide="ignore", invalid="ignore"):
        v_theta = gamma / (TWO_PI * rho) * (1.0 - np.exp(-rho2 / four_nu_t))
        segment = 0.5 * ((half_length - s) / np.sqrt(rho2 + (half_length - s) ** 2)
                         + (half_length + s) / np.sqrt(rho2 + (half_length + s) ** 2))
        u = np.zeros((3,) + x.shape)
        u[a] = np.where(rho > 0, -v_theta * db / rho * segment, 0.0)
        u[b] = np.where(rho > 0, v_theta * da / rho * segment, 0.0)
    return w, u
