"""Oracle (test infrastructure): 2D Eulerian stencil operators on ghosted ``[y][x]`` arrays,
single rank or one rank of a decomposition (``phys`` = which of (y_prev, y_next, x_prev,
x_next) are physical boundaries).  Mirrors ``sopht_mpi/numeric/eulerian_grid_ops/stencil_ops_2d``:
five overlapping boxes per call (diffusion_flux_mpi_2d.py:47-113), then zeroing of the
``g + 1`` physical bands (:116-136).  Stencil bodies live in the un-vendored ``sopht`` package
and are restated from names, supports and docstrings (parity unpinned), as in 3D.
"""
import numpy as np

ALL_PHYS = (True, True, True, True)


def wrapper_boxes(shape, g, s):
    inner, full = slice(g, -g), slice(None)
    lo = slice(g - s, g + 2 * s)

    def hi(n):
        return slice(n - (g + 2 * s), n - (g - s))

    ny, nx = shape
    return [(inner, inner), (lo, inner), (hi(ny), inner), (full, lo), (full, hi(nx))]


def zero_physical_bands(field, g, width, phys=ALL_PHYS):
    w = g + width
    yp, yn, xp, xn = phys
    if xp:
        field[..., :w] = 0
    if xn:
        field[..., -w:] = 0
    if yp:
        field[..., :w, :] = 0
    if yn:
        field[..., -w:, :] = 0


def _sh(a, dy, dx, s=1):
    ny, nx = a.shape[-2:]
    return a[..., s + dy : ny - s + dy, s + dx : nx - s + dx]


def diffusion_flux(flux, field, prefactor, g, phys=ALL_PHYS):
    """diffusion_flux_mpi_2d.py:30-136: flux = c (sum of 4 neighbours - 4 f), zero on the bands."""
    for box in wrapper_boxes(field.shape, g, 1):
        f = field[box]
        flux[box][1:-1, 1:-1] = prefactor * (_sh(f, 0, 1) + _sh(f, 0, -1) + _sh(f, 1, 0) + _sh(f, -1, 0)
                                             - field.dtype.type(4) * _sh(f, 0, 0))
    zero_physical_bands(flux, g, 1, phys)


def diffusion_timestep(field, flux, nu_dt_by_dx2, g, phys=ALL_PHYS):
    """diffusion_timestep_mpi_2d.py."""
    diffusion_flux(flux, field, nu_dt_by_dx2, g, phys)
    field[...] = field + flux


def _upwind3_face(f, vel, axis):
    """Face flux between cells i and i+1 along axis (0 = x, 1 = y) for i in [1, n-3]."""
    ax = -1 - axis

    def sh(a, k):
        n = a.shape[ax]
        sl = [slice(None)] * a.ndim
        sl[ax] = slice(1 + k, n - 2 + k)
        return a[tuple(sl)]

    t = f.dtype.type
    face_vel = t(0.5) * (sh(vel, 0) + sh(vel, 1))
    left = t(-1 / 6) * sh(f, -1) + t(5 / 6) * sh(f, 0) + t(1 / 3) * sh(f, 1)
    right = t(1 / 3) * sh(f, 0) + t(5 / 6) * sh(f, 1) + t(-1 / 6) * sh(f, 2)
    return face_vel * np.where(face_vel > 0, left, right)


def advection_timestep_eno3(field, flux, velocity, dt_by_dx, g):
    """advection_timestep_mpi_2d.py / advection_flux_mpi_2d.py: flux = 0; flux = -dt/dx div(F); f += flux
    with the same third-order upwind face flux as the 3D restatement (support 2, no band zeroing)."""
    flux[...] = 0
    for box in wrapper_boxes(field.shape, g, 2):
        f = field[box]
        total = 0
        for axis in (0, 1):
            ff = _upwind3_face(f, velocity[(axis,) + box], axis)
            ax = -1 - axis
            n = ff.shape[ax]
            hi = [slice(2, -2)] * 2
            lo = [slice(2, -2)] * 2
            hi[ax] = slice(1, n)
            lo[ax] = slice(0, n - 1)
            total = total + (ff[tuple(hi)] - ff[tuple(lo)])
        flux[box][2:-2, 2:-2] = field.dtype.type(-dt_by_dx) * total
    field[...] = field + flux


def outplane_field_curl(curl, field, prefactor, g, phys=ALL_PHYS):
    """outplane_field_curl_mpi_2d.py:30-139 (psi -> velocity): u = p d(psi)/dy, v = -p d(psi)/dx."""
    for box in wrapper_boxes(field.shape, g, 1):
        f = field[box]
        curl[(0,) + box][1:-1, 1:-1] = prefactor * (_sh(f, 1, 0) - _sh(f, -1, 0))
        curl[(1,) + box][1:-1, 1:-1] = -prefactor * (_sh(f, 0, 1) - _sh(f, 0, -1))
    zero_physical_bands(curl, g, 1, phys)


def update_vorticity_from_velocity_forcing(vorticity, forcing, prefactor, g):
    """update_vorticity_from_velocity_forcing_mpi_2d.py: w += p (d f_y / dx - d f_x / dy), no band zeroing.
    The boxes overlap only in cells they both leave alone, so each updated cell is updated once."""
    done = np.zeros(vorticity.shape, dtype=bool)
    for box in wrapper_boxes(vorticity.shape, g, 1):
        fx, fy = forcing[(0,) + box], forcing[(1,) + box]
        upd = prefactor * ((_sh(fy, 0, 1) - _sh(fy, 0, -1)) - (_sh(fx, 1, 0) - _sh(fx, -1, 0)))
        mask = ~done[box][1:-1, 1:-1]
        vorticity[box][1:-1, 1:-1] += np.where(mask, upd, 0)
        done[box][1:-1, 1:-1] = True


def penalise_field_boundary(field, width, dx, x_coords, y_coords, g, phys=ALL_PHYS):
    """penalise_field_boundary_mpi_2d.py:139-170: x front, x back, y front, y back."""
    if width == 0:
        return
    real_t = field.dtype.type
    w = g + width
    sine_prefactor = real_t((np.pi / 2) / (width * dx))
    yp, yn, xp, xn = phys
    x0, x1 = x_coords[g], x_coords[-(g + 1)]
    y0, y1 = y_coords[g], y_coords[-(g + 1)]
    if xp:
        field[:, :w] = field[:, w - 1 : w]
        field[:, :w] *= np.sin(sine_prefactor * (x_coords[:w] - x0))
    if xn:
        field[:, -w:] = field[:, -w : -w + 1]
        field[:, -w:] *= np.sin(sine_prefactor * (x1 - x_coords[-w:]))
    if yp:
        field[:w, :] = field[w - 1 : w, :]
        field[:w, :] *= np.sin(sine_prefactor * (y_coords[:w] - y0))[:, None]
    if yn:
        field[-w:, :] = field[-w : -w + 1, :]
        field[-w:, :] *= np.sin(sine_prefactor * (y1 - y_coords[-w:]))[:, None]
