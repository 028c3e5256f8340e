"""Oracle (test infrastructure): 3D Eulerian stencil operators on ghosted local arrays.

Every function mirrors one ``gen_*_pyst_mpi_kernel_3d`` wrapper of the reference
(``sopht_mpi/numeric/eulerian_grid_ops/stencil_ops_3d/``) as it behaves on ONE
rank: arrays are C-ordered ``[z][y][x]`` and include ``g`` ghost layers; ``phys``
says which of the six faces are physical domain boundaries (the reference's
``MPI.PROC_NULL`` neighbours), ordered ``(z_prev, z_next, y_prev, y_next, x_prev,
x_next)``.  Ghost exchange is the caller's job (see ``oracle.decomp``), exactly as
in the reference where it is started before and finished after the interior pass.

The reference wrappers hand seven overlapping sub-boxes of the arrays to one
pystencils kernel; pystencils computes on each box minus ``support`` layers in
every direction.  That is restated literally here (``wrapper_boxes``) because it
decides which ghost cells get written, and those feed later operators.
"""
import numpy as np

ALL_PHYS = (True, True, True, True, True, True)


# --------------------------------------------------------------------------- boxes
def wrapper_boxes(shape, g, s, curl_interior_quirk=False):
    """Sub-boxes (tuples of slices over z, y, x) used by the MPI stencil wrappers.

    Follows diffusion_flux_mpi_3d.py:50-159 (identical slicing in
    update_vorticity_from_velocity_forcing_mpi_3d.py:49-176, curl_mpi_3d.py:44-161,
    divergence_mpi_3d.py, laplacian_filter_mpi_3d.py:158-265, advection_flux_mpi_3d.py:39-193).
    ``curl_interior_quirk``: curl_mpi_3d.py:44-48 slices only z and y of the interior
    box (x is left whole); harmless, but restated.
    """
    nz, ny, nx = shape
    inner = slice(g, -g)
    full = slice(None)

    def lo(n):
        return slice(g - s, g + 2 * s)

    def hi(n):
        return slice(n - (g + 2 * s), n - (g - s))

    interior = (inner, inner, full) if curl_interior_quirk else (inner, inner, inner)
    return [
        interior,
        (lo(nz), inner, inner),
        (hi(nz), inner, inner),
        (full, lo(ny), inner),
        (full, hi(ny), inner),
        (full, full, lo(nx)),
        (full, full, hi(nx)),
    ]


def zero_physical_bands(field, g, width, phys=ALL_PHYS):
    """Zero the outer ``g + width`` layers next to physical boundaries.

    diffusion_flux_mpi_3d.py:161-192, curl_mpi_3d.py:163-194,
    laplacian_filter_mpi_3d.py:110-143.  ``field`` is (z, y, x) or (c, z, y, x).
    """
    w = g + width
    zp, zn, yp, yn, xp, xn = phys
    if xp:
        field[..., :w] = 0
    if xn:
        field[..., -w:] = 0
    if yp:
        field[..., :w, :] = 0
    if yn:
        field[..., -w:, :] = 0
    if zp:
        field[..., :w, :, :] = 0
    if zn:
        field[..., -w:, :, :] = 0


# --------------------------------------------------------------------------- bodies
# Stencil bodies of the un-vendored ``sopht`` package, restated from name, support
# and call sites (parity unpinned).  Each computes on box[1:-1] in every direction.
def _c(a, dz, dy, dx):
    """View of ``a`` (last three axes z,y,x) shifted by (dz,dy,dx), cropped by one layer."""
    nz, ny, nx = a.shape[-3:]
    return a[..., 1 + dz : nz - 1 + dz, 1 + dy : ny - 1 + dy, 1 + dx : nx - 1 + dx]


def _diffusion_flux_body(flux, field, prefactor):
    """7-point Laplacian: flux = prefactor * (sum of 6 neighbours - 6 f)."""
    _c(flux, 0, 0, 0)[...] = prefactor * (
        _c(field, 0, 0, 1)
        + _c(field, 0, 0, -1)
        + _c(field, 0, 1, 0)
        + _c(field, 0, -1, 0)
        + _c(field, 1, 0, 0)
        + _c(field, -1, 0, 0)
        - 6 * _c(field, 0, 0, 0)
    )


def _curl_components(f, prefactor):
    """Central-difference curl of (3, z, y, x) ``f`` on the cropped box (x,y,z = comps 0,1,2)."""
    fx, fy, fz = f[0], f[1], f[2]
    cx = prefactor * (_c(fz, 0, 1, 0) - _c(fz, 0, -1, 0) - _c(fy, 1, 0, 0) + _c(fy, -1, 0, 0))
    cy = prefactor * (_c(fx, 1, 0, 0) - _c(fx, -1, 0, 0) - _c(fz, 0, 0, 1) + _c(fz, 0, 0, -1))
    cz = prefactor * (_c(fy, 0, 0, 1) - _c(fy, 0, 0, -1) - _c(fx, 0, 1, 0) + _c(fx, 0, -1, 0))
    return cx, cy, cz


def _curl_body(curl, f, prefactor):
    cx, cy, cz = _curl_components(f, prefactor)
    _c(curl[0], 0, 0, 0)[...] = cx
    _c(curl[1], 0, 0, 0)[...] = cy
    _c(curl[2], 0, 0, 0)[...] = cz


def _update_vorticity_body(vort, f, prefactor):
    cx, cy, cz = _curl_components(f, prefactor)
    _c(vort[0], 0, 0, 0)[...] += cx
    _c(vort[1], 0, 0, 0)[...] += cy
    _c(vort[2], 0, 0, 0)[...] += cz


def _divergence_body(div, f, inv_dx):
    """Central-difference divergence, (0.5 * inv_dx) * sum of differences.

    The 0.5 convention is not visible in the reference tree (parity unpinned); it is
    the one that makes ``inv_dx = 1/dx`` (flow_simulators_mpi_3d.py:454-458) a true
    divergence.
    """
    fx, fy, fz = f[0], f[1], f[2]
    _c(div, 0, 0, 0)[...] = (0.5 * inv_dx) * (
        _c(fx, 0, 0, 1)
        - _c(fx, 0, 0, -1)
        + _c(fy, 0, 1, 0)
        - _c(fy, 0, -1, 0)
        + _c(fz, 1, 0, 0)
        - _c(fz, -1, 0, 0)
    )


def _filter_body(axis):
    """laplacian_filter_mpi_3d.py:62-99: flux = 0.25 * (-f+ - f- + 2 f) along one axis."""
    d = [(0, 0, 1), (0, 1, 0), (1, 0, 0)][axis]  # axis 0 = x, 1 = y, 2 = z

    def body(flux, field):
        dtype = field.dtype.type
        _c(flux, 0, 0, 0)[...] = dtype(0.25) * (
            -_c(field, *d) - _c(field, -d[0], -d[1], -d[2]) + 2 * _c(field, 0, 0, 0)
        )

    return body


# --------------------------------------------------------------------------- operators
def diffusion_flux(flux, field, prefactor, g, phys=ALL_PHYS):
    """diffusion_flux_mpi_3d.py:35-192 (scalar)."""
    for box in wrapper_boxes(field.shape, g, 1):
        _diffusion_flux_body(flux[box], field[box], prefactor)
    zero_physical_bands(flux, g, 1, phys)


def diffusion_timestep(field, flux, nu_dt_by_dx2, g, phys=ALL_PHYS):
    """diffusion_timestep_mpi_3d.py:40-59: flux then field = field + flux (whole array)."""
    diffusion_flux(flux, field, nu_dt_by_dx2, g, phys)
    field[...] = field + flux


def vector_diffusion_timestep(vector_field, flux, nu_dt_by_dx2, g, phys=ALL_PHYS):
    """diffusion_timestep_mpi_3d.py:65-90: one scalar flux buffer reused per component."""
    for c in range(3):
        diffusion_timestep(vector_field[c], flux, nu_dt_by_dx2, g, phys)


def update_vorticity_from_velocity_forcing(vorticity, forcing, prefactor, g):
    """update_vorticity_from_velocity_forcing_mpi_3d.py:27-176: w += prefactor*curl(f).

    No band zeroing; writes the ghost cells the seven boxes cover.
    """
    full = (slice(None),)
    for box in wrapper_boxes(forcing.shape[1:], g, 1):
        _update_vorticity_body(vorticity[full + box], forcing[full + box], prefactor)


def update_vorticity_from_penalised_velocity(vorticity, penalised_velocity, velocity, prefactor, g):
    """update_vorticity_from_velocity_forcing_mpi_3d.py:181-330: w += prefactor * curl(u_pen - u) on the same
    seven boxes (the stencil body is in sopht: the difference is formed value by value, parity unpinned)."""
    update_vorticity_from_velocity_forcing(vorticity, penalised_velocity - velocity, prefactor, g)


def brinkmann_penalise(penalised, penalty_factor, char_field, penalty_field, field):
    """sopht gen_brinkmann_penalise_pyst_kernel_3d / 2d (call site brinkmann_penalise_mpi_3d.py:7-19): implicit
    Brinkmann penalisation on the whole array, scalar or vector fields sharing one characteristic function.
    Body not in the reference tree (parity unpinned): the standard form of the method."""
    dtype = field.dtype.type
    lam = dtype(penalty_factor)
    penalised[...] = (field + lam * char_field * penalty_field) / (dtype(1) + lam * char_field)


def char_func_from_level_set_via_sine_heaviside(char_func, level_set, blend_width):
    """sopht gen_char_func_from_level_set_via_sine_heaviside_pyst_kernel_3d / 2d (call site
    char_func_from_level_set_mpi_3d.py:8-28): smooth sine Heaviside function of the level set, whole array.
    Body not in the reference tree (parity unpinned)."""
    dtype = level_set.dtype.type
    w = dtype(blend_width)
    smooth = dtype(0.5) * (dtype(1) + level_set / w + np.sin(dtype(np.pi) * level_set / w) / dtype(np.pi))
    char_func[...] = np.where(level_set < -w, dtype(0), np.where(level_set > w, dtype(1), smooth))


def curl(curl_field, field, prefactor, g, phys=ALL_PHYS):
    """curl_mpi_3d.py:29-194."""
    full = (slice(None),)
    for box in wrapper_boxes(field.shape[1:], g, 1, curl_interior_quirk=True):
        _curl_body(curl_field[full + box], field[full + box], prefactor)
    zero_physical_bands(curl_field, g, 1, phys)


def divergence(div, field, inv_dx, g, phys=ALL_PHYS):
    """divergence_mpi_3d.py:29-198."""
    full = (slice(None),)
    for box in wrapper_boxes(field.shape[1:], g, 1):
        _divergence_body(div[box], field[full + box], inv_dx)
    zero_physical_bands(div, g, 1, phys)


def elementwise_cross_product(result, f1, f2):
    """sopht gen_elementwise_cross_product_pyst_kernel_3d (call site
    flow_simulators_mpi_3d.py:396-401): result = f1 x f2 on the whole array."""
    r0 = f1[1] * f2[2] - f1[2] * f2[1]
    r1 = f1[2] * f2[0] - f1[0] * f2[2]
    r2 = f1[0] * f2[1] - f1[1] * f2[0]
    result[0], result[1], result[2] = r0, r1, r2


def penalise_field_boundary(field, width, dx, x_coords, y_coords, z_coords, g, phys=ALL_PHYS):
    """penalise_field_boundary_mpi_3d.py:185-242 (scalar field, in place).

    ``*_coords`` are the 1D local cell-centre coordinates including ghosts
    (flow_simulators_mpi_3d.py:139-153).  Order: x front, x back, y, z.
    """
    if width == 0:
        return
    real_t = field.dtype.type
    w = g + width
    sine_prefactor = real_t((np.pi / 2) / (width * dx))
    zp, zn, yp, yn, xp, xn = phys
    x0, x1 = x_coords[g], x_coords[-(g + 1)]
    y0, y1 = y_coords[g], y_coords[-(g + 1)]
    z0, z1 = z_coords[g], z_coords[-(g + 1)]
    if xp:
        field[:, :, :w] = field[:, :, w - 1 : w]
        field[:, :, :w] *= np.sin(sine_prefactor * (x_coords[:w] - x0))
    if xn:
        field[:, :, -w:] = field[:, :, -w : -w + 1]
        field[:, :, -w:] *= np.sin(sine_prefactor * (x1 - x_coords[-w:]))
    if yp:
        field[:, :w, :] = field[:, w - 1 : w, :]
        field[:, :w, :] *= np.sin(sine_prefactor * (y_coords[:w] - y0))[None, :, None]
    if yn:
        field[:, -w:, :] = field[:, -w : -w + 1, :]
        field[:, -w:, :] *= np.sin(sine_prefactor * (y1 - y_coords[-w:]))[None, :, None]
    if zp:
        field[:w, :, :] = field[w - 1 : w, :, :]
        field[:w, :, :] *= np.sin(sine_prefactor * (z_coords[:w] - z0))[:, None, None]
    if zn:
        field[-w:, :, :] = field[-w : -w + 1, :, :]
        field[-w:, :, :] *= np.sin(sine_prefactor * (z1 - z_coords[-w:]))[:, None, None]


def penalise_vector_field_boundary(vfield, width, dx, x_coords, y_coords, z_coords, g, phys=ALL_PHYS):
    for c in range(3):
        penalise_field_boundary(vfield[c], width, dx, x_coords, y_coords, z_coords, g, phys)


def _filter_pass(axis, flux_buf, field_buf, g, exchange=None):
    """laplacian_filter_mpi_3d.py:145-265: one directional pass with its own halo exchange."""
    if exchange is not None:
        exchange(field_buf)
    body = _filter_body(axis)
    for box in wrapper_boxes(field_buf.shape, g, 1):
        body(flux_buf[box], field_buf[box])


def laplacian_filter(field, flux_buf, field_buf, order, g, filter_type="multiplicative",
                     phys=ALL_PHYS, exchange=None, boundary_width=1):
    """laplacian_filter_mpi_3d.py:267-385 (scalar, in place). ``exchange(arr)`` ghosts one scalar;
    ``boundary_width`` is filter_flux_buffer_boundary_width (:110-143)."""
    w = boundary_width
    if filter_type == "multiplicative":
        zero_physical_bands(flux_buf, g, w, phys)
        field_buf[...] = field
        for _ in range(order):
            for axis in (0, 1, 2):  # x, y, z
                _filter_pass(axis, flux_buf, field_buf, g, exchange)
                zero_physical_bands(flux_buf, g, w, phys)
                field_buf[...] = flux_buf
        field[...] = field - flux_buf
    elif filter_type == "convolution":
        zero_physical_bands(flux_buf, g, w, phys)
        for axis in (0, 1, 2):
            field_buf[...] = field
            for _ in range(order):
                _filter_pass(axis, flux_buf, field_buf, g, exchange)
                zero_physical_bands(flux_buf, g, w, phys)
                field_buf[...] = flux_buf
            field[...] = field - flux_buf
    else:
        raise ValueError("Invalid filter type")


def vector_laplacian_filter(vfield, flux_buf, field_buf, order, g, filter_type="multiplicative",
                            phys=ALL_PHYS, exchange=None, boundary_width=1):
    for c in range(3):
        laplacian_filter(vfield[c], flux_buf, field_buf, order, g, filter_type, phys, exchange, boundary_width)


# ENO3 advection (body lives in ``sopht``; parity unpinned) ---------------------------
def _eno3_face_flux(f, vel, axis):
    """Third-order upwind-biased reconstruction of f at the +axis face, times face velocity.

    Face velocity = mean of the two adjacent cell values.  Returned on the box cropped
    by two layers at the low end and two at the high end along every direction, for the
    face between cell i and i+1 where i runs over [1, n-3] (so both faces of the cells
    [2, n-3] are available).  axis: 0 = x, 1 = y, 2 = z.
    """
    ax = {0: -1, 1: -2, 2: -3}[axis]

    def sh(a, k):
        # cells i+k for i in [1, n-3]
        n = a.shape[ax]
        sl = [slice(None)] * a.ndim
        sl[ax] = slice(1 + k, n - 2 + k)
        return a[tuple(sl)]

    t = f.dtype.type
    face_vel = t(0.5) * (sh(vel, 0) + sh(vel, 1))
    left = t(-1 / 6) * sh(f, -1) + t(5 / 6) * sh(f, 0) + t(1 / 3) * sh(f, 1)
    right = t(1 / 3) * sh(f, 0) + t(5 / 6) * sh(f, 1) + t(-1 / 6) * sh(f, 2)
    return face_vel * np.where(face_vel > 0, left, right)


def _advection_flux_body(flux, field, velocity, inv_dx):
    """Conservative flux divergence, support 2: computes on box[2:-2]^3."""
    total = 0
    for axis in (0, 1, 2):
        ff = _eno3_face_flux(field, velocity[axis], axis)
        ax = {0: -1, 1: -2, 2: -3}[axis]
        n = ff.shape[ax]
        hi = [slice(2, -2)] * 3
        lo = [slice(2, -2)] * 3
        hi[ax] = slice(1, n)
        lo[ax] = slice(0, n - 1)
        total = total + (ff[tuple(hi)] - ff[tuple(lo)])
    flux[2:-2, 2:-2, 2:-2] = inv_dx * total


def advection_flux_eno3(flux, field, velocity, inv_dx, g):
    """advection_flux_mpi_3d.py:25-193; no band zeroing."""
    full = (slice(None),)
    for box in wrapper_boxes(field.shape, g, 2):
        _advection_flux_body(flux[box], field[box], velocity[full + box], inv_dx)


def advection_timestep_eno3(field, flux, velocity, dt_by_dx, g):
    """advection_timestep_mpi_3d.py:40-60."""
    flux[...] = 0
    advection_flux_eno3(flux, field, velocity, -dt_by_dx, g)
    field[...] = field + flux
