"""Eulerian grid operators with the reference's generator API
(sopht_mpi/numeric/eulerian_grid_ops/__init__.py:1-5): every ``gen_*`` returns a callable
that takes DeviceField arguments by keyword, carries ``.kernel_support`` and validates
ghost size / arguments at generation time with the reference's exception types.
"""
import ctypes as C

import numpy as np

from ... import _lib
from ..._lib import call, darray
from ...device import as_ptr
from ...utils import check_valid_ghost_size_and_kernel_support


def _support(gen, fn, kernel_support, ghost_exchange_communicator):
    gen.kernel_support = kernel_support
    fn.kernel_support = kernel_support
    check_valid_ghost_size_and_kernel_support(ghost_exchange_communicator.ghost_size, kernel_support)
    return fn


def _check_field_type(field_type):
    if field_type != "scalar" and field_type != "vector":
        raise ValueError("Invalid field type")


# ------------------------------------------------------------------ diffusion
def gen_diffusion_flux_pyst_mpi_kernel_3d(real_t, mpi_construct, ghost_exchange_communicator, field_type="scalar"):
    """stencil_ops_3d/diffusion_flux_mpi_3d.py:10-228."""
    _check_field_type(field_type)
    ctx = ghost_exchange_communicator.ctx

    if field_type == "scalar":

        def diffusion_flux_pyst_mpi_kernel_3d(diffusion_flux, field, prefactor):
            call("smi_diffusion_flux", ctx.handle, as_ptr(diffusion_flux), as_ptr(field), float(prefactor), 1)

        fn = diffusion_flux_pyst_mpi_kernel_3d
    else:

        def vector_field_diffusion_flux_pyst_mpi_kernel_3d(vector_field_diffusion_flux, vector_field, prefactor):
            call("smi_diffusion_flux", ctx.handle, as_ptr(vector_field_diffusion_flux), as_ptr(vector_field),
                 float(prefactor), 3)

        fn = vector_field_diffusion_flux_pyst_mpi_kernel_3d
    return _support(gen_diffusion_flux_pyst_mpi_kernel_3d, fn, 1, ghost_exchange_communicator)


def gen_diffusion_timestep_euler_forward_pyst_mpi_kernel_3d(real_t, mpi_construct, ghost_exchange_communicator,
                                                            field_type="scalar"):
    """stencil_ops_3d/diffusion_timestep_mpi_3d.py:11-92."""
    _check_field_type(field_type)
    ctx = ghost_exchange_communicator.ctx

    if field_type == "scalar":

        def diffusion_timestep_euler_forward_pyst_mpi_kernel_3d(field, diffusion_flux, nu_dt_by_dx2):
            call("smi_diffusion_timestep", ctx.handle, as_ptr(field), as_ptr(diffusion_flux), float(nu_dt_by_dx2), 1)

        fn = diffusion_timestep_euler_forward_pyst_mpi_kernel_3d
    else:

        def vector_field_diffusion_timestep_euler_forward_pyst_mpi_kernel_3d(vector_field, diffusion_flux,
                                                                             nu_dt_by_dx2):
            call("smi_diffusion_timestep", ctx.handle, as_ptr(vector_field), as_ptr(diffusion_flux),
                 float(nu_dt_by_dx2), 3)

        fn = vector_field_diffusion_timestep_euler_forward_pyst_mpi_kernel_3d
    return _support(gen_diffusion_timestep_euler_forward_pyst_mpi_kernel_3d, fn, 1, ghost_exchange_communicator)


# ------------------------------------------------------------------ advection (ENO3)
def gen_advection_flux_conservative_eno3_pyst_mpi_kernel_3d(real_t, mpi_construct, ghost_exchange_communicator):
    """stencil_ops_3d/advection_flux_mpi_3d.py:8-195 (scalar fields; kernel support 2)."""
    ctx = ghost_exchange_communicator.ctx

    def advection_flux_conservative_eno3_pyst_mpi_kernel_3d(advection_flux, field, velocity, inv_dx):
        call("smi_advection_flux_eno3", ctx.handle, as_ptr(advection_flux), as_ptr(field), as_ptr(velocity),
             float(inv_dx))

    return _support(gen_advection_flux_conservative_eno3_pyst_mpi_kernel_3d,
                    advection_flux_conservative_eno3_pyst_mpi_kernel_3d, 2, ghost_exchange_communicator)


def gen_advection_timestep_euler_forward_conservative_eno3_pyst_mpi_kernel_3d(
    real_t, mpi_construct, ghost_exchange_communicator, field_type="scalar"
):
    """stencil_ops_3d/advection_timestep_mpi_3d.py:12-95."""
    _check_field_type(field_type)
    ctx = ghost_exchange_communicator.ctx

    if field_type == "scalar":

        def advection_timestep_euler_forward_conservative_eno3_pyst_mpi_kernel_3d(field, advection_flux, velocity,
                                                                                  dt_by_dx):
            call("smi_advection_timestep_eno3", ctx.handle, as_ptr(field), as_ptr(advection_flux), as_ptr(velocity),
                 float(dt_by_dx), 1)

        fn = advection_timestep_euler_forward_conservative_eno3_pyst_mpi_kernel_3d
    else:

        def vector_field_advection_timestep_euler_forward_conservative_eno3_pyst_mpi_kernel_3d(
            vector_field, advection_flux, velocity, dt_by_dx
        ):
            call("smi_advection_timestep_eno3", ctx.handle, as_ptr(vector_field), as_ptr(advection_flux),
                 as_ptr(velocity), float(dt_by_dx), 3)

        fn = vector_field_advection_timestep_euler_forward_conservative_eno3_pyst_mpi_kernel_3d
    return _support(gen_advection_timestep_euler_forward_conservative_eno3_pyst_mpi_kernel_3d, fn, 2,
                    ghost_exchange_communicator)


# ------------------------------------------------------------------ curl family
def gen_update_vorticity_from_velocity_forcing_pyst_mpi_kernel_3d(real_t, mpi_construct,
                                                                  ghost_exchange_communicator):
    """stencil_ops_3d/update_vorticity_from_velocity_forcing_mpi_3d.py:9-178."""
    ctx = ghost_exchange_communicator.ctx

    def update_vorticity_from_velocity_forcing_pyst_mpi_kernel_3d(vorticity_field, velocity_forcing_field,
                                                                  prefactor):
        call("smi_update_vorticity_from_velocity_forcing", ctx.handle, as_ptr(vorticity_field),
             as_ptr(velocity_forcing_field), float(prefactor))

    return _support(gen_update_vorticity_from_velocity_forcing_pyst_mpi_kernel_3d,
                    update_vorticity_from_velocity_forcing_pyst_mpi_kernel_3d, 1, ghost_exchange_communicator)


def gen_update_vorticity_from_penalised_velocity_pyst_mpi_kernel_3d(real_t, mpi_construct,
                                                                    ghost_exchange_communicator):
    """stencil_ops_3d/update_vorticity_from_velocity_forcing_mpi_3d.py:181-330:
    vorticity_field += prefactor * curl(penalised_velocity_field - velocity_field)."""
    ctx = ghost_exchange_communicator.ctx

    def update_vorticity_from_penalised_velocity_pyst_mpi_kernel_3d(vorticity_field, penalised_velocity_field,
                                                                    velocity_field, prefactor):
        call("smi_update_vorticity_from_penalised_velocity", ctx.handle, as_ptr(vorticity_field),
             as_ptr(penalised_velocity_field), as_ptr(velocity_field), float(prefactor))

    return _support(gen_update_vorticity_from_penalised_velocity_pyst_mpi_kernel_3d,
                    update_vorticity_from_penalised_velocity_pyst_mpi_kernel_3d, 1, ghost_exchange_communicator)


def _gen_brinkmann_penalise(gen, field_type, ncomp_vector):
    _check_field_type(field_type)
    gen.kernel_support = 0

    if field_type == "scalar":

        def brinkmann_penalise(penalised_field, penalty_factor, char_field, penalty_field, field):
            call("smi_brinkmann_penalise", field.ctx.handle, as_ptr(penalised_field), as_ptr(field), as_ptr(char_field),
                 as_ptr(penalty_field), float(penalty_factor), 1)

    else:

        def brinkmann_penalise(penalised_vector_field, penalty_factor, char_field, penalty_vector_field, vector_field):
            call("smi_brinkmann_penalise", vector_field.ctx.handle, as_ptr(penalised_vector_field),
                 as_ptr(vector_field), as_ptr(char_field), as_ptr(penalty_vector_field), float(penalty_factor),
                 ncomp_vector)

    brinkmann_penalise.kernel_support = 0
    return brinkmann_penalise


def gen_brinkmann_penalise_pyst_mpi_kernel_3d(real_t, field_type="scalar"):
    """stencil_ops_3d/brinkmann_penalise_mpi_3d.py:7-19: pointwise (kernel support 0, no exchange),
    penalised = (field + penalty_factor * char_field * penalty_field) / (1 + penalty_factor * char_field)."""
    return _gen_brinkmann_penalise(gen_brinkmann_penalise_pyst_mpi_kernel_3d, field_type, 3)


def _gen_char_func(gen, blend_width):
    gen.kernel_support = 0
    if blend_width <= 0:
        raise ValueError("blend_width must be positive")

    def char_func_from_level_set_via_sine_heaviside(char_func_field, level_set_field):
        call("smi_char_func_from_level_set", level_set_field.ctx.handle, as_ptr(char_func_field),
             as_ptr(level_set_field), float(blend_width))

    char_func_from_level_set_via_sine_heaviside.kernel_support = 0
    return char_func_from_level_set_via_sine_heaviside


def gen_char_func_from_level_set_via_sine_heaviside_pyst_mpi_kernel_3d(blend_width, real_t):
    """stencil_ops_3d/char_func_from_level_set_mpi_3d.py:8-28: characteristic function of a body from its level set
    through the smooth sine Heaviside function (0 below -blend_width, 1 above, pointwise)."""
    return _gen_char_func(gen_char_func_from_level_set_via_sine_heaviside_pyst_mpi_kernel_3d, blend_width)


def gen_curl_pyst_mpi_kernel_3d(real_t, mpi_construct, ghost_exchange_communicator):
    """stencil_ops_3d/curl_mpi_3d.py:10-196."""
    ctx = ghost_exchange_communicator.ctx

    def curl_pyst_mpi_kernel_3d(curl, field, prefactor):
        call("smi_curl", ctx.handle, as_ptr(curl), as_ptr(field), float(prefactor))

    return _support(gen_curl_pyst_mpi_kernel_3d, curl_pyst_mpi_kernel_3d, 1, ghost_exchange_communicator)


def gen_divergence_pyst_mpi_kernel_3d(real_t, mpi_construct, ghost_exchange_communicator):
    """stencil_ops_3d/divergence_mpi_3d.py:10-200."""
    ctx = ghost_exchange_communicator.ctx

    def divergence_pyst_mpi_kernel_3d(divergence, field, inv_dx):
        call("smi_divergence", ctx.handle, as_ptr(divergence), as_ptr(field), float(real_t(inv_dx)))

    return _support(gen_divergence_pyst_mpi_kernel_3d, divergence_pyst_mpi_kernel_3d, 1,
                    ghost_exchange_communicator)


# ------------------------------------------------------------------ penalise boundary
def penalty_factor_tables(width, dx, x_grid_field, y_grid_field, z_grid_field, ghost_size, real_t):
    """Sine factors of penalise_field_boundary_mpi_3d.py:50-57, 78-80, 92-94 for the low and
    high bands of each axis, evaluated on the host in ``real_t`` from the position field."""
    g, w = ghost_size, ghost_size + width
    real_t = np.dtype(real_t).type
    x = np.asarray(x_grid_field[g, g, :], dtype=real_t)
    y = np.asarray(y_grid_field[g, :, g], dtype=real_t)
    z = np.asarray(z_grid_field[:, g, g], dtype=real_t)
    sine_prefactor = real_t((np.pi / 2) / (width * dx))
    lo, hi = [], []
    for c in (x, y, z):
        start, end = c[g], c[-(g + 1)]
        lo.append(np.sin(sine_prefactor * (c[:w] - start)).astype(real_t))
        hi.append(np.sin(sine_prefactor * (end - c[-w:])).astype(real_t))
    return np.concatenate(lo).astype(np.float64), np.concatenate(hi).astype(np.float64)


def gen_penalise_field_boundary_pyst_mpi_kernel_3d(width, dx, x_grid_field, y_grid_field, z_grid_field, real_t,
                                                   mpi_construct, ghost_exchange_communicator, field_type="scalar"):
    """stencil_ops_3d/penalise_field_boundary_mpi_3d.py:9-269."""
    if width < 0 or not isinstance(width, (int, np.integer)):
        raise ValueError("invalid zone width")
    _check_field_type(field_type)
    gen_penalise_field_boundary_pyst_mpi_kernel_3d.kernel_support = 0
    ctx = ghost_exchange_communicator.ctx
    ncomp = 1 if field_type == "scalar" else 3
    if width == 0:
        if field_type == "scalar":

            def penalise_field_boundary_pyst_mpi_kernel_3d(field):
                pass

        else:

            def penalise_field_boundary_pyst_mpi_kernel_3d(vector_field):
                pass

        penalise_field_boundary_pyst_mpi_kernel_3d.kernel_support = 0
        return penalise_field_boundary_pyst_mpi_kernel_3d

    lo, hi = penalty_factor_tables(width, dx, x_grid_field, y_grid_field, z_grid_field,
                                   ghost_exchange_communicator.ghost_size, real_t)
    lo_c, hi_c = darray(lo), darray(hi)

    if field_type == "scalar":

        def penalise_field_boundary_pyst_mpi_kernel_3d(field):
            call("smi_penalise_boundary", ctx.handle, as_ptr(field), ncomp, int(width), lo_c, hi_c)

        fn = penalise_field_boundary_pyst_mpi_kernel_3d
    else:

        def penalise_vector_field_boundary_pyst_mpi_kernel_3d(vector_field):
            call("smi_penalise_boundary", ctx.handle, as_ptr(vector_field), ncomp, int(width), lo_c, hi_c)

        fn = penalise_vector_field_boundary_pyst_mpi_kernel_3d
    fn.kernel_support = 0
    fn.factor_tables = (lo, hi)
    return fn


# ------------------------------------------------------------------ Laplacian filter
def gen_laplacian_filter_mpi_kernel_3d(mpi_construct, ghost_exchange_communicator, filter_order, filter_flux_buffer,
                                       field_buffer, real_t, field_type="scalar", filter_type="multiplicative",
                                       filter_flux_buffer_boundary_width=1):
    """stencil_ops_3d/laplacian_filter_mpi_3d.py:14-421."""
    if filter_order < 0 or not isinstance(filter_order, (int, np.integer)):
        raise ValueError("Invalid filter order")
    if filter_flux_buffer_boundary_width <= 0 or not isinstance(filter_flux_buffer_boundary_width, (int, np.integer)):
        raise ValueError("Invalid value for filter flux buffer boundary zone")
    width = int(filter_flux_buffer_boundary_width)
    types = {"multiplicative": _lib.SMI_FILTER_MULTIPLICATIVE, "convolution": _lib.SMI_FILTER_CONVOLUTION}
    if filter_type not in types:
        raise ValueError("Invalid filter type")
    if field_type not in ("scalar", "vector"):
        raise ValueError("Invalid field type")
    ctx = ghost_exchange_communicator.ctx
    ftype = types[filter_type]

    if field_type == "scalar":

        def scalar_field_filter_mpi_kernel_3d(scalar_field):
            call("smi_laplacian_filter_w", ctx.handle, as_ptr(scalar_field), 1, as_ptr(filter_flux_buffer),
                 as_ptr(field_buffer), int(filter_order), ftype, width)

        fn = scalar_field_filter_mpi_kernel_3d
    else:

        def vector_field_filter_kernel_3d(vector_field):
            call("smi_laplacian_filter_w", ctx.handle, as_ptr(vector_field), 3, as_ptr(filter_flux_buffer),
                 as_ptr(field_buffer), int(filter_order), ftype, width)

        fn = vector_field_filter_kernel_3d
    return _support(gen_laplacian_filter_mpi_kernel_3d, fn, 1, ghost_exchange_communicator)


# ------------------------------------------------------------------ Poisson
class UnboundedPoissonSolverMPI3D:
    """poisson_solver_3d/UnboundedPoissonSolverMPI3D.py:14-187.

    ``backend``: "rocfft" restates the reference literally on one GPU (zero-padded doubled
    domain through rocFFT) and divides the transform over z-slabs on several (any grid size);
    "pruned" uses the library's own zero-padding-aware FFT passes, for the sizes it has plans
    for.  Default: pruned when available for the grid.
    """

    def __init__(self, grid_size_z, grid_size_y, grid_size_x, mpi_construct, ghost_size, x_range=1.0,
                 real_t=np.float64, backend=None):
        self.mpi_construct = mpi_construct
        self.grid_size_z, self.grid_size_y, self.grid_size_x = grid_size_z, grid_size_y, grid_size_x
        self.x_range = x_range
        self.y_range = x_range * (grid_size_y / grid_size_x)
        self.z_range = x_range * (grid_size_z / grid_size_x)
        self.real_t = real_t
        self.dx = real_t(x_range / grid_size_x)
        self.ctx = mpi_construct.device_context(ghost_size)
        # The transforms run on z-slabs.  A pencil decomposition moves the right-hand side to the slab decomposition
        # of the same grid and the solution back (one all-to-all of real data inside each row of the rank grid,
        # smi_pencil_to_slab / smi_slab_to_pencil), the way the reference's solver moves its fields between the
        # grid's and the doubled domain's decomposition (UnboundedPoissonSolverMPI3D.py:190-382).
        self.pencil_ctx = None
        if mpi_construct.size > 1 and (mpi_construct.grid_topology[1] > 1 or mpi_construct.grid_topology[2] > 1):
            self.pencil_ctx = self.ctx
            self.ctx = mpi_construct.slab_context(ghost_size)
            block = int(np.prod(self.pencil_ctx.local_grid_size))
            self._slab_rhs, self._slab_sol = self.ctx.field(3), self.ctx.field(3)
            self._work = (self.pencil_ctx.zeros((3 * block,)), self.pencil_ctx.zeros((3 * block,)))
        names = {"rocfft": _lib.SMI_POISSON_ROCFFT, "pruned": _lib.SMI_POISSON_PRUNED}
        if backend is None:
            backend = "pruned" if self.backend_supports("pruned", grid_size_z, grid_size_y, grid_size_x,
                                                        mpi_construct.size) else "rocfft"
        if backend not in names:
            raise ValueError(f"unknown Poisson back end {backend!r}")
        self.handle = None
        h = C.c_void_p()
        call("smi_poisson_create", C.byref(h), self.ctx.handle, int(grid_size_z), int(grid_size_y),
             int(grid_size_x), float(self.dx), float(x_range), names[backend])
        self.handle, self.backend = h, backend
        if self.pencil_ctx is not None:
            call("smi_poisson_set_pencil_route", h, self.pencil_ctx.handle, as_ptr(self._slab_rhs),
                 as_ptr(self._slab_sol), as_ptr(self._work[0]), as_ptr(self._work[1]))

    @staticmethod
    def backend_supports(backend, grid_size_z, grid_size_y, grid_size_x, nranks=1):
        """Whether ``backend`` ("rocfft" / "pruned") takes this global grid on ``nranks`` z-slabs."""
        code = {"rocfft": _lib.SMI_POISSON_ROCFFT, "pruned": _lib.SMI_POISSON_PRUNED}[backend]
        return bool(_lib.lib.smi_poisson_supports(int(grid_size_z), int(grid_size_y), int(grid_size_x), int(nranks), code))

    def solve(self, solution_field, rhs_field):
        """-del^2(solution_field) = rhs_field, unbounded domain (:133-167)."""
        if self.pencil_ctx is not None:
            return self._solve_through_slabs(solution_field, rhs_field, 1)
        call("smi_poisson_solve", self.handle, as_ptr(solution_field), as_ptr(rhs_field))

    def vector_field_solve(self, solution_vector_field, rhs_vector_field):
        """:169-187."""
        if self.pencil_ctx is not None:
            return self._solve_through_slabs(solution_vector_field, rhs_vector_field, 3)
        call("smi_poisson_vector_solve", self.handle, as_ptr(solution_vector_field), as_ptr(rhs_vector_field))

    def _solve_through_slabs(self, solution, rhs, ncomp):
        """Right-hand side to slabs, solve, solution back to pencils: one library call over the route set at
        construction (events order the two contexts' streams; the host does not wait)."""
        call("smi_poisson_pencil_solve", self.handle, as_ptr(solution), as_ptr(rhs), ncomp)

    def __del__(self):
        try:
            if self.handle is not None and self.ctx.handle:
                _lib.lib.smi_poisson_destroy(self.handle)
        except Exception:
            pass


class _DeviceFFT:
    """Shared part of FFTMPI3D / FFTMPI2D: a hipFFT real-to-complex plan pair on the rank's GPU with
    the reference's buffers and normalisation (forward unnormalised, backward normalised)."""

    def _setup(self, shape, mpi_construct, real_t):
        self.real_dtype = np.dtype(real_t).type
        if self.real_dtype not in (np.float32, np.float64):
            raise ValueError(f"unsupported real type {real_t}")
        self.complex_dtype = np.complex64 if self.real_dtype == np.float32 else np.complex128
        if np.dtype(mpi_construct.real_t) != np.dtype(real_t):
            raise ValueError("real_t differs from the precision of mpi_construct")
        self.global_shape = tuple(int(n) for n in shape)
        nranks = mpi_construct.size
        if nranks > 1:
            # mpi4py-fft's layouts for a slab decomposition (fft_mpi_3d.py:19-33): the real array divided along z,
            # its spectrum along y
            topo = tuple(int(t) for t in mpi_construct.grid_topology)
            if len(self.global_shape) != 3 or topo[1:] != (1, 1):
                raise RuntimeError("across ranks the stand-alone FFT serves z-slabs of a 3D grid")
            nz, ny, nx = self.global_shape
            if nz % nranks or ny % nranks:
                raise RuntimeError(f"{nranks} ranks must divide the z and y extents of the grid {self.global_shape}")
            self.ctx = mpi_construct.device_context(2)   # rank, stream and transport; the plan takes global sizes
            self.shape = (nz // nranks, ny, nx)
            self.fourier_shape = (nz, ny // nranks, nx // 2 + 1)
        else:
            self.ctx = mpi_construct.device_context(2)
            self.shape = self.global_shape
            self.fourier_shape = self.shape[:-1] + (self.shape[-1] // 2 + 1,)
        self.field_buffer = self.ctx.zeros(self.shape, self.real_dtype)
        self.fourier_field_buffer = self.ctx.zeros(self.fourier_shape, self.complex_dtype)
        nz, ny, nx = ((1,) + self.global_shape)[-3:]
        h = C.c_void_p()
        call("smi_fft_create", C.byref(h), self.ctx.handle, nz, ny, nx)
        self.handle = h

    def _check(self, arr, shape, dtype, what):
        if tuple(arr.shape) != shape or np.dtype(arr.dtype) != np.dtype(dtype):
            raise ValueError(f"{what} must be a {np.dtype(dtype).name} array of shape {shape}")

    def forward(self, field, fourier_field):
        """Real-to-complex transform, unnormalised (fft_mpi_3d.py:34-39)."""
        self._check(field, self.shape, self.real_dtype, "field")
        self._check(fourier_field, self.fourier_shape, self.complex_dtype, "fourier_field")
        call("smi_fft_forward", self.handle, as_ptr(field), as_ptr(fourier_field))

    def backward(self, fourier_field, inv_fourier_field):
        """Complex-to-real transform, normalised (fft_mpi_3d.py:41-48); the input is preserved."""
        self._check(fourier_field, self.fourier_shape, self.complex_dtype, "fourier_field")
        self._check(inv_fourier_field, self.shape, self.real_dtype, "inv_fourier_field")
        call("smi_fft_backward", self.handle, as_ptr(fourier_field), as_ptr(inv_fourier_field))

    def __del__(self):
        try:
            if getattr(self, "handle", None) is not None and self.ctx.handle:
                _lib.lib.smi_fft_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class FFTMPI3D(_DeviceFFT):
    """poisson_solver_3d/fft_mpi_3d.py:5-48: real (z, y, x) <-> complex (z, y, x/2 + 1) DeviceFields; across P z-slabs
    real (z / P, y, x) <-> complex (z, y / P, x/2 + 1) per rank, the layouts of mpi4py-fft's slab transform."""

    def __init__(self, grid_size_z, grid_size_y, grid_size_x, mpi_construct, real_t=np.float64):
        self.grid_size_z, self.grid_size_y, self.grid_size_x = grid_size_z, grid_size_y, grid_size_x
        self._setup((grid_size_z, grid_size_y, grid_size_x), mpi_construct, real_t)


class _DomainDoubling:
    def __init__(self, ghost_size, mpi_construct):
        if ghost_size < 0 or not isinstance(ghost_size, (int, np.integer)):
            raise ValueError(f"Ghost size {ghost_size} needs to be an integer >= 0 for field communication.")
        if mpi_construct.size > 1:
            topo = tuple(int(t) for t in mpi_construct.grid_topology)
            if len(topo) != 3 or topo[1:] != (1, 1) or topo[0] % 2:
                raise RuntimeError("across ranks the doubled-domain copies serve an even number of z-slabs of a 3D grid")
        if ghost_size < 1:
            raise NotImplementedError("device contexts carry at least one ghost layer")
        self.mpi_construct = mpi_construct
        self.ghost_size = int(ghost_size)
        self.field_grid_size = np.array(mpi_construct.local_grid_size)
        self.field_doubled_grid_size = self.field_grid_size * 2
        self.ctx = mpi_construct.device_context(self.ghost_size)

    def _check(self, local_field, local_doubled_field):
        g = self.ghost_size
        if tuple(local_field.shape) != tuple(int(n) + 2 * g for n in self.field_grid_size):
            raise ValueError("local_field must be the ghosted local array")
        if tuple(local_doubled_field.shape) != tuple(int(n) for n in self.field_doubled_grid_size):
            raise ValueError("local_doubled_field must have twice the local grid size and no ghosts")

    def copy_to_doubled_domain(self, local_field, local_doubled_field):
        """:314-347: interior of ``local_field`` -> first octant of ``local_doubled_field``."""
        self._check(local_field, local_doubled_field)
        call("smi_copy_to_doubled_domain", self.ctx.handle, as_ptr(local_field), as_ptr(local_doubled_field))

    def copy_from_doubled_domain(self, local_doubled_field, local_field):
        """:349-382."""
        self._check(local_field, local_doubled_field)
        call("smi_copy_from_doubled_domain", self.ctx.handle, as_ptr(local_doubled_field), as_ptr(local_field))


class MPIDomainDoublingCommunicator3D(_DomainDoubling):
    """poisson_solver_3d/UnboundedPoissonSolverMPI3D.py:190-382, on one rank or an even number of z-slabs: every
    rank's interior block goes to rank r // 2 of the same decomposition of the doubled grid, whose
    ``local_doubled_field`` is (2 nz_local, 2 ny, 2 nx); ranks P / 2 and up hold padding only."""


__all__ = [n for n in dir() if n.startswith("gen_")] + [
    "UnboundedPoissonSolverMPI3D", "FFTMPI3D", "MPIDomainDoublingCommunicator3D", "penalty_factor_tables"]
from .ops_2d import *  # noqa: F401,F403,E402
