"""2D Eulerian grid operators with the reference's generator API
(sopht_mpi/numeric/eulerian_grid_ops/stencil_ops_2d/, poisson_solver_2d/)."""
import ctypes as C

import numpy as np

from ..._lib import call, darray
from ...device import as_ptr
from ...utils import check_valid_ghost_size_and_kernel_support


def _support(gen, fn, kernel_support, ghost_exchange_communicator):
    gen.kernel_support = kernel_support
    fn.kernel_support = kernel_support
    check_valid_ghost_size_and_kernel_support(ghost_exchange_communicator.ghost_size, kernel_support)
    return fn


def gen_brinkmann_penalise_pyst_mpi_kernel_2d(real_t, field_type="scalar"):
    """stencil_ops_2d/brinkmann_penalise_mpi_2d.py:7-19 (pointwise, kernel support 0)."""
    from . import _gen_brinkmann_penalise

    return _gen_brinkmann_penalise(gen_brinkmann_penalise_pyst_mpi_kernel_2d, field_type, 2)


def gen_char_func_from_level_set_via_sine_heaviside_pyst_mpi_kernel_2d(blend_width, real_t):
    """stencil_ops_2d/char_func_from_level_set_mpi_2d.py:8-28 (pointwise, kernel support 0)."""
    from . import _gen_char_func

    return _gen_char_func(gen_char_func_from_level_set_via_sine_heaviside_pyst_mpi_kernel_2d, blend_width)


def gen_diffusion_flux_pyst_mpi_kernel_2d(real_t, mpi_construct, ghost_exchange_communicator):
    """stencil_ops_2d/diffusion_flux_mpi_2d.py."""
    ctx = ghost_exchange_communicator.ctx

    def diffusion_flux_pyst_mpi_kernel_2d(diffusion_flux, field, prefactor):
        call("smi2d_diffusion_flux", ctx.handle, as_ptr(diffusion_flux), as_ptr(field), float(prefactor))

    return _support(gen_diffusion_flux_pyst_mpi_kernel_2d, diffusion_flux_pyst_mpi_kernel_2d, 1,
                    ghost_exchange_communicator)


def gen_diffusion_timestep_euler_forward_pyst_mpi_kernel_2d(real_t, mpi_construct, ghost_exchange_communicator):
    """stencil_ops_2d/diffusion_timestep_mpi_2d.py."""
    ctx = ghost_exchange_communicator.ctx

    def diffusion_timestep_euler_forward_pyst_mpi_kernel_2d(field, diffusion_flux, nu_dt_by_dx2):
        call("smi2d_diffusion_timestep", ctx.handle, as_ptr(field), as_ptr(diffusion_flux), float(nu_dt_by_dx2))

    return _support(gen_diffusion_timestep_euler_forward_pyst_mpi_kernel_2d,
                    diffusion_timestep_euler_forward_pyst_mpi_kernel_2d, 1, ghost_exchange_communicator)


def gen_advection_flux_conservative_eno3_pyst_mpi_kernel_2d(real_t, mpi_construct, ghost_exchange_communicator):
    """stencil_ops_2d/advection_flux_mpi_2d.py (kernel support 2)."""
    ctx = ghost_exchange_communicator.ctx

    def advection_flux_conservative_eno3_pyst_mpi_kernel_2d(advection_flux, field, velocity, inv_dx):
        call("smi2d_advection_flux_eno3", ctx.handle, as_ptr(advection_flux), as_ptr(field), as_ptr(velocity),
             float(inv_dx))

    return _support(gen_advection_flux_conservative_eno3_pyst_mpi_kernel_2d,
                    advection_flux_conservative_eno3_pyst_mpi_kernel_2d, 2, ghost_exchange_communicator)


def gen_advection_timestep_euler_forward_conservative_eno3_pyst_mpi_kernel_2d(real_t, mpi_construct,
                                                                              ghost_exchange_communicator):
    """stencil_ops_2d/advection_timestep_mpi_2d.py (kernel support 2)."""
    ctx = ghost_exchange_communicator.ctx

    def advection_timestep_euler_forward_conservative_eno3_pyst_mpi_kernel_2d(field, advection_flux, velocity,
                                                                              dt_by_dx):
        call("smi2d_advection_timestep_eno3", ctx.handle, as_ptr(field), as_ptr(advection_flux), as_ptr(velocity),
             float(dt_by_dx))

    return _support(gen_advection_timestep_euler_forward_conservative_eno3_pyst_mpi_kernel_2d,
                    advection_timestep_euler_forward_conservative_eno3_pyst_mpi_kernel_2d, 2,
                    ghost_exchange_communicator)


def gen_outplane_field_curl_pyst_mpi_kernel_2d(real_t, mpi_construct, ghost_exchange_communicator):
    """stencil_ops_2d/outplane_field_curl_mpi_2d.py:10-141."""
    ctx = ghost_exchange_communicator.ctx

    def outplane_field_curl_pyst_mpi_kernel_2d(curl, field, prefactor):
        call("smi2d_outplane_field_curl", ctx.handle, as_ptr(curl), as_ptr(field), float(prefactor))

    return _support(gen_outplane_field_curl_pyst_mpi_kernel_2d, outplane_field_curl_pyst_mpi_kernel_2d, 1,
                    ghost_exchange_communicator)


def gen_update_vorticity_from_velocity_forcing_pyst_mpi_kernel_2d(real_t, mpi_construct,
                                                                  ghost_exchange_communicator):
    """stencil_ops_2d/update_vorticity_from_velocity_forcing_mpi_2d.py."""
    ctx = ghost_exchange_communicator.ctx

    def update_vorticity_from_velocity_forcing_pyst_mpi_kernel_2d(vorticity_field, velocity_forcing_field, prefactor):
        call("smi2d_update_vorticity_from_velocity_forcing", ctx.handle, as_ptr(vorticity_field),
             as_ptr(velocity_forcing_field), float(prefactor))

    return _support(gen_update_vorticity_from_velocity_forcing_pyst_mpi_kernel_2d,
                    update_vorticity_from_velocity_forcing_pyst_mpi_kernel_2d, 1, ghost_exchange_communicator)


def gen_penalise_field_boundary_pyst_mpi_kernel_2d(width, dx, x_grid_field, y_grid_field, real_t, mpi_construct,
                                                   ghost_exchange_communicator):
    """stencil_ops_2d/penalise_field_boundary_mpi_2d.py:9-172: sine factors evaluated on the host
    in ``real_t`` from the position field (:39-44, 64-66, 78-80)."""
    if width < 0 or not isinstance(width, (int, np.integer)):
        raise ValueError("invalid zone width")
    gen_penalise_field_boundary_pyst_mpi_kernel_2d.kernel_support = 0
    ctx = ghost_exchange_communicator.ctx
    if width == 0:

        def penalise_field_boundary_pyst_mpi_kernel_2d(field):
            pass

        penalise_field_boundary_pyst_mpi_kernel_2d.kernel_support = 0
        return penalise_field_boundary_pyst_mpi_kernel_2d
    g = ghost_exchange_communicator.ghost_size
    w = g + width
    t = np.dtype(real_t).type
    x = np.asarray(x_grid_field[g, :], dtype=t)
    y = np.asarray(y_grid_field[:, g], dtype=t)
    sine_prefactor = t((np.pi / 2) / (width * dx))
    lo, hi = [], []
    for c in (x, y):
        lo.append(np.sin(sine_prefactor * (c[:w] - c[g])).astype(t))
        hi.append(np.sin(sine_prefactor * (c[-(g + 1)] - c[-w:])).astype(t))
    lo_c, hi_c = darray(np.concatenate(lo).astype(np.float64)), darray(np.concatenate(hi).astype(np.float64))

    def penalise_field_boundary_pyst_mpi_kernel_2d(field):
        call("smi2d_penalise_boundary", ctx.handle, as_ptr(field), int(width), lo_c, hi_c)

    penalise_field_boundary_pyst_mpi_kernel_2d.kernel_support = 0
    penalise_field_boundary_pyst_mpi_kernel_2d.factor_tables = (lo_c, hi_c)  # for the fused step
    return penalise_field_boundary_pyst_mpi_kernel_2d


def gen_set_fixed_val_pyst_kernel_2d(real_t, field_type="scalar", ctx=None):
    """sopht ``gen_set_fixed_val_pyst_kernel_2d`` (call site flow_simulators_mpi_2d.py:217-220)."""
    if field_type not in ("scalar", "vector"):
        raise ValueError("Invalid field type")

    if field_type == "scalar":

        def set_fixed_val_pyst_kernel_2d(field, fixed_val):
            call("smi2d_set_fixed_val", field.ctx.handle, as_ptr(field), 1, darray([fixed_val]))

        return set_fixed_val_pyst_kernel_2d

    def vector_field_set_fixed_val_pyst_kernel_2d(vector_field, fixed_vals):
        call("smi2d_set_fixed_val", vector_field.ctx.handle, as_ptr(vector_field), 2, darray(fixed_vals))

    return vector_field_set_fixed_val_pyst_kernel_2d


def gen_add_fixed_val_pyst_kernel_2d(real_t, field_type="scalar", ctx=None):
    """sopht ``gen_add_fixed_val_pyst_kernel_2d`` (call site flow_simulators_mpi_2d.py:222-233)."""
    if field_type not in ("scalar", "vector"):
        raise ValueError("Invalid field type")

    if field_type == "scalar":

        def add_fixed_val_pyst_kernel_2d(sum_field, field, fixed_val):
            call("smi2d_add_fixed_val", field.ctx.handle, as_ptr(sum_field), as_ptr(field), 1, darray([fixed_val]))

        return add_fixed_val_pyst_kernel_2d

    def vector_field_add_fixed_val_pyst_kernel_2d(sum_field, vector_field, fixed_vals):
        call("smi2d_add_fixed_val", vector_field.ctx.handle, as_ptr(sum_field), as_ptr(vector_field), 2,
             darray(fixed_vals))

    return vector_field_add_fixed_val_pyst_kernel_2d


class UnboundedPoissonSolverMPI2D:
    """poisson_solver_2d/UnboundedPoissonSolverMPI2D.py:20-153 (hipFFT 2D transforms, fused pad / multiply / crop
    kernels).  Across row slabs every rank gathers the right-hand side of the whole plane and solves it: a 2D plane is
    a few MB, one all-gather of it costs less than the two transposes of a divided transform."""

    def __init__(self, grid_size_y, grid_size_x, mpi_construct, ghost_size, x_range=1.0, real_t=np.float64):
        self.mpi_construct = mpi_construct
        self.grid_size_y, self.grid_size_x = grid_size_y, grid_size_x
        self.x_range = x_range
        self.y_range = x_range * (grid_size_y / grid_size_x)
        self.real_t = np.dtype(real_t).type
        self.dx = self.real_t(x_range / grid_size_x)
        self.ghost_size = ghost_size
        self.ctx = mpi_construct.device_context(ghost_size)
        green = self.construct_greens_function_field()
        h = C.c_void_p()
        call("smi2d_poisson_create", C.byref(h), self.ctx.handle, green.ctypes.data, float(self.dx))
        self.handle = h

    def construct_greens_function_field(self):
        """:61-107: G = -ln(r) / (2 pi) on the doubled grid with mirrored distances, regularised at 0."""
        t, dx = self.real_t, self.dx
        n2y, n2x = 2 * self.grid_size_y, 2 * self.grid_size_x
        lx = np.linspace(0 * dx, (n2x - 1) * dx, n2x).astype(t)
        ly = np.linspace(0 * dx, (n2y - 1) * dx, n2y).astype(t)
        xg, yg = np.meshgrid(lx, ly)
        dist = np.sqrt(np.minimum(xg, 2 * self.x_range - xg) ** 2 + np.minimum(yg, 2 * self.y_range - yg) ** 2)
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            green = -np.log(dist) / (2 * np.pi)
        green[0, 0] = -(2 * np.log(dx / np.sqrt(np.pi)) - 1) / (4 * np.pi)
        return np.ascontiguousarray(green.astype(t, copy=False))

    def solve(self, solution_field, rhs_field):
        call("smi2d_poisson_solve", self.handle, as_ptr(solution_field), as_ptr(rhs_field))

    def __del__(self):
        try:
            from ..._lib import lib

            if getattr(self, "handle", None) and self.ctx.handle:
                lib.smi2d_poisson_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def _late():
    from . import _DeviceFFT, _DomainDoubling

    class FFTMPI2D(_DeviceFFT):
        """poisson_solver_2d/fft_mpi_2d.py:5-45: real (y, x) <-> complex (y, x/2 + 1) DeviceFields."""

        def __init__(self, grid_size_y, grid_size_x, mpi_construct, real_t=np.float64):
            self.grid_size_y, self.grid_size_x = grid_size_y, grid_size_x
            self._setup((grid_size_y, grid_size_x), mpi_construct, real_t)

    class MPIDomainDoublingCommunicator2D(_DomainDoubling):
        """poisson_solver_2d/UnboundedPoissonSolverMPI2D.py:156-352 on one rank."""

    return FFTMPI2D, MPIDomainDoublingCommunicator2D


FFTMPI2D, MPIDomainDoublingCommunicator2D = _late()
