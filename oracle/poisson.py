"""Oracle (test infrastructure): unbounded Poisson solve, 3D and 2D.

Restates UnboundedPoissonSolverMPI3D.py:22-187 / UnboundedPoissonSolverMPI2D.py:20-153 on
one rank: Hockney-Eastwood domain doubling, Green's function convolution through an
unnormalised forward / normalised backward real FFT (fft_mpi_3d.py:34-48; the
reference's own test pins that convention to scipy.fft.rfftn,
tests/test_numeric/test_eulerian_grid_ops/test_poisson_solver_3d/test_fft_mpi_3d.py:105-115).
"""
import os

import numpy as np
import scipy.fft as spfft

_WORKERS = int(os.environ.get("ORACLE_FFT_WORKERS", os.cpu_count() or 1))


class UnboundedPoissonSolver3D:
    def __init__(self, grid_size_z, grid_size_y, grid_size_x, x_range=1.0, real_t=np.float64):
        self.nz, self.ny, self.nx = grid_size_z, grid_size_y, grid_size_x
        self.real_t = real_t
        self.x_range = x_range
        self.y_range = x_range * (grid_size_y / grid_size_x)
        self.z_range = x_range * (grid_size_z / grid_size_x)
        self.dx = real_t(x_range / grid_size_x)
        self.doubled = np.zeros((2 * self.nz, 2 * self.ny, 2 * self.nx), dtype=real_t)
        self._construct_fourier_greens_function_field()

    def greens_function_field(self):
        """UnboundedPoissonSolverMPI3D.py:82-114 (single rank: substart = 0)."""
        real_t, dx = self.real_t, self.dx
        n2z, n2y, n2x = self.doubled.shape
        lx = np.linspace(0 * dx, (n2x - 1) * dx, n2x).astype(real_t)
        ly = np.linspace(0 * dx, (n2y - 1) * dx, n2y).astype(real_t)
        lz = np.linspace(0 * dx, (n2z - 1) * dx, n2z).astype(real_t)
        zg, yg, xg = np.meshgrid(lz, ly, lx, indexing="ij")
        dist = np.sqrt(
            np.minimum(xg, 2 * self.x_range - xg) ** 2
            + np.minimum(yg, 2 * self.y_range - yg) ** 2
            + np.minimum(zg, 2 * self.z_range - zg) ** 2
        )
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            green = (1 / dist) / (4 * np.pi)
        green[0, 0, 0] = 1 / (4 * np.pi * dx)
        return green.astype(real_t, copy=False)

    def _construct_fourier_greens_function_field(self):
        green_hat = spfft.rfftn(self.greens_function_field(), workers=_WORKERS)
        self.fourier_greens_function_times_dx_cubed = green_hat * (self.dx**3)

    def solve(self, solution_field, rhs_field, g):
        """UnboundedPoissonSolverMPI3D.py:133-167. Fields carry ``g`` ghost layers; only
        their interiors are read / written."""
        nz, ny, nx = self.nz, self.ny, self.nx
        inner = (slice(g, g + nz), slice(g, g + ny), slice(g, g + nx))
        self.doubled[...] = 0
        self.doubled[:nz, :ny, :nx] = rhs_field[inner]
        fourier = spfft.rfftn(self.doubled, workers=_WORKERS)
        conv = fourier * self.fourier_greens_function_times_dx_cubed
        self.doubled[...] = spfft.irfftn(conv, s=self.doubled.shape, workers=_WORKERS)
        solution_field[inner] = self.doubled[:nz, :ny, :nx]

    def vector_field_solve(self, solution_vector_field, rhs_vector_field, g):
        for c in range(3):
            self.solve(solution_vector_field[c], rhs_vector_field[c], g)


class UnboundedPoissonSolver2D:
    def __init__(self, grid_size_y, grid_size_x, x_range=1.0, real_t=np.float64):
        self.ny, self.nx = grid_size_y, grid_size_x
        self.real_t = real_t
        self.x_range = x_range
        self.y_range = x_range * (grid_size_y / grid_size_x)
        self.dx = real_t(x_range / grid_size_x)
        self.doubled = np.zeros((2 * self.ny, 2 * self.nx), dtype=real_t)
        green_hat = spfft.rfftn(self.greens_function_field(), workers=_WORKERS)
        self.fourier_greens_function_times_dx_squared = green_hat * (self.dx**2)

    def greens_function_field(self):
        """UnboundedPoissonSolverMPI2D.py:61-107: G = -ln(r) / (2 pi), regularised at r = 0."""
        real_t, dx = self.real_t, self.dx
        n2y, n2x = self.doubled.shape
        lx = np.linspace(0 * dx, (n2x - 1) * dx, n2x).astype(real_t)
        ly = np.linspace(0 * dx, (n2y - 1) * dx, n2y).astype(real_t)
        xg, yg = np.meshgrid(lx, ly)
        dist = np.sqrt(
            np.minimum(xg, 2 * self.x_range - xg) ** 2
            + np.minimum(yg, 2 * self.y_range - yg) ** 2
        )
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            green = -np.log(dist) / (2 * np.pi)
        green[0, 0] = -(2 * np.log(dx / np.sqrt(np.pi)) - 1) / (4 * np.pi)
        return green.astype(real_t, copy=False)

    def solve(self, solution_field, rhs_field, g):
        ny, nx = self.ny, self.nx
        inner = (slice(g, g + ny), slice(g, g + nx))
        self.doubled[...] = 0
        self.doubled[:ny, :nx] = rhs_field[inner]
        fourier = spfft.rfftn(self.doubled, workers=_WORKERS)
        conv = fourier * self.fourier_greens_function_times_dx_squared
        self.doubled[...] = spfft.irfftn(conv, s=self.doubled.shape, workers=_WORKERS)
        solution_field[inner] = self.doubled[:ny, :nx]
