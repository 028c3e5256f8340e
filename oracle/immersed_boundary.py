"""Oracle (test infrastructure): immersed-boundary Eulerian <-> Lagrangian operators.

Restates EulerianLagrangianGridCommunicatorMPI3D.py (support/index :116-178, cosine
weights :443-479, Peskin weights :482-589, E->L :181-326, L->E :329-427, ghost sum
:677-792), its 2D twin, and VirtualBoundaryForcingMPI.py:278-459, for ONE rank.
Array layouts are the reference's: Lagrangian arrays ``(dim, M)`` in the Lagrangian
dtype (float64), index ``(dim, M)`` int64 ordered x, y, z, support
``(dim, 4, 4, 4, M)`` with window axes ``[z][y][x]``, weights ``(4, 4, 4, M)``.
"""
import numpy as np


def local_eulerian_grid_support_of_lagrangian_grid(
    support, nearest_index, lag_positions, dx, eul_grid_coord_shift, interp_kernel_width,
    local_substart_coord_shift,
):
    """EulerianLagrangianGridCommunicatorMPI3D.py:136-176 (any dim).

    ``local_substart_coord_shift`` = (substart index in x, y[, z] order) - ghost_size.
    """
    dim = lag_positions.shape[0]
    w = interp_kernel_width
    shift = np.asarray(local_substart_coord_shift, dtype=np.int64).reshape(dim, 1)
    nearest_index[...] = (lag_positions - eul_grid_coord_shift) // dx - shift
    x = np.arange(-w + 1, w + 1)
    grids = np.meshgrid(*([x] * dim), indexing="ij")  # axes ordered (z, y, x) in 3D, (y, x) in 2D
    support_idx = np.stack(grids[::-1])  # component 0 = x offsets, last = z (or y) offsets
    win = (2 * w,) * dim
    support[...] = (
        nearest_index.reshape((dim,) + (1,) * dim + (-1,))
        + support_idx.reshape((dim,) + win + (1,))
        + shift.reshape((dim,) + (1,) * dim + (-1,))
    ) * dx + eul_grid_coord_shift - lag_positions.reshape((dim,) + (1,) * dim + (-1,))


def cosine_interpolation_weights(weights, support, dx, real_t):
    """:457-477.  Mutates ``support`` (divides by dx) like the reference."""
    dim = support.shape[0]
    support /= dx
    # numba types the closure constant as float64 arithmetic rounded once by real_t(...)
    acc = float(real_t((0.25 / float(dx)) ** dim))
    for a in range(dim):
        acc = acc * (1.0 + np.cos(float(real_t(0.5 * np.pi)) * support[a]))
    weights[...] = acc


def peskin_interpolation_weights(weights, support, dx, real_t):
    """:496-587 (Peskin 2002, eq. 6.27).  Mutates ``support`` (|.|/dx)."""
    dim = support.shape[0]
    support[...] = np.fabs(support) / dx
    acc = (0.125 / float(dx)) ** dim
    for a in range(dim):
        r = support[a]
        acc = acc * (
            (r < 1.0) * (3.0 - 2 * r + np.sqrt(np.fabs(1 + 4 * r - 4 * r**2)))
            + (r >= 1.0) * (r < 2.0) * (5.0 - 2 * r - np.sqrt(np.fabs(-7 + 12 * r - 4 * r**2)))
        )
    weights[...] = np.asarray(acc).astype(real_t)


def _window(nearest_index, i, w):
    """Slices [idx - w + 1 : idx + w + 1] per axis, outermost axis first (z, y, x)."""
    dim = nearest_index.shape[0]
    return tuple(
        slice(int(nearest_index[a, i]) - w + 1, int(nearest_index[a, i]) + w + 1)
        for a in range(dim - 1, -1, -1)
    )


def eulerian_to_lagrangian_grid_interpolation(lag_field, eul_field, weights, nearest_index, dx, w=2):
    """:203-324.  ``eul_field`` (z,y,x) with ``lag_field`` (M,), or (dim,z,y,x) with (dim,M)."""
    dim = nearest_index.shape[0]
    m = nearest_index.shape[1]
    dx_pow = float(dx) ** dim
    for i in range(m):
        win = _window(nearest_index, i, w)
        if lag_field.ndim == 1:
            lag_field[i] = np.sum(eul_field[win] * weights[..., i]) * dx_pow
        else:
            for c in range(lag_field.shape[0]):
                lag_field[c, i] = np.sum(eul_field[(c,) + win] * weights[..., i]) * dx_pow


def lagrangian_to_eulerian_grid_interpolation(eul_field, lag_field, weights, nearest_index, w=2):
    """:349-425, serial over nodes, so overlapping windows accumulate in node order."""
    m = nearest_index.shape[1]
    for i in range(m):
        win = _window(nearest_index, i, w)
        if lag_field.ndim == 1:
            eul_field[win] += lag_field[i] * weights[..., i]
        else:
            for c in range(lag_field.shape[0]):
                eul_field[(c,) + win] += lag_field[c, i] * weights[..., i]


def clear_ghost_cells(field, g, grid_dim=3):
    """:786-792 (2D: EulerianLagrangianGridCommunicatorMPI2D.py:646-652) for the last ``grid_dim``
    axes of ``field``; a leading component axis is left alone."""
    field[..., :g] = 0
    field[..., -g:] = 0
    field[..., :g, :] = 0
    field[..., -g:, :] = 0
    if grid_dim == 3:
        field[..., :g, :, :] = 0
        field[..., -g:, :, :] = 0


def ghost_sum_single_rank_3d(field, g):
    """:677-784 on one rank: every neighbour is PROC_NULL, nothing is added, ghosts are zeroed."""
    field[:g, :, :] = 0
    field[-g:, :, :] = 0
    field[:, :g, :] = 0
    field[:, -g:, :] = 0
    field[:, :, :g] = 0
    field[:, :, -g:] = 0


class VirtualBoundaryForcing:
    """VirtualBoundaryForcingMPI.py on one rank (``assume_data_locality`` semantics: every
    Lagrangian node is local).  Eulerian arrays carry ``ghost_size`` ghost layers."""

    def __init__(self, virtual_boundary_stiffness_coeff, virtual_boundary_damping_coeff, grid_dim, dx,
                 num_lag_nodes, ghost_size, eul_real_t=np.float32, lag_real_t=np.float64,
                 eul_grid_coord_shift=None, interp_kernel_width=2, interp_kernel_type="cosine",
                 substart_idx=None, start_time=0.0):
        self.grid_dim = grid_dim
        self.dx = dx
        self.k = virtual_boundary_stiffness_coeff
        self.c = virtual_boundary_damping_coeff
        self.w = interp_kernel_width
        self.g = ghost_size
        self.eul_real_t = eul_real_t
        self.kernel_type = interp_kernel_type
        self.time = start_time
        self.shift = eul_real_t(dx / 2) if eul_grid_coord_shift is None else eul_grid_coord_shift
        sub = np.zeros(grid_dim, dtype=np.int64) if substart_idx is None else np.asarray(substart_idx)
        self.local_shift = sub - ghost_size  # x, y, z order
        m = num_lag_nodes
        win = (2 * self.w,) * grid_dim
        self.nearest_index = np.empty((grid_dim, m), dtype=np.int64)
        self.support = np.empty((grid_dim,) + win + (m,), dtype=lag_real_t)
        self.weights = np.empty(win + (m,), dtype=lag_real_t)
        self.lag_flow_velocity = np.zeros((grid_dim, m), dtype=lag_real_t)
        self.lag_position_mismatch = np.zeros_like(self.lag_flow_velocity)
        self.lag_velocity_mismatch = np.zeros_like(self.lag_flow_velocity)
        self.lag_forcing = np.zeros_like(self.lag_flow_velocity)

    def compute_interaction_force_on_lag_grid(self, eul_velocity, lag_positions, lag_velocities):
        """VirtualBoundaryForcingMPI.py:333-406."""
        local_eulerian_grid_support_of_lagrangian_grid(
            self.support, self.nearest_index, lag_positions, self.dx, self.shift, self.w, self.local_shift
        )
        if self.kernel_type == "cosine":
            cosine_interpolation_weights(self.weights, self.support, self.dx, self.eul_real_t)
        else:
            peskin_interpolation_weights(self.weights, self.support, self.dx, self.eul_real_t)
        eulerian_to_lagrangian_grid_interpolation(
            self.lag_flow_velocity, eul_velocity, self.weights, self.nearest_index, self.dx, self.w
        )
        self.lag_velocity_mismatch[...] = self.lag_flow_velocity - lag_velocities
        self.lag_forcing[...] = self.k * self.lag_position_mismatch + self.c * self.lag_velocity_mismatch

    def compute_interaction_force_on_eul_and_lag_grid(self, eul_forcing, eul_velocity, lag_positions,
                                                      lag_velocities):
        """:408-429 followed by the single-rank ghost sum (communicator :99-113)."""
        self.compute_interaction_force_on_lag_grid(eul_velocity, lag_positions, lag_velocities)
        lagrangian_to_eulerian_grid_interpolation(
            eul_forcing, self.lag_forcing, self.weights, self.nearest_index, self.w
        )
        clear_ghost_cells(eul_forcing, self.g, self.grid_dim)

    def time_step(self, dt):
        """:452-459."""
        self.lag_position_mismatch[...] = self.lag_position_mismatch + dt * self.lag_velocity_mismatch
        self.time += dt
