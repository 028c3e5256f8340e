"""Oracle (test infrastructure): emulate the reference's z-slab decomposition in one process.

Mirrors MPIConstruct3D (utils/mpi_utils_3d.py:32-83, slab mode ``rank_distribution=(0,1,1)``),
the face halo exchange MPIGhostCommunicator3D.exchange_scalar_field_faces_init
(:756-842: only the interior extent of each face travels, no edges / corners), the
scatter / gather of MPIFieldCommunicator3D (:1192-1306: interiors only) and the ghost sum
of MPIGhostSumCommunicator3D (EulerianLagrangianGridCommunicatorMPI3D.py:677-792).
"""
import numpy as np


class ZSlabs:
    def __init__(self, grid_size, num_ranks, ghost_size):
        nz, ny, nx = grid_size
        if nz % num_ranks:
            raise RuntimeError("Invalid domain decomposition")
        self.P = num_ranks
        self.g = ghost_size
        self.global_size = (nz, ny, nx)
        self.local_size = (nz // num_ranks, ny, nx)

    def phys(self, rank):
        """(z_prev, z_next, y_prev, y_next, x_prev, x_next) physical-boundary flags."""
        return (rank == 0, rank == self.P - 1, True, True, True, True)

    def local_shape(self):
        g = self.g
        return tuple(n + 2 * g for n in self.local_size)

    def scatter(self, global_interior):
        """Global interior array (nz,ny,nx) or (c,nz,ny,nx) -> list of ghosted local arrays;
        ghosts are zero (the reference scatters interiors only)."""
        g, lz = self.g, self.local_size[0]
        lead = global_interior.shape[:-3]
        out = []
        for r in range(self.P):
            loc = np.zeros(lead + self.local_shape(), dtype=global_interior.dtype)
            loc[..., g:-g, g:-g, g:-g] = global_interior[..., r * lz : (r + 1) * lz, :, :]
            out.append(loc)
        return out

    def gather(self, locals_):
        g = self.g
        return np.concatenate([loc[..., g:-g, g:-g, g:-g] for loc in locals_], axis=-3)

    def exchange_faces(self, locals_):
        """z-face halo exchange of the interior (y, x) extent, every rank at once."""
        g = self.g
        for r in range(self.P - 1):
            lo, hi = locals_[r], locals_[r + 1]
            # r's last interior planes -> r+1's low ghosts; r+1's first interior planes -> r's high ghosts
            hi[..., :g, g:-g, g:-g] = lo[..., -2 * g : -g, g:-g, g:-g]
            lo[..., -g:, g:-g, g:-g] = hi[..., g : 2 * g, g:-g, g:-g]

    def ghost_sum(self, locals_):
        """Scalar ghost sum across z-slabs (y / x neighbours are PROC_NULL), then clear ghosts.

        Order as the reference: first every rank sends its high z ghost planes (full padded
        planes) to the next rank, which adds them to its first interior planes; then the low
        ghost planes go to the previous rank."""
        g = self.g
        for r in range(self.P - 1):
            locals_[r + 1][g : 2 * g, :, :] += locals_[r][-g:, :, :]
        for r in range(1, self.P):
            locals_[r - 1][-2 * g : -g, :, :] += locals_[r][:g, :, :]
        for loc in locals_:
            loc[:g, :, :] = 0
            loc[-g:, :, :] = 0
            loc[:, :g, :] = 0
            loc[:, -g:, :] = 0
            loc[:, :, :g] = 0
            loc[:, :, -g:] = 0

    def lagrangian_owner(self, lag_positions, dx, eul_grid_coord_shift):
        """utils/mpi_utils_3d.py:1352-1384 for slabs: owner = int((Z - shift) / (dx * n_local_z))."""
        sub = dx * self.local_size[0]
        coords_z = ((lag_positions[2] - eul_grid_coord_shift) / sub).astype(np.int32)
        if np.any(coords_z >= self.P):
            raise RuntimeError("Lagrangian node is found outside of Eulerian domain!")
        return coords_z
