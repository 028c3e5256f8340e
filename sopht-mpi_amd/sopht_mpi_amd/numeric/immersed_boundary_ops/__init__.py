"""Immersed-boundary operators with the reference's names
(sopht_mpi/numeric/immersed_boundary_ops/__init__.py:1-8)."""
import numpy as np

from ... import _lib
from ..._lib import call, i64array
from ...device import EMPTY_SUPPORT, DeviceField, as_ptr
from ...utils import MPILagrangianFieldCommunicator2D, MPILagrangianFieldCommunicator3D

_KERNEL_TYPES = {"cosine": _lib.SMI_DELTA_COSINE, "peskin": _lib.SMI_DELTA_PESKIN}


class EulerianLagrangianGridCommunicatorMPI3D:
    """EulerianLagrangianGridCommunicatorMPI3D.py:7-113.  The four kernels take device arrays
    positionally, in the reference's layouts: index (3, M) int64 (x, y, z), support
    (3, 4, 4, 4, M) float64, weights (4, 4, 4, M) float64, Lagrangian fields (3, M) or (M,)."""

    def __init__(self, dx, eul_grid_coord_shift, interp_kernel_width, real_t, mpi_construct, ghost_size,
                 n_components=1, interp_kernel_type="cosine"):
        if ghost_size < interp_kernel_width:
            raise ValueError(
                f"ghost size ({ghost_size}) needs to be >= interp kernel width ({interp_kernel_width})"
            )
        if interp_kernel_width != 2:
            raise AssertionError("Interpolation kernel inconsistent with interpolation kernel width!")
        if n_components not in (1, 3):
            raise AssertionError("invalid number of components for interpolation!")
        if interp_kernel_type not in _KERNEL_TYPES:
            raise ValueError(
                "Invalid interpolation kernel type. Currently supported types are 'cosine' and 'peskin'."
            )
        self.mpi_construct = mpi_construct
        self.ctx = mpi_construct.device_context(ghost_size)
        self.dx = float(dx)
        self.coord_shift = float(eul_grid_coord_shift)
        self.n_components = n_components
        self.kernel_type = _KERNEL_TYPES[interp_kernel_type]
        # :41-46, x, y, z order
        self.mpi_substart_idx = np.flip(mpi_construct.grid.coords * mpi_construct.local_grid_size)
        self.mpi_local_substart_coord_shift = self.mpi_substart_idx - ghost_size
        self._substart = i64array(self.mpi_substart_idx)

    def local_eulerian_grid_support_of_lagrangian_grid_kernel(self, local_eul_grid_support_of_lag_grid,
                                                              nearest_eul_grid_index_to_lag_grid, lag_positions):
        m = lag_positions.shape[-1]
        call("smi_ib_support", self.ctx.handle, as_ptr(local_eul_grid_support_of_lag_grid),
             as_ptr(nearest_eul_grid_index_to_lag_grid), as_ptr(lag_positions), m, self.dx, self.coord_shift,
             self._substart)

    def interpolation_weights_kernel(self, interp_weights, local_eul_grid_support_of_lag_grid):
        m = interp_weights.shape[-1]
        call("smi_ib_weights", self.ctx.handle, as_ptr(interp_weights), as_ptr(local_eul_grid_support_of_lag_grid),
             m, self.dx, self.kernel_type)

    def eulerian_to_lagrangian_grid_interpolation_kernel(self, lag_grid_field, eul_grid_field, interp_weights,
                                                         nearest_eul_grid_index_to_lag_grid):
        m = lag_grid_field.shape[-1]
        call("smi_ib_interpolate", self.ctx.handle, as_ptr(lag_grid_field), as_ptr(eul_grid_field),
             as_ptr(interp_weights), as_ptr(nearest_eul_grid_index_to_lag_grid), m, self.n_components, self.dx)

    def lagrangian_to_eulerian_grid_interpolation_kernel_without_ghost_sum(self, eul_grid_field, lag_grid_field,
                                                                           interp_weights,
                                                                           nearest_eul_grid_index_to_lag_grid):
        m = lag_grid_field.shape[-1]
        call("smi_ib_spread", self.ctx.handle, as_ptr(eul_grid_field), as_ptr(lag_grid_field),
             as_ptr(interp_weights), as_ptr(nearest_eul_grid_index_to_lag_grid), m, self.n_components)

    def eulerian_grid_ghost_sum(self, local_field):
        call("smi_ghost_sum", self.ctx.handle, as_ptr(local_field), self.n_components)

    def lagrangian_to_eulerian_grid_interpolation_kernel(self, eul_grid_field, lag_grid_field, interp_weights,
                                                         nearest_eul_grid_index_to_lag_grid):
        """:99-113: spread, then add ghost contributions to the neighbours and zero the ghosts."""
        self.lagrangian_to_eulerian_grid_interpolation_kernel_without_ghost_sum(
            eul_grid_field, lag_grid_field, interp_weights, nearest_eul_grid_index_to_lag_grid
        )
        self.eulerian_grid_ghost_sum(local_field=eul_grid_field)


class EulerianLagrangianGridCommunicatorMPI2D:
    """EulerianLagrangianGridCommunicatorMPI2D.py:7-113: index (2, M) int64 (x, y), support
    (2, 4, 4, M) float64, weights (4, 4, M) float64, Lagrangian fields (2, M) or (M,).  The library takes the
    first cell of this rank's rows from the context (row slabs: ``rank * local rows``, x starts at 0)."""

    def __init__(self, dx, eul_grid_coord_shift, interp_kernel_width, real_t, mpi_construct, ghost_size,
                 n_components=1, interp_kernel_type="cosine"):
        if ghost_size < interp_kernel_width:
            raise ValueError(
                f"ghost size ({ghost_size}) needs to be >= interp kernel width ({interp_kernel_width})"
            )
        if interp_kernel_width != 2:
            raise AssertionError("Interpolation kernel inconsistent with interpolation kernel width!")
        if n_components not in (1, 2):
            raise AssertionError("invalid number of components for interpolation!")
        if interp_kernel_type not in _KERNEL_TYPES:
            raise ValueError(
                "Invalid interpolation kernel type. Currently supported types are 'cosine' and 'peskin'."
            )
        self.mpi_construct = mpi_construct
        self.ctx = mpi_construct.device_context(ghost_size)
        self.dx = float(dx)
        self.coord_shift = float(eul_grid_coord_shift)
        self.n_components = n_components
        self.kernel_type = _KERNEL_TYPES[interp_kernel_type]
        # :41-46, x, y order
        self.mpi_substart_idx = np.flip(mpi_construct.grid.coords * mpi_construct.local_grid_size).astype(np.int64)
        self.mpi_local_substart_coord_shift = self.mpi_substart_idx - ghost_size

    def local_eulerian_grid_support_of_lagrangian_grid_kernel(self, local_eul_grid_support_of_lag_grid,
                                                              nearest_eul_grid_index_to_lag_grid, lag_positions):
        call("smi2d_ib_support", self.ctx.handle, as_ptr(local_eul_grid_support_of_lag_grid),
             as_ptr(nearest_eul_grid_index_to_lag_grid), as_ptr(lag_positions), lag_positions.shape[-1], self.dx,
             self.coord_shift)

    def interpolation_weights_kernel(self, interp_weights, local_eul_grid_support_of_lag_grid):
        call("smi2d_ib_weights", self.ctx.handle, as_ptr(interp_weights), as_ptr(local_eul_grid_support_of_lag_grid),
             interp_weights.shape[-1], self.dx, self.kernel_type)

    def eulerian_to_lagrangian_grid_interpolation_kernel(self, lag_grid_field, eul_grid_field, interp_weights,
                                                         nearest_eul_grid_index_to_lag_grid):
        call("smi2d_ib_interpolate", self.ctx.handle, as_ptr(lag_grid_field), as_ptr(eul_grid_field),
             as_ptr(interp_weights), as_ptr(nearest_eul_grid_index_to_lag_grid), lag_grid_field.shape[-1],
             self.n_components, self.dx)

    def lagrangian_to_eulerian_grid_interpolation_kernel_without_ghost_sum(self, eul_grid_field, lag_grid_field,
                                                                           interp_weights,
                                                                           nearest_eul_grid_index_to_lag_grid):
        call("smi2d_ib_spread", self.ctx.handle, as_ptr(eul_grid_field), as_ptr(lag_grid_field),
             as_ptr(interp_weights), as_ptr(nearest_eul_grid_index_to_lag_grid), lag_grid_field.shape[-1],
             self.n_components, 0)

    def lagrangian_to_eulerian_grid_interpolation_kernel(self, eul_grid_field, lag_grid_field, interp_weights,
                                                         nearest_eul_grid_index_to_lag_grid):
        """:99-113: spread, then the ghost sum (ghost rows are added to the neighbouring ranks' edge rows, then
        the ghost cells are cleared)."""
        call("smi2d_ib_spread", self.ctx.handle, as_ptr(eul_grid_field), as_ptr(lag_grid_field),
             as_ptr(interp_weights), as_ptr(nearest_eul_grid_index_to_lag_grid), lag_grid_field.shape[-1],
             self.n_components, 1)


class _GhostSumCommunicator:
    """What the Eulerian-Lagrangian communicators use after a spread, as a class of its own like the reference's
    (EulerianLagrangianGridCommunicatorMPI3D.py:592-840, ...MPI2D.py:521-652): ``ghost_sum`` adds what landed in this
    rank's ghost cells to the neighbours' edge cells and zeroes the ghosts, ``clear_ghost_cells`` only zeroes them.
    Scalar fields, or vector fields with the components leading."""

    grid_dim = None

    def __init__(self, ghost_size, mpi_construct):
        if ghost_size < 0 or not isinstance(ghost_size, (int, np.integer)):
            raise ValueError(f"Ghost size {ghost_size} needs to be an integer >= 0 for field communication.")
        self.mpi_construct = mpi_construct
        self.ghost_size = int(ghost_size)
        self.ctx = mpi_construct.device_context(self.ghost_size)
        self.field_size = np.asarray(mpi_construct.local_grid_size) + 2 * self.ghost_size

    def _ncomp(self, field):
        return 1 if field.ndim == self.grid_dim else int(field.shape[0])


class MPIGhostSumCommunicator3D(_GhostSumCommunicator):
    grid_dim = 3

    def ghost_sum(self, local_field):
        call("smi_ghost_sum", self.ctx.handle, as_ptr(local_field), self._ncomp(local_field))

    def clear_ghost_cells(self, field):
        call("smi_clear_ghost_cells", self.ctx.handle, as_ptr(field), self._ncomp(field))


class MPIGhostSumCommunicator2D(_GhostSumCommunicator):
    grid_dim = 2

    def ghost_sum(self, local_field):
        call("smi2d_ghost_sum", self.ctx.handle, as_ptr(local_field), self._ncomp(local_field), 0)

    def clear_ghost_cells(self, field):
        call("smi2d_ghost_sum", self.ctx.handle, as_ptr(field), self._ncomp(field), 1)


class VirtualBoundaryForcingMPI:
    """VirtualBoundaryForcingMPI.py:21-459.  3D as described below; 2D runs the
    communicator's four kernels one after the other and forms the penalty force with a fifth.

    Global Lagrangian arrays (``global_lag_grid_*``) are host numpy arrays owned by the master
    rank as in the reference; the ``local_lag_grid_*`` arrays of this slab's nodes live on the
    device.  Steps 1-5 of the interaction run as one fused kernel (index, delta weights,
    E->L interpolation, mismatch, penalty force); the spread is a second kernel.
    """

    def __init__(self, mpi_construct, ghost_size, virtual_boundary_stiffness_coeff, virtual_boundary_damping_coeff,
                 grid_dim, dx, eul_grid_coord_shift=None, interp_kernel_width=None,
                 enable_eul_grid_forcing_reset=True, start_time=0.0, master_rank=0,
                 global_lag_grid_position_field=None, assume_data_locality=False, interp_kernel_type="cosine"):
        if grid_dim != 2 and grid_dim != 3:
            raise ValueError("Invalid grid dimensions for virtual boundary forcing!")
        self.grid_dim = grid_dim
        if grid_dim == 2 and mpi_construct.size == 1:
            assume_data_locality = True  # one rank owns every node
        self.virtual_boundary_stiffness_coeff = virtual_boundary_stiffness_coeff
        self.virtual_boundary_damping_coeff = virtual_boundary_damping_coeff
        self.time = start_time
        self.assume_data_locality = assume_data_locality
        self.eul_grid_real_t = mpi_construct.real_t
        self.lag_grid_real_t = np.dtype(np.float64)
        if eul_grid_coord_shift is None:
            eul_grid_coord_shift = self.eul_grid_real_t(dx / 2)
        self.interp_kernel_width = 2 if interp_kernel_width is None else interp_kernel_width
        self.ghost_size = ghost_size
        if self.interp_kernel_width > ghost_size:
            raise ValueError(
                f"Field ghost size {ghost_size} needs to be larger than "
                f"interpolation kernel width {self.interp_kernel_width}"
            )
        self.mpi_construct = mpi_construct
        self.master_rank = master_rank
        self.dx = float(dx)
        comm_cls = EulerianLagrangianGridCommunicatorMPI3D if grid_dim == 3 else EulerianLagrangianGridCommunicatorMPI2D
        self.eul_lag_grid_communicator = comm_cls(
            dx=dx, eul_grid_coord_shift=eul_grid_coord_shift, interp_kernel_width=self.interp_kernel_width,
            real_t=self.eul_grid_real_t, n_components=grid_dim, mpi_construct=mpi_construct, ghost_size=ghost_size,
            interp_kernel_type=interp_kernel_type,
        )
        self.ctx = self.eul_lag_grid_communicator.ctx
        if not assume_data_locality:
            lag_comm_cls = MPILagrangianFieldCommunicator3D if grid_dim == 3 else MPILagrangianFieldCommunicator2D
            self.mpi_lagrangian_field_communicator = lag_comm_cls(
                eul_grid_dx=dx, eul_grid_coord_shift=eul_grid_coord_shift, mpi_construct=mpi_construct,
                master_rank=master_rank, real_t=np.float64,
            )
            self.mpi_lagrangian_field_communicator.map_lagrangian_nodes_based_on_position(
                global_lag_positions=global_lag_grid_position_field
            )
            self.local_num_lag_nodes = self.mpi_lagrangian_field_communicator.local_num_lag_nodes
            self.global_num_lag_nodes = self.mpi_lagrangian_field_communicator.rank_address.shape[-1]
        else:
            self.local_num_lag_nodes = global_lag_grid_position_field.shape[-1]
            self.global_num_lag_nodes = self.local_num_lag_nodes
        self._init_local_buffers(self.local_num_lag_nodes)
        self._init_global_buffers()
        self.enable_eul_grid_forcing_reset = enable_eul_grid_forcing_reset
        if enable_eul_grid_forcing_reset:
            self.compute_interaction_forcing = (
                self.compute_interaction_force_on_eul_and_lag_grid_with_eul_grid_forcing_reset
            )
        else:
            self.compute_interaction_forcing = self.compute_interaction_force_on_eul_and_lag_grid

    # ---------------------------------------------------------------- buffers
    def _init_global_buffers(self):
        shape = (self.grid_dim, self.global_num_lag_nodes)
        self.global_lag_grid_position_mismatch_field = np.zeros(shape, dtype=np.float64)
        self.global_lag_grid_velocity_mismatch_field = np.zeros(shape, dtype=np.float64)
        self._global_lag_grid_forcing_field = np.zeros(shape, dtype=np.float64)
        self._forcing_on_device_is_newer = False

    @property
    def global_lag_grid_forcing_field(self):
        """Host (dim, M) array of the Lagrangian forcing (the master rank's, as in the reference).  On a single rank
        the copy from the device is made when somebody reads the array, not after every interaction: that copy is a
        synchronisation point, and a fixed or prescribed body never looks at the node forces."""
        if self._forcing_on_device_is_newer:
            self._forcing_on_device_is_newer = False
            self._global_lag_grid_forcing_field[...] = self.local_lag_grid_forcing_field.get()
        return self._global_lag_grid_forcing_field

    def _init_local_buffers(self, num_lag_nodes):
        """:201-236, device resident.  Support / weights arrays are not materialised by the
        fused path; ask the communicator's kernels for them when needed."""
        shape = (self.grid_dim, num_lag_nodes)
        z = lambda: self.ctx.zeros(shape, np.float64)  # noqa: E731
        self.local_lag_grid_flow_velocity_field = z()
        self.local_lag_grid_position_mismatch_field = z()
        self.local_lag_grid_velocity_mismatch_field = z()
        self.local_lag_grid_forcing_field = z()
        # positions and velocities share one allocation so that they travel in one host-to-device copy
        self._local_lag_grid_state = self.ctx.zeros((2,) + shape, np.float64)
        self.local_lag_grid_position_field = self._local_lag_grid_state[0]
        self.local_lag_grid_velocity_field = self._local_lag_grid_state[1]
        if self.grid_dim == 2:
            win = (4, 4, num_lag_nodes)
            self.local_eul_grid_support_of_lag_grid = self.ctx.zeros((2,) + win, np.float64)
            self.nearest_eul_grid_index_to_lag_grid = self.ctx.zeros((2, num_lag_nodes), np.int64)
            self.interp_weights = self.ctx.zeros(win, np.float64)

    # True: gather and re-scatter the mismatch state on every call like the reference (:243-251), e.g. to keep the
    # global mismatch arrays current at all times; the default does it only when a node changes slabs
    resync_every_call = False

    def update_buffers(self, global_lag_grid_position_field):
        """:238-276: re-map nodes to slabs; re-scatter the mismatch state when ownership changed."""
        comm = self.mpi_lagrangian_field_communicator
        # The owner map first (host arithmetic on the master rank and M int32 on the wire).  The mismatch state
        # leaves the device, travels to the master and back out only when a node has changed slabs; between such
        # events (most calls: a body moves a fraction of a cell per step, and the interpolation runs once per
        # sub-step of the body's own integrator) the reference's per-call gather of both fields is skipped, and
        # the global mismatch arrays are brought up to date on demand (sync_global_mismatch_fields).
        new_address = comm.owner_ranks_of(global_lag_grid_position_field)
        if np.array_equal(new_address, comm.rank_address) and not self.resync_every_call:
            return
        self.sync_global_mismatch_fields()   # with the ownership the local arrays were laid out for
        comm.adopt_owner_ranks(new_address)
        self.local_num_lag_nodes = comm.local_num_lag_nodes
        self._init_local_buffers(self.local_num_lag_nodes)
        pm = np.zeros((self.grid_dim, self.local_num_lag_nodes))
        vm = np.zeros_like(pm)
        comm.scatter_global_field(pm, self.global_lag_grid_position_mismatch_field)
        comm.scatter_global_field(vm, self.global_lag_grid_velocity_mismatch_field)
        self.local_lag_grid_position_mismatch_field.set(pm)
        self.local_lag_grid_velocity_mismatch_field.set(vm)

    def sync_global_mismatch_fields(self):
        """Gather the device-resident mismatch state of every slab into ``global_lag_grid_position_mismatch_field``
        and ``global_lag_grid_velocity_mismatch_field`` on the master rank (:243-251 does this on every call;
        here it happens when ownership changes, and whenever a caller asks, e.g. before writing a restart)."""
        if self.assume_data_locality or self.mpi_construct.size == 1:
            self.global_lag_grid_position_mismatch_field[...] = self.local_lag_grid_position_mismatch_field.get()
            self.global_lag_grid_velocity_mismatch_field[...] = self.local_lag_grid_velocity_mismatch_field.get()
            return
        comm = self.mpi_lagrangian_field_communicator
        comm.gather_local_field(self.global_lag_grid_position_mismatch_field,
                                self.local_lag_grid_position_mismatch_field.get())
        comm.gather_local_field(self.global_lag_grid_velocity_mismatch_field,
                                self.local_lag_grid_velocity_mismatch_field.get())

    def scatter_global_mismatch_fields(self, global_lag_grid_position_field=None):
        """The converse, for a restart: the global mismatch arrays (just read from a file on the master rank) go to the
        device arrays of the slabs that own the nodes at ``global_lag_grid_position_field`` - the re-map, buffer
        re-initialisation and two scatters of flow_past_freely_rotating_rod_case.py:228-247 in one call."""
        if self.assume_data_locality or self.mpi_construct.size == 1:
            self.local_lag_grid_position_mismatch_field.set(self.global_lag_grid_position_mismatch_field)
            self.local_lag_grid_velocity_mismatch_field.set(self.global_lag_grid_velocity_mismatch_field)
            return
        comm = self.mpi_lagrangian_field_communicator
        comm.map_lagrangian_nodes_based_on_position(global_lag_grid_position_field)
        self.local_num_lag_nodes = comm.local_num_lag_nodes
        self._init_local_buffers(self.local_num_lag_nodes)
        pm = np.zeros((self.grid_dim, self.local_num_lag_nodes))
        vm = np.zeros_like(pm)
        comm.scatter_global_field(pm, self.global_lag_grid_position_mismatch_field)
        comm.scatter_global_field(vm, self.global_lag_grid_velocity_mismatch_field)
        self.local_lag_grid_position_mismatch_field.set(pm)
        self.local_lag_grid_velocity_mismatch_field.set(vm)

    # ---------------------------------------------------------------- the pointwise kernels, one at a time
    @staticmethod
    def _axpby(out, a, x, b, y):
        dst = as_ptr(out)     # TypeError for anything but a device array
        call("smi_ib_axpby", out.ctx.handle, dst, float(a), x.tracked_ptr(), float(b), y.tracked_ptr(), out.size)

    @staticmethod
    def compute_lag_grid_velocity_mismatch_field(lag_grid_velocity_mismatch_field, lag_grid_flow_velocity_field,
                                                 lag_grid_body_velocity_field):
        """:278-293 on device-resident float64 arrays: flow velocity - body velocity."""
        VirtualBoundaryForcingMPI._axpby(lag_grid_velocity_mismatch_field, 1.0, lag_grid_flow_velocity_field, -1.0,
                                         lag_grid_body_velocity_field)

    @staticmethod
    def update_lag_grid_position_mismatch_field_via_euler_forward(lag_grid_position_mismatch_field,
                                                                  lag_grid_velocity_mismatch_field, dt):
        """:295-310: position mismatch += dt * velocity mismatch."""
        VirtualBoundaryForcingMPI._axpby(lag_grid_position_mismatch_field, 1.0, lag_grid_position_mismatch_field, dt,
                                         lag_grid_velocity_mismatch_field)

    @staticmethod
    def compute_lag_grid_forcing_field(lag_grid_forcing_field, lag_grid_position_mismatch_field,
                                       lag_grid_velocity_mismatch_field, virtual_boundary_stiffness_coeff,
                                       virtual_boundary_damping_coeff):
        """:312-331: the penalty force of the virtual boundary method."""
        VirtualBoundaryForcingMPI._axpby(lag_grid_forcing_field, virtual_boundary_stiffness_coeff,
                                         lag_grid_position_mismatch_field, virtual_boundary_damping_coeff,
                                         lag_grid_velocity_mismatch_field)

    # ---------------------------------------------------------------- interaction
    def _upload_lagrangian_state(self, global_lag_grid_position_field, global_lag_grid_velocity_field):
        m = self.local_num_lag_nodes
        if self.assume_data_locality or self.mpi_construct.size == 1:
            pos = np.asarray(global_lag_grid_position_field, dtype=np.float64)
            vel = np.asarray(global_lag_grid_velocity_field, dtype=np.float64)
        else:
            comm = self.mpi_lagrangian_field_communicator
            pos = np.zeros((self.grid_dim, m))
            vel = np.zeros((self.grid_dim, m))
            comm.scatter_global_field(pos, global_lag_grid_position_field)
            comm.scatter_global_field(vel, global_lag_grid_velocity_field)
        if m:
            self._local_lag_grid_state.set(np.stack((pos, vel)))

    def compute_interaction_force_on_lag_grid(self, local_eul_grid_velocity_field, global_lag_grid_position_field,
                                              global_lag_grid_velocity_field, _download=True):
        """:333-406.  ``_download=False`` (internal): the caller fetches the forces itself, after it has
        queued the kernels that follow, so that the device does not idle during the copy."""
        multi = self.mpi_construct.size > 1
        if isinstance(global_lag_grid_position_field, DeviceResidentNodes):
            # the node arrays are on the device already (a framed forcing grid formed them there): no upload
            self._check_inside_domain(global_lag_grid_position_field)
        else:
            if not self.assume_data_locality and multi:
                self.update_buffers(global_lag_grid_position_field)
            elif not self.assume_data_locality:
                # one slab owns every node: only the inside-the-domain check of the mapping remains
                self.mpi_lagrangian_field_communicator._compute_lag_nodes_rank_address(global_lag_grid_position_field)
            self._upload_lagrangian_state(global_lag_grid_position_field, global_lag_grid_velocity_field)
        comm = self.eul_lag_grid_communicator
        if self.grid_dim == 2:
            # :360-397 step by step: support and index, weights, E -> L velocity, mismatch, penalty force
            comm.local_eulerian_grid_support_of_lagrangian_grid_kernel(
                self.local_eul_grid_support_of_lag_grid, self.nearest_eul_grid_index_to_lag_grid,
                self.local_lag_grid_position_field)
            comm.interpolation_weights_kernel(self.interp_weights, self.local_eul_grid_support_of_lag_grid)
            comm.eulerian_to_lagrangian_grid_interpolation_kernel(
                self.local_lag_grid_flow_velocity_field, local_eul_grid_velocity_field, self.interp_weights,
                self.nearest_eul_grid_index_to_lag_grid)
            call("smi_ib_penalty_force", self.ctx.handle, as_ptr(self.local_lag_grid_forcing_field),
                 as_ptr(self.local_lag_grid_velocity_mismatch_field), as_ptr(self.local_lag_grid_flow_velocity_field),
                 as_ptr(self.local_lag_grid_velocity_field), as_ptr(self.local_lag_grid_position_mismatch_field),
                 self.grid_dim * self.local_num_lag_nodes, float(self.virtual_boundary_stiffness_coeff),
                 float(self.virtual_boundary_damping_coeff))
            if _download:
                self._download_lag_grid_forcing()
            return
        call("smi_ib_forcing_fused", self.ctx.handle, as_ptr(self.local_lag_grid_forcing_field),
             as_ptr(self.local_lag_grid_velocity_mismatch_field), as_ptr(self.local_lag_grid_flow_velocity_field),
             as_ptr(self.local_lag_grid_position_mismatch_field), as_ptr(self.local_lag_grid_position_field),
             as_ptr(self.local_lag_grid_velocity_field), local_eul_grid_velocity_field.tracked_ptr(),  # read only
             self.local_num_lag_nodes, comm.dx, comm.coord_shift, comm._substart,
             float(self.virtual_boundary_stiffness_coeff), float(self.virtual_boundary_damping_coeff),
             comm.kernel_type)
        if _download:
            self._download_lag_grid_forcing()

    def _download_lag_grid_forcing(self):
        """Gather of the Lagrangian forcing to the master rank (:402-406): a collective across ranks, on demand
        (see ``global_lag_grid_forcing_field``) on one."""
        if not self.assume_data_locality and self.mpi_construct.size > 1:
            self.mpi_lagrangian_field_communicator.gather_local_field(
                self._global_lag_grid_forcing_field, self.local_lag_grid_forcing_field.get()
            )
        else:
            self._forcing_on_device_is_newer = True

    def _check_inside_domain(self, nodes):
        """The reference stops when a node leaves the Eulerian domain (utils/mpi_utils_3d.py:1371-1379).  With the
        nodes on the device the box around them is checked; only when the box sticks out are the nodes fetched."""
        comm = self.mpi_lagrangian_field_communicator if not self.assume_data_locality else None
        if comm is None:
            return
        extent = comm.eul_subblock_dx[::-1] * np.asarray(self.mpi_construct.grid_topology)[::-1]   # x, y, z
        lo, hi = nodes.lo - comm.eul_grid_coord_shift, nodes.hi - comm.eul_grid_coord_shift
        if np.all(hi < extent) and np.all(lo >= 0):
            return
        comm._compute_lag_nodes_rank_address(self.local_lag_grid_position_field.get())

    def compute_interaction_force_on_eul_and_lag_grid(self, local_eul_grid_forcing_field,
                                                      local_eul_grid_velocity_field, global_lag_grid_position_field,
                                                      global_lag_grid_velocity_field):
        """:408-429."""
        self.compute_interaction_force_on_lag_grid(
            local_eul_grid_velocity_field, global_lag_grid_position_field, global_lag_grid_velocity_field,
            _download=False,
        )
        comm = self.eul_lag_grid_communicator
        if self.grid_dim == 2:
            comm.lagrangian_to_eulerian_grid_interpolation_kernel(
                local_eul_grid_forcing_field, self.local_lag_grid_forcing_field, self.interp_weights,
                self.nearest_eul_grid_index_to_lag_grid)
            self._download_lag_grid_forcing()  # after the spread has been queued
            return
        # the spread states its footprint (the 4^3 windows of the nodes), so the record of where the forcing
        # field is non-zero survives and the flow step can skip the rest of the grid
        footprint = self._spread_footprint(global_lag_grid_position_field)
        call("smi_ib_spread_fused", self.ctx.handle, local_eul_grid_forcing_field.tracked_ptr(),
             as_ptr(self.local_lag_grid_forcing_field), as_ptr(self.local_lag_grid_position_field),
             self.local_num_lag_nodes, comm.dx, comm.coord_shift, comm._substart, comm.kernel_type)
        if footprint is None:
            local_eul_grid_forcing_field._alloc.support = None
        elif footprint:
            local_eul_grid_forcing_field._alloc.widen_support(*footprint)
        if self.mpi_construct.size > 1:
            comm.eulerian_grid_ghost_sum(local_field=local_eul_grid_forcing_field)
        else:
            # one rank: the ghost sum only clears the ghost cells, which takes nothing out of the footprint
            call("smi_clear_ghost_cells", self.ctx.handle, local_eul_grid_forcing_field.tracked_ptr(), self.grid_dim)
        self._download_lag_grid_forcing()  # after the spread has been queued

    def _spread_footprint(self, global_lag_grid_position_field):
        """Padded-index corners (z, y, x order) of the cells the spread of these nodes can touch: the
        window [index - 1, index + 2] of every node (EulerianLagrangianGridCommunicatorMPI3D.py:221-235) with
        index = floor((X - shift) / dx) + ghost.  ``()`` without nodes, None when it cannot be stated
        (several ranks: the owner map decides which nodes are local)."""
        if self.mpi_construct.size > 1 or self.grid_dim != 3:
            return None
        comm = self.eul_lag_grid_communicator
        if isinstance(global_lag_grid_position_field, DeviceResidentNodes):
            if global_lag_grid_position_field.num_nodes == 0:
                return ()
            pos = np.stack((global_lag_grid_position_field.lo, global_lag_grid_position_field.hi), axis=1)
        else:
            pos = np.asarray(global_lag_grid_position_field, dtype=np.float64)
        if pos.shape[-1] == 0:
            return ()
        idx = np.floor((pos - comm.coord_shift) / comm.dx).astype(np.int64) + self.ghost_size
        # one more cell on either side: the device floors the same quotient in its own arithmetic
        lo, hi = idx.min(axis=1) - 2, idx.max(axis=1) + 3
        return tuple(lo[::-1]), tuple(hi[::-1])

    def compute_interaction_force_on_eul_and_lag_grid_with_eul_grid_forcing_reset(
        self, local_eul_grid_forcing_field, local_eul_grid_velocity_field, global_lag_grid_position_field,
        global_lag_grid_velocity_field,
    ):
        """:431-450."""
        call("smi_set_fixed_val" if self.grid_dim == 3 else "smi2d_set_fixed_val", self.ctx.handle,
             as_ptr(local_eul_grid_forcing_field), self.grid_dim, _lib.darray([0.0] * self.grid_dim))
        local_eul_grid_forcing_field._alloc.support = EMPTY_SUPPORT
        self.compute_interaction_force_on_eul_and_lag_grid(
            local_eul_grid_forcing_field, local_eul_grid_velocity_field, global_lag_grid_position_field,
            global_lag_grid_velocity_field,
        )

    def time_step(self, dt):
        """:452-459."""
        n = self.grid_dim * self.local_num_lag_nodes
        call("smi_ib_mismatch_step", self.ctx.handle, as_ptr(self.local_lag_grid_position_mismatch_field),
             as_ptr(self.local_lag_grid_velocity_mismatch_field), n, float(dt))
        self.time += dt


class DeviceResidentNodes:
    """Stands in for the host (3, M) position / velocity arrays of an interaction whose node arrays are on the
    device already (a framed forcing grid formed them there): all the host needs is a box, in physical
    coordinates (x, y, z), that holds every node."""

    def __init__(self, lo, hi, num_nodes):
        self.lo, self.hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
        self.num_nodes = int(num_nodes)
        self.shape = (3, self.num_nodes)


__all__ = ["DeviceResidentNodes", "EulerianLagrangianGridCommunicatorMPI2D", "EulerianLagrangianGridCommunicatorMPI3D",
           "MPIGhostSumCommunicator2D", "MPIGhostSumCommunicator3D", "VirtualBoundaryForcingMPI"]
