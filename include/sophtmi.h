/*
 * sophtmi.h -- C ABI of libsophtmi.so: the MI355X (gfx950) implementation of the
 * sopht-mpi Eulerian flow-step hot path.
 *
 * The reference (fankiat/sopht-mpi) is pure Python and has no FFI boundary; its
 * boundary is a Python calling convention (generator functions returning operator
 * closures).  Every entry point below names the reference operator it replaces as
 * file:line relative to the reference tree.  Conventions:
 *
 *   - every function returns 0 on success, non-zero on failure; the message of the
 *     last failure on the calling thread is smi_last_error().  Nothing throws.
 *   - all field pointers are DEVICE pointers obtained from smi_malloc.  Fields are
 *     C-ordered [z][y][x] arrays of the context's real type INCLUDING `ghost` ghost
 *     layers per side, vector fields are (3, z, y, x) with component 0 = x, exactly
 *     the reference's local arrays (flow_simulators_mpi_3d.py:170-196).
 *   - real scalars cross the ABI as double and are rounded to the context's real
 *     type inside (pass the reference's real_t(...) value; the round trip is exact).
 *   - calls enqueue on the context's HIP stream and return; smi_sync waits.  Entry
 *     points that return a value to the host (reductions, downloads) synchronise.
 *   - one context = one GPU = one z-slab of the global grid (the reference's
 *     rank_distribution=(0,1,1) slab mode, utils/mpi_utils_3d.py:32-66).
 */
#ifndef SOPHTMI_H
#define SOPHTMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct smi_ctx smi_ctx;
typedef struct smi_poisson smi_poisson;

enum { SMI_F32 = 0, SMI_F64 = 1 };
/* Poisson back ends */
enum { SMI_POISSON_ROCFFT = 0, SMI_POISSON_PRUNED = 1 };
/* Laplacian filter types (laplacian_filter_mpi_3d.py:387-397) */
enum { SMI_FILTER_MULTIPLICATIVE = 0, SMI_FILTER_CONVOLUTION = 1 };
/* delta kernels (EulerianLagrangianGridCommunicatorMPI3D.py:81-97) */
enum { SMI_DELTA_COSINE = 0, SMI_DELTA_PESKIN = 1 };

const char* smi_last_error(void);
int smi_version(void);
int smi_device_count(int* count);

/* ---- context / topology -------------------------------------------------------
 * Replaces MPIConstruct3D + MPIGhostCommunicator3D construction
 * (utils/mpi_utils_3d.py:14-83, 118-139).  nz/ny/nx: LOCAL interior sizes of this
 * slab; rank/nranks: position in the z-slab decomposition (physical z boundaries
 * are rank 0's low face and rank nranks-1's high face; y and x faces are always
 * physical).  dtype: SMI_F32 / SMI_F64.
 *
 * Devices and threads: the create calls make `device` the calling thread's current HIP device; every other entry
 * point launches on the context's own stream WITHOUT switching devices, and HIP binds a stream to the device it
 * was created on.  The supported arrangement is the reference's: one process per GPU, one context per process,
 * calls from the thread that created the context.  A process that holds contexts on several devices must make
 * the context's device current (hipSetDevice) on the calling thread before each call. */
int smi_ctx_create(smi_ctx** out, int device, int dtype, int nz, int ny, int nx, int ghost,
                   int rank, int nranks);
int smi_ctx_destroy(smi_ctx* ctx);
int smi_sync(smi_ctx* ctx);
/* Two contexts of one process (a pencil context and the slab context its Poisson solve runs on) have their own
 * streams: what has been queued on `other` completes before what is queued on `waiter` afterwards starts, without a
 * host synchronisation (the reference's calls are synchronous, so it has no counterpart). */
int smi_ctx_wait_for(smi_ctx* waiter, smi_ctx* other);
/* Execution switches of a context (no counterpart in the reference; tests and measurements).  smi_ctx_create reads
 * them once from the environment and no entry point consults the environment afterwards:
 *   "no_fused_advect_diffuse" (SMI_NO_FUSED_ADVECT_DIFFUSE)  1: advection and diffusion as two passes
 *   "fused_split_launch"      (SMI_FUSED_SPLIT_LAUNCH)       1: interior z chunks and outer z chunks as two launches
 *   "fused_zchunk"            (SMI_FUSED_ZCHUNK)             planes per workgroup of the one-pass kernel, 0 = automatic
 *   "check_forcing_support"   (SMI_CHECK_FORCING_SUPPORT)    1: the step verifies a promised forcing box before use
 *   "transpose_chunks"        (SMI_FFT_TRANSPOSE_CHUNKS)     1, 2 or 4 z chunks per spectral transpose across ranks
 *                                                            (default: 1) */
int smi_ctx_set_option(smi_ctx* ctx, const char* name, int value);
int smi_ctx_get_option(smi_ctx* ctx, const char* name, int* value);

/* RCCL communicator over the slab ranks.  unique_id: the 128-byte ncclUniqueId made
 * by smi_comm_unique_id on rank 0 and broadcast by the host launcher.  Replaces
 * MPI.COMM_WORLD.Create_cart (utils/mpi_utils_3d.py:64-66). */
int smi_comm_unique_id(void* out128);
int smi_comm_init(smi_ctx* ctx, const void* unique_id128);
/* Host-side transport hook.  When set, every halo exchange (kind 0: all ghost planes; kind 16 d:
 * only the d planes next to the interior) and ghost sum (kind 1) the library needs is handed to
 * `cb` instead of RCCL; the callback moves whole
 * ghost planes with smi_planes_d2h / smi_planes_h2d.  For ranks that share one GPU.
 * Replaces nothing in the reference (its transport is always MPI). */
typedef int (*smi_exchange_cb)(void* user, void* field, int ncomp, int kind);
int smi_set_exchange_callback(smi_ctx* ctx, smi_exchange_cb cb, void* user);
/* Same hook for the FFT transposes: block r (bytes_per_peer bytes) of the device buffer
 * `send` goes to rank r, block s of `recv` comes from rank s.  `user` is the pointer given
 * to smi_set_exchange_callback. */
typedef int (*smi_alltoall_cb)(void* user, const void* send, void* recv, size_t bytes_per_peer);
int smi_set_alltoall_callback(smi_ctx* ctx, smi_alltoall_cb cb);

/* ---- pencil decomposition -------------------------------------------------------
 * The reference's default for several ranks (utils/mpi_utils_3d.py:32-66: MPI.Compute_dims over z and y,
 * x kept whole).  smi_ctx_set_pencil turns a context created as rank `rank_z * nranks_y + rank_y` of
 * nranks_z * nranks_y ranks (the Cartesian order of Create_cart) into one block of a (nranks_z, nranks_y)
 * grid: nz and ny of smi_ctx_create are then the local sizes of that block.  Halo exchange, ghost sum and
 * the band logic of every operator follow the block's position; the Poisson solve runs on z-slabs
 * (smi_pencil_to_slab / smi_slab_to_pencil move the right-hand side and the solution).
 * Host-staged transports for tests: a pairwise exchange of device buffers with the two y neighbours
 * (peer -1 = none) and an all-to-all of equal blocks inside a group of `count` consecutive ranks. */
int smi_ctx_set_pencil(smi_ctx* ctx, int rank_z, int nranks_z, int rank_y, int nranks_y);
/* The same with x divided as well (rank_distribution (1, 1, 0), (1, 0, 0), (0, 1, 0) of the reference's operator
 * tests): rank = (rank_z * nranks_y + rank_y) * nranks_x + rank_x, nz / ny / nx of smi_ctx_create the local sizes of
 * the block.  Exchanges add the x faces (packed columns), the zero bands of the operators apply on physical x faces
 * only, the Poisson solve is routed through z-slabs as for pencils.  Kernels that work on 16-byte row vectors fall
 * back to their scalar forms when the local row (nx + 2 ghost) is no multiple of four. */
int smi_ctx_set_grid3(smi_ctx* ctx, int rank_z, int nranks_z, int rank_y, int nranks_y, int rank_x, int nranks_x);
typedef int (*smi_sendrecv_cb)(void* user, int peer_lo, int peer_hi, const void* send_lo, const void* send_hi,
                               void* recv_lo, void* recv_hi, size_t bytes);
int smi_set_sendrecv_callback(smi_ctx* ctx, smi_sendrecv_cb cb);
typedef int (*smi_group_alltoall_cb)(void* user, const void* send, void* recv, size_t bytes_per_peer,
                                     int first_rank, int count);
int smi_set_group_alltoall_callback(smi_ctx* ctx, smi_group_alltoall_cb cb);
/* ... and a few point-to-point messages of `bytes` bytes each between device buffers (the Isend / Irecv pairs of
 * MPIDomainDoublingCommunicator3D, UnboundedPoissonSolverMPI3D.py:314-382): sends[i] goes to rank send_peers[i],
 * recvs[j] comes from rank recv_peers[j]. */
typedef int (*smi_p2p_cb)(void* user, int nsend, const int* send_peers, const void* const* sends, int nrecv,
                          const int* recv_peers, void* const* recvs, size_t bytes);
int smi_set_p2p_callback(smi_ctx* ctx, smi_p2p_cb cb);
/* ---- general block topology: the ghost exchange on any decomposition, periodic or not ---------------
 * MPIConstruct3D / 2D accept any rank_distribution and periodic_domain (utils/mpi_utils_3d.py:7-83,
 * utils/mpi_utils_2d.py:9-84), and the reference's own tests of MPIGhostCommunicator3D / 2D
 * (utils/mpi_utils_3d.py:86-1142, utils/mpi_utils_2d.py:87-421) run on periodic domains with x divided.
 * smi_ctx_set_block turns a context created as rank `(cz * Py + cy) * Px + cx` (the Cartesian order of Create_cart)
 * into block (cz, cy, cx) of a (Pz, Py, Px) grid with the given periodic axes; smi_halo_exchange_general then fills
 * the ghost cells of `field` from the neighbouring blocks (faces; with full_exchange also edges and vertices; a
 * periodic axis with one rank wraps onto the block itself).  Messages travel through smi_set_p2p_callback or RCCL.
 * The operators of this library keep x whole and the domain unbounded: on such a context they fail with an
 * error, only the exchange is served. */
int smi_ctx_set_block(smi_ctx* ctx, const int* coords_zyx, const int* topo_zyx, const int* periodic_zyx);
/* smi_ctx_create + smi_ctx_set_block for blocks as thin as their ghost layers (smi_ctx_create asks for two interior
 * layers per ghost layer, which the operators need; the reference's communicator tests run 8-point axes over 4 ranks
 * with 2 ghost layers). */
int smi_ctx_create_block(smi_ctx** out, int device, int dtype, int nz, int ny, int nx, int ghost, int rank, int nranks,
                         const int* coords_zyx, const int* topo_zyx, const int* periodic_zyx);
int smi_halo_exchange_general(smi_ctx* ctx, void* field, int ncomp, int full_exchange);
/* MPIDomainDoublingCommunicator-style redistribution between the two decompositions of one grid:
 * `pencil_field` (ncomp components, ghosted, on `pencil`) <-> `slab_field` (ghosted, on `slab`, a context of
 * the same rank and rank count created with nz = global nz / nranks, ny = global ny).  Interiors only.
 * work_a / work_b: two device buffers of ncomp * nz_pencil * ny_pencil * nx elements. */
int smi_pencil_to_slab(smi_ctx* pencil, smi_ctx* slab, void* slab_field, const void* pencil_field, int ncomp,
                       void* work_a, void* work_b);
int smi_slab_to_pencil(smi_ctx* pencil, smi_ctx* slab, void* pencil_field, const void* slab_field, int ncomp,
                       void* work_a, void* work_b);
/* allreduce of n host doubles over the slab ranks through RCCL; op: 0 = sum, 1 = min,
 * 2 = max.  Replaces grid.allreduce (flow_simulators_mpi_3d.py:448, 465-467, 475). */
int smi_allreduce_f64(smi_ctx* ctx, double* host_inout, int n, int op);

/* ---- in-library kernel timing (HIP events on the context's stream) -----------------
 * No counterpart in the reference (it has no timing code).  Slots are named kernels or
 * kernel groups; bytes = algorithmic bytes summed over the launches. */
int smi_prof_enable(smi_ctx* ctx, int on);
int smi_prof_reset(smi_ctx* ctx);
int smi_prof_num_slots(void);
int smi_prof_get(smi_ctx* ctx, int slot, char* name64, int64_t* launches, double* total_ms,
                 double* alg_bytes);
/* the individual durations (ms) behind a slot's total, in launch order, at most `capacity` (the library keeps the
 * first 4096 per slot since smi_prof_reset); the slot "flow_step" brackets each whole smi_flow_step_ns3d */
int smi_prof_get_samples(smi_ctx* ctx, int slot, double* out_ms, int capacity, int* count);

/* ---- memory ---------------------------------------------------------------- */
int smi_malloc(smi_ctx* ctx, size_t bytes, void** dptr);
int smi_free(smi_ctx* ctx, void* dptr);
int smi_memcpy_h2d(smi_ctx* ctx, void* dst, const void* src, size_t bytes);
int smi_memcpy_d2h(smi_ctx* ctx, void* dst, const void* src, size_t bytes);
int smi_memcpy_d2d(smi_ctx* ctx, void* dst, const void* src, size_t bytes);
int smi_memset_zero(smi_ctx* ctx, void* dptr, size_t bytes);

/* ---- halo exchange ------------------------------------------------------------
 * MPIGhostCommunicator3D.exchange_scalar_field_init / exchange_vector_field_init +
 * exchange_finalise (utils/mpi_utils_3d.py:756-842, 1125-1142) for z-slabs: the first /
 * last `ghost` interior planes go to the neighbour's ghost planes.  Whole padded planes
 * travel (they are contiguous), so the y/x ghost columns of the z ghost planes carry the
 * neighbour's values instead of stale local ones; no interior cell reads them.
 * No-op on a single rank.  Needs smi_comm_init when nranks > 1. */
int smi_halo_exchange(smi_ctx* ctx, void* field, int ncomp);
/* Host-staged transport, for ranks that share one GPU (tests) or have no RCCL
 * communicator: copy `nplanes` whole z planes starting at padded plane index z0 of every
 * component to / from a host buffer laid out [ncomp][nplanes][py][px]; accumulate != 0
 * adds instead of overwriting (ghost sum). */
int smi_planes_d2h(smi_ctx* ctx, const void* field, int ncomp, int z0, int nplanes, void* host);
int smi_planes_h2d(smi_ctx* ctx, void* field, int ncomp, int z0, int nplanes, const void* host,
                   int accumulate);

/* ---- elementwise operators (sopht elementwise kernels; call sites cited) -------- */
/* gen_set_fixed_val_pyst_kernel_3d, flow_simulators_mpi_3d.py:303-305, 422-424 */
int smi_set_fixed_val(smi_ctx* ctx, void* field, int ncomp, const double* vals);
/* gen_add_fixed_val_pyst_kernel_3d, flow_simulators_mpi_3d.py:309-318 */
int smi_add_fixed_val(smi_ctx* ctx, void* sum_field, const void* field, int ncomp, const double* vals);
/* gen_elementwise_sum_pyst_kernel_3d, diffusion_timestep_mpi_3d.py:57-59 */
int smi_elementwise_sum(smi_ctx* ctx, void* sum_field, const void* f1, const void* f2, int ncomp);
/* gen_elementwise_copy_pyst_kernel_3d, laplacian_filter_mpi_3d.py:279 */
int smi_elementwise_copy(smi_ctx* ctx, void* dst, const void* src, int ncomp);
/* gen_elementwise_saxpby_pyst_kernel_3d, laplacian_filter_mpi_3d.py:313-319 */
int smi_elementwise_saxpby(smi_ctx* ctx, void* sum_field, const void* f1, double a, const void* f2,
                           double b, int ncomp);
/* gen_elementwise_cross_product_pyst_kernel_3d, flow_simulators_mpi_3d.py:267-269, 396-401 */
int smi_cross_product(smi_ctx* ctx, void* result, const void* f1, const void* f2);

/* ---- stencil operators ----------------------------------------------------------- */
/* gen_diffusion_flux_pyst_mpi_kernel_3d, stencil_ops_3d/diffusion_flux_mpi_3d.py:10-228 */
int smi_diffusion_flux(smi_ctx* ctx, void* flux, const void* field, double prefactor, int ncomp);
/* gen_diffusion_timestep_euler_forward_pyst_mpi_kernel_3d, diffusion_timestep_mpi_3d.py:11-92.
 * `flux` is ONE scalar scratch field, reused per component as in the reference. */
int smi_diffusion_timestep(smi_ctx* ctx, void* field, void* flux, double nu_dt_by_dx2, int ncomp);
/* gen_update_vorticity_from_velocity_forcing_pyst_mpi_kernel_3d,
 * update_vorticity_from_velocity_forcing_mpi_3d.py:9-178 */
int smi_update_vorticity_from_velocity_forcing(smi_ctx* ctx, void* vorticity, void* forcing,
                                               double prefactor);
/* gen_update_vorticity_from_penalised_velocity_pyst_mpi_kernel_3d
 * (stencil_ops_3d/update_vorticity_from_velocity_forcing_mpi_3d.py:181-330):
 * vorticity += prefactor * curl(penalised_velocity - velocity) on the same cells; both inputs are exchanged first. */
int smi_update_vorticity_from_penalised_velocity(smi_ctx* ctx, void* vorticity, void* penalised_velocity,
                                                 void* velocity, double prefactor);
/* Pointwise helpers of the penalisation method, kernel support 0 (every cell of the arrays, ghost cells included,
 * no exchange; a 2D context takes its plane).  gen_brinkmann_penalise_pyst_mpi_kernel_3d / _2d
 * (stencil_ops_3d/brinkmann_penalise_mpi_3d.py:7-19): penalised = (field + lambda chi penalty) / (1 + lambda chi) for
 * ncomp components sharing the one-component char_field.
 * gen_char_func_from_level_set_via_sine_heaviside_pyst_mpi_kernel_3d / _2d
 * (stencil_ops_3d/char_func_from_level_set_mpi_3d.py:8-28): chi = 0 below -blend_width, 1 above blend_width,
 * (1 + phi / w + sin(pi phi / w) / pi) / 2 in between.  The bodies of both are in the sopht package, which the
 * reference tree does not hold: the formulas are the published ones of that method. */
int smi_brinkmann_penalise(smi_ctx* ctx, void* penalised_field, const void* field, const void* char_field,
                           const void* penalty_field, double penalty_factor, int ncomp);
int smi_char_func_from_level_set(smi_ctx* ctx, void* char_func_field, const void* level_set_field, double blend_width);
/* The rotational-form advection of rotational_form_navier_stokes_timestep in one pass
 * (flow_simulators_mpi_3d.py:395-405: elementwise_cross_product(buffer, velocity, vorticity) then
 * update_vorticity_from_velocity_forcing(vorticity, buffer, dt / (2 dx))), written to another array:
 * out = vorticity + prefactor * curl(velocity x vorticity) on the cells the update writes,
 * out = vorticity elsewhere.  `out` must not alias the inputs; all three are vector fields. */
int smi_rotational_advection_update(smi_ctx* ctx, void* out, void* velocity, void* vorticity, double prefactor);
/* The advection and the diffusion of rotational_form_navier_stokes_timestep (flow_simulators_mpi_3d.py:395-411:
 * cross product, vorticity update with dt / (2 dx), vector diffusion timestep with nu dt / dx^2), written to another
 * array: out = w* + c lap7(w*) where the diffusion flux is not zeroed, w* elsewhere, w* = the result of
 * smi_rotational_advection_update.  One pass over the arrays where the context allows it (one rank or z-slabs with
 * at least two ghost layers, 16-byte aligned rows; across slabs one exchange of both ghost planes of velocity and
 * vorticity runs beside the interior chunks), otherwise - pencils, unaligned rows - the two operators through
 * `scratch` (a vector field).  No array may alias another. */
int smi_advection_diffusion_update(smi_ctx* ctx, void* out, void* velocity, void* vorticity, void* scratch,
                                   double prefactor, double nu_dt_by_dx2);
/* gen_curl_pyst_mpi_kernel_3d, curl_mpi_3d.py:10-196 */
int smi_curl(smi_ctx* ctx, void* curl, void* field, double prefactor);
/* gen_divergence_pyst_mpi_kernel_3d, divergence_mpi_3d.py:10-200 */
int smi_divergence(smi_ctx* ctx, void* divergence, void* field, double inv_dx);
/* gen_penalise_field_boundary_pyst_mpi_kernel_3d, penalise_field_boundary_mpi_3d.py:9-269.
 * fac_lo/fac_hi: host arrays, per axis x,y,z: 3 * (ghost+width) sine factors for the low
 * band (index 0 = outermost cell) and for the high band (index 0 = first cell of the band). */
int smi_penalise_boundary(smi_ctx* ctx, void* field, int ncomp, int width, const double* fac_lo,
                          const double* fac_hi);
/* gen_laplacian_filter_mpi_kernel_3d, laplacian_filter_mpi_3d.py:14-421 */
int smi_laplacian_filter(smi_ctx* ctx, void* field, int ncomp, void* filter_flux_buffer,
                         void* field_buffer, int order, int filter_type);
/* The same with filter_flux_buffer_boundary_width (laplacian_filter_mpi_3d.py:23, 110-143): the filter
 * flux is cleared on the ghost_size + boundary_width layers next to every physical face
 * (smi_laplacian_filter is boundary_width = 1, the reference's default). */
int smi_laplacian_filter_w(smi_ctx* ctx, void* field, int ncomp, void* filter_flux_buffer,
                           void* field_buffer, int order, int filter_type, int boundary_width);
/* gen_advection_timestep_euler_forward_conservative_eno3_pyst_mpi_kernel_3d,
 * advection_timestep_mpi_3d.py:12-95 (+ advection_flux_mpi_3d.py:8-195) */
int smi_advection_timestep_eno3(smi_ctx* ctx, void* field, void* flux, void* velocity,
                                double dt_by_dx, int ncomp);
/* gen_advection_flux_conservative_eno3_pyst_mpi_kernel_3d (stencil_ops_3d/advection_flux_mpi_3d.py:8-195):
 * flux = inv_dx * (conservative upwind flux divergence) on the cells the wrapper's boxes cover; the other
 * cells of flux keep their values (the wrapper never writes them). */
int smi_advection_flux_eno3(smi_ctx* ctx, void* flux, void* field, void* velocity, double inv_dx);

/* ---- reductions (flow_simulators_mpi_3d.py:426-476); LOCAL values, the host
 *      launcher combines ranks (allreduce MIN / MAX / SUM) -------------------------- */
int smi_reduce_max_abs_sum(smi_ctx* ctx, const void* vector_field, double* out);
int smi_reduce_max(smi_ctx* ctx, const void* field, int ncomp, double* out);
/* max over the interior of |u_x| + |u_y| + |u_z| (compute_stable_timestep, flow_simulators_mpi_3d.py:426-449)
 * of the velocity array `velocity` as the last smi_flow_step_ns3d on ctx wrote it, gathered by that step's curl
 * kernel: no pass over the field.  Returns an error when no such value is at hand or when that step wrote another
 * array; it does not see later writes to the field. */
int smi_step_velocity_max_abs_sum(smi_ctx* ctx, const void* velocity, double* out);
int smi_reduce_sum_sq(smi_ctx* ctx, const void* field, int ncomp, double* out);

/* ---- unbounded Poisson solver ---------------------------------------------------
 * UnboundedPoissonSolverMPI3D (poisson_solver_3d/UnboundedPoissonSolverMPI3D.py:22-187),
 * FFTMPI3D (fft_mpi_3d.py:5-48) and MPIDomainDoublingCommunicator3D (:190-382).
 * dx: the real_t grid spacing; x_range: the user's x extent (the Green's function uses
 * 2 * x_range, :102-106); global_n*: GLOBAL interior sizes. */
int smi_poisson_create(smi_poisson** out, smi_ctx* ctx, int global_nz, int global_ny, int global_nx,
                       double dx, double x_range, int backend);
/* 1 when smi_poisson_create would accept this global grid on `nranks` slabs with this back end, else 0: the
 * rocFFT back end takes any grid whose planes divide over the ranks (one rank: the doubled grid whole, as the
 * reference's solver forms it; z-slabs: a divided transform - rows where they live, an all-to-all to spectral
 * columns, doubled (2nz, 2ny) planes per rank - in place of mpi4py-fft's, fft_mpi_3d.py:27-48); the pruned one
 * needs nx, ny and nz to be powers of two from 8 to 2048 or one of 192, 320, 384, 640, 768.  No counterpart in
 * the reference (mpi4py-fft plans any size). */
int smi_poisson_supports(int global_nz, int global_ny, int global_nx, int nranks, int backend);
int smi_poisson_destroy(smi_poisson* p);
int smi_poisson_solve(smi_poisson* p, void* solution, const void* rhs);
int smi_poisson_vector_solve(smi_poisson* p, void* solution, const void* rhs);
/* A pencil decomposition solves through the z-slab decomposition of the same grid, as the reference's solver moves
 * its fields between the grid's and the transform's decomposition (UnboundedPoissonSolverMPI3D.py:190-382).
 * smi_poisson_set_pencil_route names, for a solver created on the slab context, the pencil context of the same rank,
 * two three-component slab arrays (right-hand side, solution) and two work arrays of 3 * (pencil interior cells)
 * elements; they stay the caller's, and a null `pencil` withdraws the route.  smi_poisson_pencil_solve then solves
 * for ncomp = 1 or 3 components of pencil fields (smi_pencil_to_slab, the solve, smi_slab_to_pencil, ordered by
 * events between the two contexts' streams), and smi_flow_step_ns3d accepts the pencil context with that solver. */
int smi_poisson_set_pencil_route(smi_poisson* p, smi_ctx* pencil, void* slab_rhs, void* slab_sol, void* work_a,
                                 void* work_b);
int smi_poisson_pencil_solve(smi_poisson* p, void* solution, const void* rhs, int ncomp);
/* smi_poisson_solve for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same kernels and
 * bit-identical results, but the spectral work buffers come from a fixed 64 MiB array inside the library
 * (one per GPU, held until the process ends) instead of the buffers smi_poisson_create allocated; the call
 * itself makes no hipMalloc / hipFree or other allocating call.  Needs a single-rank solver created with
 * SMI_POISSON_PRUNED whose scalar work space fits (up to 128^3 in single, 64^3 in double precision); larger
 * grids are rejected with an error.  Every call on a GPU shares that GPU's array: calls that overlap on one
 * GPU (two contexts, two streams) race.  smi_poisson_noalloc_capacity: the size of the array in bytes. */
int smi_poisson_solve_noalloc(smi_poisson* p, void* solution, const void* rhs);
size_t smi_poisson_noalloc_capacity(void);
/* debug / test: download the compact Green's-function spectrum */
size_t smi_poisson_greens_size(smi_poisson* p);
int smi_poisson_greens_download(smi_poisson* p, void* host);

/* ---- stand-alone FFT plans and doubled-domain copies ------------------------------------------
 * FFTMPI3D (poisson_solver_3d/fft_mpi_3d.py:5-48) / FFTMPI2D (poisson_solver_2d/fft_mpi_2d.py:5-45):
 * real (nz, ny, nx) <-> complex (nz, ny, nx/2 + 1), contiguous, no ghosts; nz = 1 makes a 2D plan.
 * forward is unnormalised, backward divides by nz*ny*nx and leaves its input intact.
 * nz, ny, nx are the GLOBAL sizes.  On a context of P z-slabs (3D) the arrays are mpi4py-fft's for a slab
 * decomposition: real (nz / P, ny, nx) per rank, spectrum (nz, ny / P, nx/2 + 1) per rank; P must divide nz and ny. */
typedef struct smi_fft smi_fft;
int smi_fft_create(smi_fft** out, smi_ctx* ctx, int nz, int ny, int nx);
int smi_fft_destroy(smi_fft* f);
int smi_fft_forward(smi_fft* f, const void* field, void* fourier_field);
int smi_fft_backward(smi_fft* f, const void* fourier_field, void* inv_fourier_field);
/* MPIDomainDoublingCommunicator3D.copy_to_doubled_domain / copy_from_doubled_domain
 * (UnboundedPoissonSolverMPI3D.py:314-382; 2D: UnboundedPoissonSolverMPI2D.py:273-352): interior of
 * the ghosted local field <-> first octant (quadrant) of the contiguous doubled array.  On P z-slabs (P even)
 * local_doubled_field is this rank's (2 nz_local, 2 ny, 2 nx) slab of the doubled grid: the blocks of ranks 2r and
 * 2r + 1 arrive on rank r < P / 2, the other ranks hold padding only. */
int smi_copy_to_doubled_domain(smi_ctx* ctx, const void* local_field, void* local_doubled_field);
int smi_copy_from_doubled_domain(smi_ctx* ctx, const void* local_doubled_field, void* local_field);

/* ---- immersed boundary ------------------------------------------------------------
 * EulerianLagrangianGridCommunicatorMPI3D.py:116-589 and VirtualBoundaryForcingMPI.py:278-459.
 * Lagrangian arrays are DEVICE (3, M) float64; index is (3, M) int64 ordered x, y, z;
 * support is (3,4,4,4,M) float64, weights (4,4,4,M) float64.  substart: global index of
 * this slab's first interior cell in x, y, z order; coord_shift: eul_grid_coord_shift. */
int smi_ib_support(smi_ctx* ctx, void* support, void* nearest_index, const void* lag_positions,
                   int64_t m, double dx, double coord_shift, const int64_t* substart_xyz);
int smi_ib_weights(smi_ctx* ctx, void* weights, void* support, int64_t m, double dx, int kernel_type);
int smi_ib_interpolate(smi_ctx* ctx, void* lag_field, const void* eul_field, const void* weights,
                       const void* nearest_index, int64_t m, int ncomp, double dx);
int smi_ib_spread(smi_ctx* ctx, void* eul_field, const void* lag_field, const void* weights,
                  const void* nearest_index, int64_t m, int ncomp);
/* Choice of the scatter-add in smi_ib_spread / smi_ib_spread_fused on this context.  off (default): one wave per
 * node, one floating-point atomic add per window cell; overlapping windows accumulate in whatever order the hardware
 * serves them, so the last bits differ from run to run.  on: the nodes are added one after the other in index order,
 * each with a single rounding to the field type, as the reference's serial loop over nodes does
 * (EulerianLagrangianGridCommunicatorMPI3D.py:404-425): reproducible and in the reference's order, a few microseconds
 * per node; meant for parity runs. */
int smi_set_ordered_spread(smi_ctx* ctx, int on);
/* MPIGhostSumCommunicator3D.ghost_sum (EulerianLagrangianGridCommunicatorMPI3D.py:677-792) */
int smi_ghost_sum(smi_ctx* ctx, void* field, int ncomp);
/* MPIGhostSumCommunicator3D.clear_ghost_cells (:786-792) */
int smi_clear_ghost_cells(smi_ctx* ctx, void* field, int ncomp);
/* Fused steps 1-5 of compute_interaction_force_on_lag_grid (VirtualBoundaryForcingMPI.py:360-397):
 * index + weights + E->L velocity + mismatch + penalty force, no support/weights arrays. */
int smi_ib_forcing_fused(smi_ctx* ctx, void* lag_forcing, void* lag_velocity_mismatch,
                         void* lag_flow_velocity, const void* lag_position_mismatch,
                         const void* lag_positions, const void* lag_velocities,
                         const void* eul_velocity, int64_t m, double dx, double coord_shift,
                         const int64_t* substart_xyz, double stiffness, double damping,
                         int kernel_type);
/* Fused L->E spread that recomputes the weights from positions (step 2 of :423-429). */
int smi_ib_spread_fused(smi_ctx* ctx, void* eul_forcing, const void* lag_forcing,
                        const void* lag_positions, int64_t m, double dx, double coord_shift,
                        const int64_t* substart_xyz, int kernel_type);
/* update_lag_grid_position_mismatch_field_via_euler_forward (:297-310) */
int smi_ib_mismatch_step(smi_ctx* ctx, void* position_mismatch, const void* velocity_mismatch,
                         int64_t n, double dt);
/* VirtualBoundaryForcingMPI.py:278-331 (compute_lag_grid_velocity_mismatch_field,
 * compute_lag_grid_forcing_field): velocity_mismatch = flow_velocity - body_velocity,
 * lag_forcing = stiffness * position_mismatch + damping * velocity_mismatch, on n = dim * nodes
 * device-resident doubles.  The step-by-step (2D) interaction uses it; the fused 3D kernel has
 * it built in. */
int smi_ib_penalty_force(smi_ctx* ctx, void* lag_forcing, void* velocity_mismatch, const void* flow_velocity,
                         const void* body_velocity, const void* position_mismatch, int64_t n,
                         double stiffness, double damping);

/* The same three kernels one at a time, as the reference exposes them (static methods of
 * VirtualBoundaryForcingMPI, :278-331): out = a x + b y on n device-resident doubles (out may be x or y). */
int smi_ib_axpby(smi_ctx* ctx, void* out, double a, const void* x, double b, const void* y, int64_t n);

/* Forcing grids of framed bodies formed on the device (replaces the host arithmetic of sopht's
 * SphereForcingGrid / CosseratRodSurfaceForcingGrid compute_lag_grid_position_field,
 * compute_lag_grid_velocity_field and transfer_forcing_from_grid_to_body; call sites
 * immersed_body_flow_interaction_mpi.py:138-144, 171-175).  A node belongs to one frame (the sphere;
 * a rod element): position = centre + Q^T (scale * local offset), velocity = v + (Q^T omega) x arm.
 * frames: n_frames records of 19 doubles on the device - centre (3), directors row by row (9),
 * velocity (3), angular velocity in the material frame (3), scale (1).  local_offsets, positions,
 * velocities, arms: (3, m) doubles; frame_of_node: m int32. */
int smi_ib_frame_nodes(smi_ctx* ctx, void* positions, void* velocities, void* arms, const void* local_offsets,
                       const void* frame_of_node, const void* frames, int n_frames, int64_t m);
/* loads[6 e .. 6 e + 5] = sum over the nodes of frame e of (force, arm x force), lab frame; the nodes of
 * a frame are contiguous, frame_start holds n_frames + 1 int32 offsets.  Reproducible (fixed order). */
int smi_ib_frame_loads(smi_ctx* ctx, void* loads, const void* lag_forcing, const void* arms,
                       const void* frame_start, int n_frames, int64_t m);

/* ---- fused flow step ---------------------------------------------------------------
 * One navier_stokes / navier_stokes_with_forcing step of UnboundedFlowSimulator3D
 * (flow_simulators_mpi_3d.py:395-424) with fused kernels.  forcing may be NULL
 * (flow_type "navier_stokes").  free_stream: 3 doubles or NULL.  filter_order 0 = off. */
int smi_flow_step_ns3d(smi_ctx* ctx, smi_poisson* p, void* vorticity, void* velocity,
                       void* stream_func, void* buffer_vector, void* buffer_scalar, void* forcing,
                       double dt, double nu, double dx, int penalty_width, const double* fac_lo,
                       const double* fac_hi, const double* free_stream, int filter_order,
                       int filter_type);
/* The same step for a caller that can exchange the roles of `vorticity` and `buffer_vector` afterwards (the two
 * arrays of flow_simulators_mpi_3d.py:191-194; the reference treats buffer_vector as scratch).  Advection and
 * diffusion (:395-411) then run as ONE pass from `vorticity` into `buffer_vector` and the rest of the step works on
 * `buffer_vector`.  *result_in_buffer = 1: the new vorticity is in `buffer_vector`, `vorticity` holds scratch;
 * 0: the result is in `vorticity`, as with smi_flow_step_ns3d (unaligned rows, one ghost layer across slabs, filter
 * settings that need the buffer).  z-slabs are served like a single rank. */
int smi_flow_step_ns3d_swap(smi_ctx* ctx, smi_poisson* p, void* vorticity, void* velocity, void* stream_func,
                            void* buffer_vector, void* buffer_scalar, void* forcing, double dt, double nu, double dx,
                            int penalty_width, const double* fac_lo, const double* fac_hi, const double* free_stream,
                            int filter_order, int filter_type, int* result_in_buffer);
/* Optional promise for the next smi_flow_step_ns3d call on ctx (single rank): the forcing field is zero
 * outside the padded-index box [lo, hi] (z, y, x order, inclusive; lo > hi on an axis = zero everywhere;
 * NULL, NULL withdraws it).  The step then runs update_vorticity_from_velocity_forcing
 * (flow_simulators_mpi_3d.py:415-421) and the forcing reset (:422-424) only where that box can reach; the
 * results are those of the full-array operators.  No counterpart in the reference, whose forcing update
 * always sweeps the whole grid. */
int smi_set_forcing_support(smi_ctx* ctx, const int* lo_zyx, const int* hi_zyx);
/* The promise checked: number of non-zero cells of `forcing` outside the box promised on ctx (0 without a promise).
 * With the context option "check_forcing_support" (SMI_CHECK_FORCING_SUPPORT) the step runs this check itself and
 * fails instead of using a box that does not hold - for callers that write the forcing field through raw pointers
 * and want their bookkeeping verified.  No counterpart in the reference. */
int smi_check_forcing_support(smi_ctx* ctx, const void* forcing, long long* outside_nonzero);

/* ---- 2D flow path --------------------------------------------------------------------
 * Operators of UnboundedFlowSimulator2D (simulator/flow/flow_simulators_mpi_2d.py:166-298) on
 * ghosted [y][x] arrays of (ny + 2g) * (nx + 2g) reals; vectors are (2, y, x) with component
 * 0 = x.  ctx must have been created with nz = 1; its ny, nx and ghost describe the rows of this
 * rank (utils/mpi_utils_2d.py:33-39: the ranks divide y, x stays whole).  Operators that read neighbours
 * exchange the ghost rows with the adjacent ranks first, as the reference's wrappers do. */
typedef struct smi_poisson2d smi_poisson2d;
/* stencil_ops_2d/diffusion_flux_mpi_2d.py:30-136 */
int smi2d_diffusion_flux(smi_ctx* ctx, void* flux, const void* field, double prefactor);
/* stencil_ops_2d/diffusion_timestep_mpi_2d.py */
int smi2d_diffusion_timestep(smi_ctx* ctx, void* field, void* flux, double nu_dt_by_dx2);
/* stencil_ops_2d/advection_timestep_mpi_2d.py, advection_flux_mpi_2d.py (kernel support 2) */
int smi2d_advection_timestep_eno3(smi_ctx* ctx, void* field, void* flux, const void* velocity,
                                  double dt_by_dx);
/* stencil_ops_2d/advection_flux_mpi_2d.py: the flux alone (cells outside the wrapper's boxes keep their values) */
int smi2d_advection_flux_eno3(smi_ctx* ctx, void* flux, const void* field, const void* velocity, double inv_dx);
/* stencil_ops_2d/outplane_field_curl_mpi_2d.py:30-139 (stream function -> velocity) */
int smi2d_outplane_field_curl(smi_ctx* ctx, void* curl, const void* field, double prefactor);
/* stencil_ops_2d/update_vorticity_from_velocity_forcing_mpi_2d.py */
int smi2d_update_vorticity_from_velocity_forcing(smi_ctx* ctx, void* vorticity,
                                                 const void* forcing, double prefactor);
/* stencil_ops_2d/penalise_field_boundary_mpi_2d.py:9-172; factor tables [x band | y band] */
int smi2d_penalise_boundary(smi_ctx* ctx, void* field, int width, const double* fac_lo,
                            const double* fac_hi);
/* sopht set_fixed_val / add_fixed_val as used by flow_simulators_mpi_2d.py:217-233 */
int smi2d_set_fixed_val(smi_ctx* ctx, void* field, int ncomp, const double* vals);
int smi2d_add_fixed_val(smi_ctx* ctx, void* sum_field, const void* field, int ncomp,
                        const double* vals);
/* interior max of |u| + |v| and of a scalar (flow_simulators_mpi_2d.py:300-328) */
int smi2d_reduce_max_abs_sum(smi_ctx* ctx, const void* vector_field, double* out);
int smi2d_reduce_max(smi_ctx* ctx, const void* field, double* out);
/* poisson_solver_2d/UnboundedPoissonSolverMPI2D.py:20-153.  greens_host: the (2ny, 2nx)
 * Green's function field of :61-107 in the context's real type, built by the caller. */
int smi2d_poisson_create(smi_poisson2d** out, smi_ctx* ctx, const void* greens_host, double dx);
int smi2d_poisson_destroy(smi_poisson2d* p);
int smi2d_poisson_solve(smi_poisson2d* p, void* solution, const void* rhs);
/* MPIGhostCommunicator2D.exchange_scalar_field_init / exchange_vector_field_init + exchange_finalise
 * (utils/mpi_utils_2d.py:286-420) for row slabs: ghost rows <- the neighbours' edge rows, stream ordered */
int smi2d_halo_exchange(smi_ctx* ctx, void* field, int ncomp);
/* UnboundedFlowSimulator2D.navier_stokes_timestep / navier_stokes_with_forcing_timestep,
 * flow_simulators_mpi_2d.py:260-298, as one call: forcing may be NULL (no forcing update), free_stream
 * NULL (no free stream), fac_lo / fac_hi as for smi2d_penalise_boundary (ignored when the width is 0). */
int smi2d_flow_step_ns(smi_ctx* ctx, smi_poisson2d* p, void* vorticity, void* velocity, void* stream_func,
                       void* buffer_scalar, void* forcing, double dt, double nu, double dx, int penalty_width,
                       const double* fac_lo, const double* fac_hi, const double* free_stream);
/* immersed_boundary_ops/EulerianLagrangianGridCommunicatorMPI2D.py:116-520: support (2,4,4,M) and
 * nearest index (2,M) of the nodes, delta-function weights (4,4,M), Eulerian -> Lagrangian
 * interpolation, Lagrangian -> Eulerian spreading (ghost_sum != 0: followed by the ghost sum of
 * :521-652: ghost rows are added to the neighbouring ranks' edge rows, then all ghost cells are cleared). */
int smi2d_ib_support(smi_ctx* ctx, void* support, void* nearest_index, const void* lag_positions,
                     int64_t m, double dx, double coord_shift);
int smi2d_ib_weights(smi_ctx* ctx, void* weights, void* support, int64_t m, double dx,
                     int kernel_type);
int smi2d_ib_interpolate(smi_ctx* ctx, void* lag_field, const void* eul_field, const void* weights,
                         const void* nearest_index, int64_t m, int ncomp, double dx);
int smi2d_ib_spread(smi_ctx* ctx, void* eul_field, const void* lag_field, const void* weights,
                    const void* nearest_index, int64_t m, int ncomp, int ghost_sum);
/* MPIGhostSumCommunicator2D.ghost_sum (clear_only = 0) and clear_ghost_cells (clear_only != 0) on their own
 * (EulerianLagrangianGridCommunicatorMPI2D.py:563-652) */
int smi2d_ghost_sum(smi_ctx* ctx, void* field, int ncomp, int clear_only);

#ifdef __cplusplus
}
#endif
#endif /* SOPHTMI_H */
