// Internal definitions shared by the libsophtmi translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "sophtmi.h"

struct SmiProf;

struct smi_ctx {
  int device;
  int dtype;           // SMI_F32 / SMI_F64
  int nz, ny, nx, g;   // local interior sizes, ghost layers
  int pz, py, px;      // padded sizes
  size_t plane;        // py * px
  size_t vol;          // pz * py * px
  int rank, nranks;
  int phys_zlo, phys_zhi;  // is the low / high z face a physical boundary
  // Pencil decomposition (smi_ctx_set_pencil): the ranks form a (nranks_z, nranks_y) grid, rank = rank_z * nranks_y +
  // rank_y (the Cartesian order of the reference's MPI.Compute_dims / Create_cart topology); x is never divided.
  // z-slabs are nranks_y = 1.
  int rank_z, nranks_z, rank_y, nranks_y;
  int phys_ylo, phys_yhi;
  // smi_ctx_set_grid3: x divided as well, rank = (rank_z * nranks_y + rank_y) * nranks_x + rank_x.  The stencil
  // operators, the exchanges and the immersed-boundary kernels serve it; the transforms do not (a Poisson solve is
  // routed through the z-slab decomposition, the one-pass kernels fall back where rows are no whole vectors).
  int rank_x, nranks_x;
  int phys_xlo, phys_xhi;
  // y faces are strided in memory: they are packed into / unpacked from these device buffers
  // ([0] to the previous y rank, [1] to the next, [2] from the previous, [3] from the next; 3 components of g rows
  // of every plane each), allocated by smi_ctx_set_pencil
  void* ybuf[4];
  size_t ybuf_cap;
  smi_sendrecv_cb sendrecv_cb;
  smi_group_alltoall_cb group_alltoall_cb;
  smi_p2p_cb p2p_cb;
  // General block topology (smi_ctx_set_block): the ranks form a (Pz, Py, Px) grid in Cartesian order, any axis may
  // be divided and any axis periodic.  Only the ghost exchange (smi_halo_exchange_general) serves such a context.
  int blk_on;
  int blk_coords[3], blk_topo[3], blk_periodic[3];
  void* blk_buf[4];      // packed faces: to the low / high neighbour, from the low / high neighbour
  size_t blk_cap;
  void* dbl_stage[3];    // staging blocks of the doubled-domain copies across ranks (smi_fftapi.hip)
  size_t dbl_stage_cap;
  hipStream_t stream;
  hipStream_t comm_stream;
  hipEvent_t ev_a, ev_b;
  hipEvent_t ev_order;  // smi_ctx_wait_for: what another context's stream waits for
  void* red_partials;  // device, double[kRedBlocks]
  double* red_host;    // pinned
  ncclComm_t comm;
  bool comm_ready;
  void* halo_recv[2];  // device staging: [0] from previous rank, [1] from next rank
  size_t halo_cap;     // bytes per staging buffer
  void* fac_dev;       // small device scratch (allreduce staging)
  smi_exchange_cb exchange_cb;
  smi_alltoall_cb alltoall_cb;
  SmiProf* prof;
  void* exchange_user;
  int halo_depth;  // 0: exchange all g ghost planes; d < g: only the d planes next to the interior
  // Promise for the next fused step (smi_set_forcing_support): the forcing field is zero outside the padded
  // index box [lo, hi] (z, y, x order, inclusive; lo > hi = zero everywhere).  Consumed by that step.
  int fbox_valid;
  int fbox_lo[3], fbox_hi[3];
  // max over the interior of |u_x| + |u_y| + |u_z| of the velocity the last fused step wrote, gathered by the
  // curl kernel itself: kUmaxSlots bit patterns of non-negative reals (device), valid flag, and the value
  // of the cells outside the curl's compute set (the free stream)
  void* umax_slots;
  int umax_valid;
  double umax_band;
  const void* umax_field;  // the velocity array that maximum belongs to
  int ordered_spread;      // smi_set_ordered_spread: Lagrangian -> Eulerian spreading adds nodes in index order
  // Switches of the one-pass advection + diffusion kernel, read from the environment ONCE when the context is
  // created (SMI_NO_FUSED_ADVECT_DIFFUSE, SMI_FUSED_SPLIT_LAUNCH, SMI_FUSED_ZCHUNK) and changed afterwards only
  // through smi_ctx_set_option: no_fused = run the two-pass form; split_launch = the two launches of the overlapped
  // multi-rank schedule on a single rank; fused_zchunk = planes per workgroup (0: chosen from the grid).
  int opt_no_fused, opt_split_launch, opt_fused_zchunk;
  // z chunks each spectral transpose of a multi-rank Poisson solve is cut into (SMI_FFT_TRANSPOSE_CHUNKS, 1 / 2 / 4)
  int opt_transpose_chunks;
  // SMI_CHECK_FORCING_SUPPORT / "check_forcing_support": the fused step verifies a promised forcing box (every cell
  // outside it must be zero) before it relies on it, and fails if it does not hold; a debugging aid for callers of
  // smi_set_forcing_support, one extra sweep of the forcing field per step
  int opt_check_forcing_support;
};
constexpr int kUmaxSlots = 64;
// the context's pinned host words (smi_ctx::red_host): 8 doubles for the reductions, kUmaxSlots words for the step's
// velocity maximum, and from kPinnedStageOffset on a staging area for small host transfers
constexpr size_t kPinnedStageOffset = 1024, kPinnedStageBytes = 7168, kPinnedBytes = kPinnedStageOffset + kPinnedStageBytes;
static_assert(sizeof(double) * (8 + kUmaxSlots) <= kPinnedStageOffset, "pinned words overlap");

constexpr int kRedBlocks = 2048;

void smi_set_error(const char* fmt, ...);

#define SMI_CHECK_HIP(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      smi_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)

#define SMI_CHECK_NCCL(expr)                                                              \
  do {                                                                                    \
    ncclResult_t _e = (expr);                                                             \
    if (_e != ncclSuccess) {                                                              \
      smi_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, ncclGetErrorString(_e)); \
      return 1;                                                                           \
    }                                                                                     \
  } while (0)

#define SMI_REQUIRE(cond, ...)    \
  do {                            \
    if (!(cond)) {                \
      smi_set_error(__VA_ARGS__); \
      return 1;                   \
    }                             \
  } while (0)

#define SMI_CHECK(expr)      \
  do {                       \
    int _r = (expr);         \
    if (_r != 0) return _r;  \
  } while (0)

#define SMI_LAUNCH_CHECK() SMI_CHECK_HIP(hipGetLastError())

// Run `body` with T bound to the context's real type.
#define SMI_DISPATCH(ctx, ...)      \
  do {                              \
    if ((ctx)->dtype == SMI_F32) {  \
      using T = float;              \
      __VA_ARGS__                   \
    } else {                        \
      using T = double;             \
      __VA_ARGS__                   \
    }                               \
  } while (0)

static inline size_t smi_elem_size(const smi_ctx* c) { return c->dtype == SMI_F32 ? 4 : 8; }

template <typename T>
struct alignas(sizeof(T) * 4) Vec4 {
  T v[4];
};

// 16-byte (f32) / 32-byte (f64) vector access of four consecutive reals; p must be
// aligned to 4 elements.
template <typename T>
__device__ __forceinline__ Vec4<T> load4(const T* p) {
  return *reinterpret_cast<const Vec4<T>*>(p);
}
template <typename T>
__device__ __forceinline__ void store4(T* p, const Vec4<T>& v) {
  *reinterpret_cast<Vec4<T>*>(p) = v;
}

// internal cross-TU helpers
int smi_halo_exchange_impl(smi_ctx* ctx, void* field, int ncomp);
int smi_halo_exchange2_impl(smi_ctx* ctx, void* field_a, int ncomp_a, void* field_b, int ncomp_b);
// the same exchange queued on `stream` (RCCL transport only: the host-staged callback works on the context's stream)
int smi_halo_exchange2_on(smi_ctx* ctx, void* field_a, int ncomp_a, void* field_b, int ncomp_b, hipStream_t stream);
int smi_y_face_exchange(smi_ctx* ctx, void* field, int ncomp, int depth, bool sum, hipStream_t stream);
// A few equal-sized messages between device buffers of the ranks (smi_fftapi.hip): sends[i] goes to rank
// send_peers[i], recvs[j] comes from rank recv_peers[j]; one RCCL group or the point-to-point callback.
int smi_p2p_exchange(smi_ctx* c, int nsend, const int* send_peers, const void* const* sends, int nrecv,
                     const int* recv_peers, void* const* recvs, size_t bytes);
int smi_diffusion_apply_impl(smi_ctx* c, void* out, void* in, double prefactor, int ncomp);
int smi_filter_apply_impl(smi_ctx* c, void* out, void* in, int ncomp, const double* penx, int* pen_done);
int smi_advect_vorticity_impl(smi_ctx* c, void* out, void* velocity, void* vorticity, double prefactor);
int smi_advect_diffuse_impl(smi_ctx* c, void* out, void* velocity, void* vorticity, double pref_curl, double pref_diff,
                            int* ran, const double* penx);
int smi_penalise_boundary_impl(smi_ctx* c, void* field, int ncomp, int width, const double* fac_lo,
                               const double* fac_hi, int skip_x);
int smi_curl_plus_const_impl(smi_ctx* c, void* curl, void* field, double prefactor, const double* add,
                             bool track_max = false);
// forcing update and forcing reset restricted to the rows and planes the support box can reach
int smi_forcing_update_boxed_impl(smi_ctx* c, void* vorticity, void* forcing, double prefactor);
int smi_forcing_zero_boxed_impl(smi_ctx* c, void* forcing);
// looks up, once per device, the device address of the fixed scratch array of smi_poisson_solve_noalloc
int smi_pruned_scratch_init(int device);
// true when this context has neighbours to exchange with
static inline bool smi_multi_rank(const smi_ctx* c) { return c->nranks > 1; }
