// 2D flow path of libsophtmi (gfx950): the operators of UnboundedFlowSimulator2D
// (sopht_mpi/simulator/flow/flow_simulators_mpi_2d.py:166-298) on ghosted [y][x] arrays.
//
// 2D grids are tiny next to the 3D ones (256^2 in the reference's own case), so these
// kernels are plain one-lane-per-cell kernels: coalesced along x, neighbours through L1/L2.
// The context must have been created with nz = 1; its ny, nx and ghost width describe the
// plane of this rank, arrays are (ny + 2g) * (nx + 2g) reals owned by the caller.  Several ranks
// divide the rows (the reference's default rank_distribution of MPIConstruct2D, mpi_utils_2d.py:33-39):
// whole padded rows are contiguous, so ghost rows travel as they lie, and the unbounded Poisson
// solve gathers the right-hand side on every rank and solves the whole plane there (a 2D plane is
// a few MB: one all-gather of it costs less than the two transposes of a divided transform).
//
// Which cells an operator writes follows the five overlapping boxes of the reference
// wrappers (diffusion_flux_mpi_2d.py:47-113) and their band zeroing (:116-136), exactly as in
// 3D: with core_s(a) = a in [g+s, p-g-s-1], face_s(a) = a in [g, g+s-1] or [p-g-s, p-g-1],
// ext_s(a) = a in [s, p-s-1]:
//   D   = { core_1(y), core_1(x) }                                             flux, curl
//   U_s = { y in [g, p-g-1], core_s(x) } + { ext_s(y), face_s(x) }             forcing update (s = 1),
//                                                                              advection (s = 2)
#include <hipfft/hipfft.h>

#include "smi_common.h"
#include "smi_prof.h"

#define SMI_CHECK_FFT2(expr)                                                              \
  do {                                                                                    \
    hipfftResult _e = (expr);                                                             \
    if (_e != HIPFFT_SUCCESS) {                                                           \
      smi_set_error("%s:%d: %s -> hipfft error %d", __FILE__, __LINE__, #expr, (int)_e);  \
      return 1;                                                                           \
    }                                                                                     \
  } while (0)

// smi_comm.hip: block r of `send` goes to rank r, block s of `recv` comes from rank s
int smi_alltoall_impl(smi_ctx* c, const void* send, void* recv, size_t bytes_per_peer, hipStream_t stream);

struct smi_poisson2d {
  smi_ctx* ctx;
  int ny, nx;       // the whole plane (ny = rows of all ranks)
  int ny_local;     // rows of this rank, starting at row0 of the plane
  int row0;
  // Several ranks: the transform is divided, rows <-> spectral columns (smi2d_poisson_solve).  KL columns of the
  // nx + 1 (padded to KL * nranks) per rank; row_fwd / row_inv: batched 1D transforms of this rank's rows (2nx
  // points); col: batched strided 1D transform of its columns (2ny points), both directions.
  int KL;
  hipfftHandle row_fwd, row_inv, col;
  void* row_real;   // (ny_local, 2nx) reals
  void* row_spec;   // (ny_local, nx + 1) complex
  void* xchg;       // (nranks, ny_local, KL) complex: blocks by peer
  void* col_spec;   // (2ny, KL) complex
  void* ghat_cols;  // (2ny, KL) complex: this rank's columns of ghat
  hipfftHandle fwd, inv;
  // The captured step of smi2d_flow_step_ns (single rank): executable graph, what it was captured for, and the
  // device array its kernels read the per-step coefficients from
  hipGraph_t step_graph;
  hipGraphExec_t step_exec;
  int step_state;          // 0: not captured, 1: ready, -1: capture failed once, eager from then on
  int eager_steps;         // steps run launch by launch so far (the first ones always are: lazy set-up of the plans)
  const void* step_key[5];
  int step_width, step_has_fs;
  double step_fac[2][32];
  double* step_coef;
  void* real_buf;   // (2ny, 2nx) reals
  void* cplx_buf;   // (2ny, nx + 1) complex
  void* ghat;       // (2ny, nx + 1) complex, times dx^2 / (4 ny nx)
};

namespace {

constexpr int kB = 256;

struct G2 {
  int py, px, g;
  long long n;  // py * px
  int ylo, yhi;  // is the low / high y face a physical boundary (false next to another rank's slab of rows)
};

G2 geo2(const smi_ctx* c) { return G2{c->py, c->px, c->g, (long long)c->py * c->px, c->phys_ylo, c->phys_yhi}; }
int nblocks(long long n) { return (int)((n + kB - 1) / kB); }

__device__ __forceinline__ bool core_s(int a, int p, int g, int s) { return a >= g + s && a <= p - g - s - 1; }
__device__ __forceinline__ bool face_s(int a, int p, int g, int s) {
  return (a >= g && a <= g + s - 1) || (a >= p - g - s && a <= p - g - 1);
}
__device__ __forceinline__ bool ext_s(int a, int p, int s) { return a >= s && a <= p - s - 1; }
// rows of the zero-banded compute set: s rows are left out next to a physical face, none next to another rank
__device__ __forceinline__ bool core_y(const G2& G, int y, int s) {
  return y >= G.g + (G.ylo ? s : 0) && y <= G.py - G.g - 1 - (G.yhi ? s : 0);
}
__device__ __forceinline__ bool in_update_set(int y, int x, const G2& G, int s) {
  return (y >= G.g && y <= G.py - G.g - 1 && core_s(x, G.px, G.g, s)) ||
         (ext_s(y, G.py, s) && face_s(x, G.px, G.g, s));
}

// A coefficient of a step: a value, or - in the replays of a captured step, whose kernel arguments are frozen - a
// slot of a small device array that the host refreshes before every replay (the time step changes from step to step).
template <typename T>
struct Coef {
  T value;
  const double* slot;
  __device__ __forceinline__ T get() const { return slot ? (T)*slot : value; }
};
template <typename T>
Coef<T> coef_of(double v) {
  return Coef<T>{(T)v, nullptr};
}

// flux = c (f(x+1) + f(x-1) + f(y+1) + f(y-1) - 4 f) on D, 0 elsewhere; ADD: out = f + flux
template <typename T, bool ADD>
__global__ void __launch_bounds__(kB) k2_diffusion(T* __restrict__ out, const T* __restrict__ f, Coef<T> cc, G2 G) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G.n) return;
  const T c = cc.get();
  const int y = (int)(i / G.px), x = (int)(i - (long long)y * G.px);
  T v = T(0);
  if (core_y(G, y, 1) && core_s(x, G.px, G.g, 1))
    v = c * (f[i + 1] + f[i - 1] + f[i + G.px] + f[i - G.px] - T(4) * f[i]);
  out[i] = ADD ? f[i] + v : v;
}

template <typename T>
__device__ __forceinline__ T upwind3_face2(const T* f, const T* v, long long stride) {
  const T vf = T(0.5) * (v[0] + v[stride]);
  const T left = T(-1.0 / 6) * f[-stride] + T(5.0 / 6) * f[0] + T(1.0 / 3) * f[stride];
  const T right = T(1.0 / 3) * f[0] + T(5.0 / 6) * f[stride] + T(-1.0 / 6) * f[2 * stride];
  return vf * (vf > T(0) ? left : right);
}

// flux = c * (dF_x/dx + dF_y/dy) with the third-order upwind face flux, on U_2, 0 elsewhere
// KEEP: cells outside U_2 keep their value (the stand-alone flux operator)
template <typename T, bool KEEP = false>
__global__ void __launch_bounds__(kB) k2_advection_flux(T* __restrict__ flux, const T* __restrict__ f,
                                                       const T* __restrict__ vel, Coef<T> cc, G2 G) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G.n) return;
  const T c = cc.get();
  const int y = (int)(i / G.px), x = (int)(i - (long long)y * G.px);
  T val = T(0);
  const bool in_u = in_update_set(y, x, G, 2);
  if (in_u) {
    const T* fc = f + i;
    const T* vx = vel + i;
    const T* vy = vel + G.n + i;
    const T dx = upwind3_face2(fc, vx, 1) - upwind3_face2(fc - 1, vx - 1, 1);
    const T dy = upwind3_face2(fc, vy, (long long)G.px) - upwind3_face2(fc - G.px, vy - G.px, (long long)G.px);
    val = c * (dx + dy);
  }
  if (!KEEP || in_u) flux[i] = val;
}

template <typename T>
__global__ void __launch_bounds__(kB) k2_sum(T* __restrict__ out, const T* __restrict__ a, const T* __restrict__ b,
                                            long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i];
}

// u = p d(psi)/dy, v = -p d(psi)/dx on D, 0 elsewhere (outplane_field_curl_mpi_2d.py); ADD: plus a constant per
// component on every cell (the free stream of flow_simulators_mpi_2d.py:222-233, added in the store instead of in a
// second pass over the velocity)
template <typename T, bool ADD = false>
__global__ void __launch_bounds__(kB) k2_outplane_curl(T* __restrict__ curl, const T* __restrict__ f, Coef<T> pc, G2 G,
                                                      Coef<T> au = Coef<T>{T(0), nullptr},
                                                      Coef<T> av = Coef<T>{T(0), nullptr}) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G.n) return;
  const T p = pc.get(), add_u = au.get(), add_v = av.get();
  const int y = (int)(i / G.px), x = (int)(i - (long long)y * G.px);
  T u = T(0), v = T(0);
  const bool in_d = core_y(G, y, 1) && core_s(x, G.px, G.g, 1);
  if (in_d) {
    u = p * (f[i + G.px] - f[i - G.px]);
    v = -p * (f[i + 1] - f[i - 1]);
  }
  curl[i] = ADD ? (in_d ? u : T(0)) + add_u : u;
  curl[G.n + i] = ADD ? (in_d ? v : T(0)) + add_v : v;
}

// w += p (d f_y / dx - d f_x / dy) on U_1 (update_vorticity_from_velocity_forcing_mpi_2d.py)
template <typename T>
__global__ void __launch_bounds__(kB) k2_update_vorticity(T* __restrict__ w, const T* __restrict__ f, Coef<T> pc, G2 G) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G.n) return;
  const T p = pc.get();
  const int y = (int)(i / G.px), x = (int)(i - (long long)y * G.px);
  if (!in_update_set(y, x, G, 1)) return;
  const T* fx = f + i;
  const T* fy = f + G.n + i;
  w[i] += p * ((fy[1] - fy[-1]) - (fx[G.px] - fx[-G.px]));
}

struct PenFac2 {
  double lo[2][16], hi[2][16];  // [axis x, y][band index]
};

// AXIS 0: one lane per row, AXIS 1: one lane per column (penalise_field_boundary_mpi_2d.py:139-170)
template <typename T, int AXIS>
__global__ void __launch_bounds__(kB) k2_penalise(T* __restrict__ f, G2 G, int w, PenFac2 fac, int do_lo, int do_hi) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  const int lines = AXIS == 0 ? G.py : G.px;
  if (l >= lines) return;
  T* base = AXIS == 0 ? f + (long long)l * G.px : f + l;
  const long long stride = AXIS == 0 ? 1 : G.px;
  const int p = AXIS == 0 ? G.px : G.py;
  if (do_lo) {
    const T lo_src = base[(w - 1) * stride];
    for (int k = 0; k < w; ++k) base[k * stride] = lo_src * (T)fac.lo[AXIS][k];
  }
  if (do_hi) {
    const T hi_src = base[(p - w) * stride];
    for (int k = 0; k < w; ++k) base[(p - w + k) * stride] = hi_src * (T)fac.hi[AXIS][k];
  }
}

// out = in + val (add) or val, over ncomp components of n cells each with one constant per component (at most 2)
template <typename T>
__global__ void __launch_bounds__(kB) k2_set_add(T* __restrict__ out, const T* __restrict__ in, T val0, T val1,
                                                long long n, long long total, int add) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const T val = i < n ? val0 : val1;
  out[i] = add ? in[i] + val : val;
}

// Reductions over the interior in two launches: up to kRed2Blocks partial maxima, then one block over those.
// MODE 0: max of |u| + |v|; MODE 1: max of f.
constexpr int kRed2Blocks = 128;
template <typename T, int MODE>
__global__ void __launch_bounds__(256) k2_reduce_partial(const T* __restrict__ f, G2 G, double* __restrict__ partials) {
  __shared__ double part[256];
  const int ny = G.py - 2 * G.g, nx = G.px - 2 * G.g;
  const long long cells = (long long)ny * nx;
  double m = -1.0e300;
  for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < cells; k += (long long)gridDim.x * blockDim.x) {
    const int y = (int)(k / nx) + G.g, x = (int)(k % nx) + G.g;
    const long long i = (long long)y * G.px + x;
    const T v = MODE == 0 ? (T)(fabs(f[i]) + fabs(f[G.n + i])) : f[i];
    m = fmax(m, (double)v);
  }
  part[threadIdx.x] = m;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) part[threadIdx.x] = fmax(part[threadIdx.x], part[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = part[0];
}
__global__ void __launch_bounds__(kRed2Blocks) k2_reduce_final(const double* __restrict__ partials, int n,
                                                              double* __restrict__ out) {
  __shared__ double part[kRed2Blocks];
  part[threadIdx.x] = (int)threadIdx.x < n ? partials[threadIdx.x] : -1.0e300;
  __syncthreads();
  for (int s = kRed2Blocks / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) part[threadIdx.x] = fmax(part[threadIdx.x], part[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = part[0];
}

// ---- Poisson helpers
template <typename T>
__global__ void __launch_bounds__(kB) k2_pad_load(T* __restrict__ dbl, const T* __restrict__ rhs, int ny, int nx,
                                                 int g, int px) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n = 4LL * ny * nx;
  if (i >= n) return;
  const int y = (int)(i / (2 * nx)), x = (int)(i - (long long)y * 2 * nx);
  dbl[i] = (y < ny && x < nx) ? rhs[(long long)(y + g) * px + x + g] : T(0);
}
template <typename T>
__global__ void __launch_bounds__(kB) k2_cmul(T* __restrict__ a, const T* __restrict__ b, long long ncplx) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ncplx) return;
  const T ar = a[2 * i], ai = a[2 * i + 1], br = b[2 * i], bi = b[2 * i + 1];
  a[2 * i] = ar * br - ai * bi;
  a[2 * i + 1] = ar * bi + ai * br;
}
template <typename T>
__global__ void __launch_bounds__(kB) k2_scale(T* __restrict__ a, T s, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] *= s;
}
// rows row0 .. row0 + ny - 1 of the doubled plane -> the interior of this rank's solution array
template <typename T>
__global__ void __launch_bounds__(kB) k2_extract(T* __restrict__ sol, const T* __restrict__ dbl, int ny, int nx,
                                                int g, int px, int row0) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)ny * nx) return;
  const int y = (int)(i / nx), x = (int)(i - (long long)y * nx);
  sol[(long long)(y + g) * px + x + g] = dbl[(long long)(y + row0) * 2 * nx + x];
}
// ---- divided transform (several ranks)
// this rank's interior rows, zero padded to 2nx
template <typename T>
__global__ void __launch_bounds__(kB) k2_pad_rows(T* __restrict__ rows, const T* __restrict__ rhs, int ny, int nx,
                                                 int g, int px) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2LL * ny * nx) return;
  const int y = (int)(i / (2 * nx)), x = (int)(i - (long long)y * 2 * nx);
  rows[i] = x < nx ? rhs[(long long)(y + g) * px + x + g] : T(0);
}
// row spectra (ny, nx + 1) <-> blocks by peer (P, ny, KL): peer r holds columns r KL .. r KL + KL - 1 (zeros past nx)
template <typename T, bool TO_BLOCKS>
__global__ void __launch_bounds__(kB) k2_col_blocks(T* __restrict__ spec, T* __restrict__ blocks, int ny, int nx,
                                                   int KL, int P) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)P * ny * KL) return;
  const int k = (int)(i % KL);
  const long long t = i / KL;
  const int y = (int)(t % ny), r = (int)(t / ny);
  const int col = r * KL + k;
  if (TO_BLOCKS) {
    const bool in = col <= nx;
    const long long src = (long long)y * (nx + 1) + col;
    blocks[2 * i] = in ? spec[2 * src] : T(0);
    blocks[2 * i + 1] = in ? spec[2 * src + 1] : T(0);
  } else if (col <= nx) {
    const long long dst = (long long)y * (nx + 1) + col;
    spec[2 * dst] = blocks[2 * i];
    spec[2 * dst + 1] = blocks[2 * i + 1];
  }
}
template <typename T>
__global__ void __launch_bounds__(kB) k2_fill_zero(T* __restrict__ a, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = T(0);
}
// columns c0 .. c0 + KL - 1 of the (rows, nx + 1) spectrum -> (rows, KL), zeros past nx
template <typename T>
__global__ void __launch_bounds__(kB) k2_take_cols(T* __restrict__ out, const T* __restrict__ full, int rows, int nx,
                                                  int KL, int c0) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)rows * KL) return;
  const int k = (int)(i % KL), y = (int)(i / KL), col = c0 + k;
  const long long src = (long long)y * (nx + 1) + col;
  out[2 * i] = col <= nx ? full[2 * src] : T(0);
  out[2 * i + 1] = col <= nx ? full[2 * src + 1] : T(0);
}
// interior of a ghosted array -> dense (ny, nx)
template <typename T>
__global__ void __launch_bounds__(kB) k2_pack_interior(T* __restrict__ dense, const T* __restrict__ f, int ny, int nx,
                                                      int g, int px) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)ny * nx) return;
  const int y = (int)(i / nx), x = (int)(i - (long long)y * nx);
  dense[i] = f[(long long)(y + g) * px + x + g];
}

int require_2d(const smi_ctx* c, const char* who) {
  SMI_REQUIRE(c, "%s: null context", who);
  SMI_REQUIRE(c->nz == 1, "%s: the 2D operators need a context created with nz = 1", who);
  return 0;
}

// MPIGhostCommunicator2D (utils/mpi_utils_2d.py:87-421) for slabs of rows: whole padded rows are contiguous, so the
// first / last g interior rows travel as they lie into the neighbour's ghost rows.
int halo2(smi_ctx* c, void* field, int ncomp) {
  if (c->nranks == 1) return 0;
  const size_t es = smi_elem_size(c), row = (size_t)c->px * es, n = (size_t)c->py * c->px * es;
  const size_t bytes = (size_t)c->g * row;
  const bool lo = c->rank_y > 0, hi = c->rank_y < c->nranks_y - 1;
  for (int k = 0; k < ncomp; ++k) {
    char* base = (char*)field + k * n;
    char* send_lo = base + (size_t)c->g * row;
    char* recv_lo = base;
    char* send_hi = base + (size_t)(c->py - 2 * c->g) * row;
    char* recv_hi = base + (size_t)(c->py - c->g) * row;
    if (c->sendrecv_cb) {
      SMI_CHECK_HIP(hipStreamSynchronize(c->stream));
      int rc = c->sendrecv_cb(c->exchange_user, lo ? c->rank - 1 : -1, hi ? c->rank + 1 : -1, send_lo, send_hi, recv_lo,
                              recv_hi, bytes);
      SMI_REQUIRE(rc == 0, "2D halo exchange callback failed (%d)", rc);
    } else {
      SMI_REQUIRE(c->comm_ready, "2D halo exchange across ranks needs smi_comm_init or a send-receive callback");
      SMI_CHECK_NCCL(ncclGroupStart());
      if (lo) {
        SMI_CHECK_NCCL(ncclSend(send_lo, bytes, ncclChar, c->rank - 1, c->comm, c->stream));
        SMI_CHECK_NCCL(ncclRecv(recv_lo, bytes, ncclChar, c->rank - 1, c->comm, c->stream));
      }
      if (hi) {
        SMI_CHECK_NCCL(ncclSend(send_hi, bytes, ncclChar, c->rank + 1, c->comm, c->stream));
        SMI_CHECK_NCCL(ncclRecv(recv_hi, bytes, ncclChar, c->rank + 1, c->comm, c->stream));
      }
      SMI_CHECK_NCCL(ncclGroupEnd());
    }
  }
  return 0;
}

template <typename T>
__global__ void __launch_bounds__(kB) k2_add(T* __restrict__ dst, const T* __restrict__ src, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

// Ghost sum across the row slabs (EulerianLagrangianGridCommunicatorMPI2D.py:523-620): what the spread left in my
// ghost rows belongs to the neighbour's edge rows.  My low ghost rows are added to the previous rank's last g
// interior rows, my high ghost rows to the next rank's first g interior rows.
int ghost_sum2(smi_ctx* c, void* field, int ncomp) {
  if (c->nranks == 1) return 0;
  const size_t es = smi_elem_size(c), row = (size_t)c->px * es, n = (size_t)c->py * c->px * es;
  const size_t bytes = (size_t)c->g * row;
  const long long cells = (long long)c->g * c->px;
  SMI_REQUIRE((size_t)ncomp * bytes <= c->halo_cap, "2D ghost sum: staging buffer too small");
  const bool lo = c->rank_y > 0, hi = c->rank_y < c->nranks_y - 1;
  for (int k = 0; k < ncomp; ++k) {
    char* base = (char*)field + k * n;
    char* send_lo = base;
    char* send_hi = base + (size_t)(c->py - c->g) * row;
    char* recv_lo = (char*)c->halo_recv[0] + k * bytes;
    char* recv_hi = (char*)c->halo_recv[1] + k * bytes;
    if (c->sendrecv_cb) {
      SMI_CHECK_HIP(hipStreamSynchronize(c->stream));
      int rc = c->sendrecv_cb(c->exchange_user, lo ? c->rank - 1 : -1, hi ? c->rank + 1 : -1, send_lo, send_hi, recv_lo,
                              recv_hi, bytes);
      SMI_REQUIRE(rc == 0, "2D ghost sum callback failed (%d)", rc);
    } else {
      SMI_REQUIRE(c->comm_ready, "2D ghost sum across ranks needs smi_comm_init or a send-receive callback");
      SMI_CHECK_NCCL(ncclGroupStart());
      if (lo) {
        SMI_CHECK_NCCL(ncclSend(send_lo, bytes, ncclChar, c->rank - 1, c->comm, c->stream));
        SMI_CHECK_NCCL(ncclRecv(recv_lo, bytes, ncclChar, c->rank - 1, c->comm, c->stream));
      }
      if (hi) {
        SMI_CHECK_NCCL(ncclSend(send_hi, bytes, ncclChar, c->rank + 1, c->comm, c->stream));
        SMI_CHECK_NCCL(ncclRecv(recv_hi, bytes, ncclChar, c->rank + 1, c->comm, c->stream));
      }
      SMI_CHECK_NCCL(ncclGroupEnd());
    }
    SMI_DISPATCH(c, {
      if (lo) k2_add<T><<<nblocks(cells), kB, 0, c->stream>>>((T*)(base + (size_t)c->g * row), (const T*)recv_lo, cells);
      if (hi)
        k2_add<T><<<nblocks(cells), kB, 0, c->stream>>>((T*)(base + (size_t)(c->py - 2 * c->g) * row), (const T*)recv_hi,
                                                        cells);
    });
    SMI_LAUNCH_CHECK();
  }
  return 0;
}

// Every rank's block of `bytes` bytes -> all blocks in rank order on every rank.  `send` holds room for one copy of
// the block per rank: the host-staged transport only knows an all-to-all, which gets the block replicated.
template <typename T>
int fft_forward(smi_poisson2d* p, void* real_in, void* cplx_out) {
  if (sizeof(T) == 4)
    SMI_CHECK_FFT2(hipfftExecR2C(p->fwd, (hipfftReal*)real_in, (hipfftComplex*)cplx_out));
  else
    SMI_CHECK_FFT2(hipfftExecD2Z(p->fwd, (hipfftDoubleReal*)real_in, (hipfftDoubleComplex*)cplx_out));
  return 0;
}

}  // namespace

extern "C" int smi2d_diffusion_flux(smi_ctx* c, void* flux, const void* field, double prefactor) {
  SMI_CHECK(require_2d(c, "smi2d_diffusion_flux"));
  SMI_REQUIRE(flux && field && flux != field, "smi2d_diffusion_flux: bad argument");
  SMI_CHECK(halo2(c, (void*)field, 1));
  G2 G = geo2(c);
  SMI_DISPATCH(c, { k2_diffusion<T, false><<<nblocks(G.n), kB, 0, c->stream>>>((T*)flux, (const T*)field, coef_of<T>(prefactor), G); });
  SMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int smi2d_diffusion_timestep(smi_ctx* c, void* field, void* flux, double nu_dt_by_dx2) {
  SMI_CHECK(smi2d_diffusion_flux(c, flux, field, nu_dt_by_dx2));
  G2 G = geo2(c);
  SMI_DISPATCH(c, { k2_sum<T><<<nblocks(G.n), kB, 0, c->stream>>>((T*)field, (const T*)field, (const T*)flux, G.n); });
  SMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int smi2d_advection_timestep_eno3(smi_ctx* c, void* field, void* flux, const void* velocity,
                                             double dt_by_dx) {
  SMI_CHECK(require_2d(c, "smi2d_advection_timestep_eno3"));
  SMI_REQUIRE(field && flux && velocity && flux != field, "smi2d_advection_timestep_eno3: bad argument");
  SMI_REQUIRE(c->g >= 2, "smi2d_advection_timestep_eno3: kernel support 2 needs ghost_size >= 2 (got %d)", c->g);
  SMI_CHECK(halo2(c, field, 1));
  SMI_CHECK(halo2(c, (void*)velocity, 2));
  G2 G = geo2(c);
  SMI_DISPATCH(c, {
    k2_advection_flux<T><<<nblocks(G.n), kB, 0, c->stream>>>((T*)flux, (const T*)field, (const T*)velocity,
                                                            coef_of<T>(-dt_by_dx), G);
    k2_sum<T><<<nblocks(G.n), kB, 0, c->stream>>>((T*)field, (const T*)field, (const T*)flux, G.n);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

// gen_advection_flux_conservative_eno3_pyst_mpi_kernel_2d (advection_flux_mpi_2d.py): the flux alone
extern "C" int smi2d_advection_flux_eno3(smi_ctx* c, void* flux, const void* field, const void* velocity,
                                         double inv_dx) {
  SMI_CHECK(require_2d(c, "smi2d_advection_flux_eno3"));
  SMI_REQUIRE(field && flux && velocity && flux != field, "smi2d_advection_flux_eno3: bad argument");
  SMI_REQUIRE(c->g >= 2, "smi2d_advection_flux_eno3: kernel support 2 needs ghost_size >= 2 (got %d)", c->g);
  SMI_CHECK(halo2(c, (void*)field, 1));
  SMI_CHECK(halo2(c, (void*)velocity, 2));
  G2 G = geo2(c);
  SMI_DISPATCH(c, {
    k2_advection_flux<T, true><<<nblocks(G.n), kB, 0, c->stream>>>((T*)flux, (const T*)field, (const T*)velocity,
                                                                  coef_of<T>(inv_dx), G);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int smi2d_outplane_field_curl(smi_ctx* c, void* curl, const void* field, double prefactor) {
  SMI_CHECK(require_2d(c, "smi2d_outplane_field_curl"));
  SMI_REQUIRE(curl && field && curl != field, "smi2d_outplane_field_curl: bad argument");
  SMI_CHECK(halo2(c, (void*)field, 1));
  G2 G = geo2(c);
  SMI_DISPATCH(c, { k2_outplane_curl<T><<<nblocks(G.n), kB, 0, c->stream>>>((T*)curl, (const T*)field, coef_of<T>(prefactor), G); });
  SMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int smi2d_update_vorticity_from_velocity_forcing(smi_ctx* c, void* vorticity, const void* forcing,
                                                            double prefactor) {
  SMI_CHECK(require_2d(c, "smi2d_update_vorticity_from_velocity_forcing"));
  SMI_REQUIRE(vorticity && forcing && vorticity != forcing, "smi2d_update_vorticity_from_velocity_forcing: bad argument");
  SMI_CHECK(halo2(c, (void*)forcing, 2));
  G2 G = geo2(c);
  SMI_DISPATCH(c, {
    k2_update_vorticity<T><<<nblocks(G.n), kB, 0, c->stream>>>((T*)vorticity, (const T*)forcing, coef_of<T>(prefactor), G);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int smi2d_penalise_boundary(smi_ctx* c, void* field, int width, const double* fac_lo, const double* fac_hi) {
  SMI_CHECK(require_2d(c, "smi2d_penalise_boundary"));
  SMI_REQUIRE(field, "smi2d_penalise_boundary: bad argument");
  SMI_REQUIRE(width >= 0, "invalid zone width");
  if (width == 0) return 0;
  const int w = c->g + width;
  SMI_REQUIRE(w <= 16, "smi2d_penalise_boundary: ghost + width = %d exceeds 16", w);
  SMI_REQUIRE(fac_lo && fac_hi, "smi2d_penalise_boundary: null factor tables");
  SMI_REQUIRE(2 * w <= c->px && 2 * w <= c->py, "smi2d_penalise_boundary: penalty bands overlap on this grid");
  PenFac2 fac;
  for (int a = 0; a < 2; ++a)
    for (int k = 0; k < w; ++k) {
      fac.lo[a][k] = fac_lo[a * w + k];
      fac.hi[a][k] = fac_hi[a * w + k];
    }
  G2 G = geo2(c);
  SMI_DISPATCH(c, {
    k2_penalise<T, 0><<<nblocks(G.py), kB, 0, c->stream>>>((T*)field, G, w, fac, 1, 1);
    if (c->phys_ylo || c->phys_yhi)
      k2_penalise<T, 1><<<nblocks(G.px), kB, 0, c->stream>>>((T*)field, G, w, fac, c->phys_ylo, c->phys_yhi);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

static int set_add2(smi_ctx* c, void* out, const void* in, int ncomp, const double* vals, int add) {
  G2 G = geo2(c);
  SMI_REQUIRE(ncomp <= 2, "2D fields have one or two components (got %d)", ncomp);
  const long long total = (long long)ncomp * G.n;
  SMI_DISPATCH(c, {
    k2_set_add<T><<<nblocks(total), kB, 0, c->stream>>>((T*)out, (const T*)in, (T)vals[0], (T)vals[ncomp - 1], G.n, total,
                                                       add);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int smi2d_set_fixed_val(smi_ctx* c, void* field, int ncomp, const double* vals) {
  SMI_CHECK(require_2d(c, "smi2d_set_fixed_val"));
  SMI_REQUIRE(field && vals && ncomp >= 1, "smi2d_set_fixed_val: bad argument");
  return set_add2(c, field, field, ncomp, vals, 0);
}

extern "C" int smi2d_add_fixed_val(smi_ctx* c, void* sum_field, const void* field, int ncomp, const double* vals) {
  SMI_CHECK(require_2d(c, "smi2d_add_fixed_val"));
  SMI_REQUIRE(sum_field && field && vals && ncomp >= 1, "smi2d_add_fixed_val: bad argument");
  return set_add2(c, sum_field, field, ncomp, vals, 1);
}

template <int MODE>
static int reduce2(smi_ctx* c, const void* f, double* out) {
  G2 G = geo2(c);
  double* partials = (double*)c->red_partials;
  const long long cells = (long long)c->ny * c->nx;
  const int blocks = (int)((cells + 1023) / 1024 < kRed2Blocks ? (cells + 1023) / 1024 : kRed2Blocks);
  SMI_DISPATCH(c, { k2_reduce_partial<T, MODE><<<blocks, 256, 0, c->stream>>>((const T*)f, G, partials); });
  k2_reduce_final<<<1, kRed2Blocks, 0, c->stream>>>(partials, blocks, partials + kRed2Blocks);
  SMI_LAUNCH_CHECK();
  SMI_CHECK_HIP(hipMemcpyAsync(c->red_host, partials + kRed2Blocks, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  SMI_CHECK_HIP(hipStreamSynchronize(c->stream));
  *out = c->red_host[0];
  return 0;
}

extern "C" int smi2d_reduce_max_abs_sum(smi_ctx* c, const void* vector_field, double* out) {
  SMI_CHECK(require_2d(c, "smi2d_reduce_max_abs_sum"));
  SMI_REQUIRE(vector_field && out, "smi2d_reduce_max_abs_sum: bad argument");
  return reduce2<0>(c, vector_field, out);
}

extern "C" int smi2d_reduce_max(smi_ctx* c, const void* field, double* out) {
  SMI_CHECK(require_2d(c, "smi2d_reduce_max"));
  SMI_REQUIRE(field && out, "smi2d_reduce_max: bad argument");
  return reduce2<1>(c, field, out);
}

// ---- unbounded Poisson solve in 2D (UnboundedPoissonSolverMPI2D.py:61-153): pad to (2ny, 2nx),
// real-to-complex FFT, multiply by the transformed Green's function times dx^2, inverse, crop.
extern "C" int smi2d_poisson_create(smi_poisson2d** out, smi_ctx* c, const void* greens_host, double dx) {
  SMI_CHECK(require_2d(c, "smi2d_poisson_create"));
  SMI_REQUIRE(out && greens_host, "smi2d_poisson_create: bad argument");
  smi_poisson2d* p = new smi_poisson2d();
  memset(p, 0, sizeof(*p));
  p->ctx = c;
  p->ny_local = c->ny;
  p->row0 = c->rank_y * c->ny;
  p->ny = c->ny * c->nranks;
  p->nx = c->nx;
  const size_t es = smi_elem_size(c);
  const size_t nreal = 4 * (size_t)p->ny * p->nx, ncplx = 2 * (size_t)p->ny * (p->nx + 1);
  const bool f32 = c->dtype == SMI_F32;
  const int P = c->nranks;
  SMI_CHECK_HIP(hipSetDevice(c->device));
  SMI_CHECK_HIP(hipMalloc(&p->real_buf, nreal * es));
  SMI_CHECK_HIP(hipMalloc(&p->cplx_buf, ncplx * 2 * es));
  SMI_CHECK_HIP(hipMalloc(&p->ghat, ncplx * 2 * es));
  SMI_CHECK_FFT2(hipfftPlan2d(&p->fwd, 2 * p->ny, 2 * p->nx, f32 ? HIPFFT_R2C : HIPFFT_D2Z));
  SMI_CHECK_FFT2(hipfftSetStream(p->fwd, c->stream));
  SMI_CHECK_HIP(hipMemcpyAsync(p->real_buf, greens_host, nreal * es, hipMemcpyHostToDevice, c->stream));
  // dx^2 of the reference and the 1 / (2ny 2nx) of its normalised backward transform, folded in
  const double scale = dx * dx / (4.0 * (double)p->ny * (double)p->nx);
  SMI_DISPATCH(c, {
    SMI_CHECK((fft_forward<T>(p, p->real_buf, p->ghat)));
    k2_scale<T><<<nblocks(2 * (long long)ncplx), kB, 0, c->stream>>>((T*)p->ghat, (T)scale, 2 * (long long)ncplx);
  });
  SMI_LAUNCH_CHECK();
  if (P == 1) {
    SMI_CHECK_FFT2(hipfftPlan2d(&p->inv, 2 * p->ny, 2 * p->nx, f32 ? HIPFFT_C2R : HIPFFT_Z2D));
    SMI_CHECK_FFT2(hipfftSetStream(p->inv, c->stream));
  } else {
    // Divided transform: keep this rank's columns of the Green's spectrum, drop the whole-plane buffers and plan
    // (they served the set-up only), and plan the batched row and column transforms.
    p->KL = (p->nx + 1 + P - 1) / P;
    const size_t ncols = 2 * (size_t)p->ny * p->KL;
    SMI_CHECK_HIP(hipMalloc(&p->ghat_cols, ncols * 2 * es));
    SMI_DISPATCH(c, {
      k2_take_cols<T><<<nblocks((long long)ncols), kB, 0, c->stream>>>((T*)p->ghat_cols, (const T*)p->ghat, 2 * p->ny,
                                                                      p->nx, p->KL, c->rank * p->KL);
    });
    SMI_LAUNCH_CHECK();
    SMI_CHECK_HIP(hipStreamSynchronize(c->stream));
    hipfftDestroy(p->fwd);
    p->fwd = 0;
    SMI_CHECK_HIP(hipFree(p->real_buf));
    SMI_CHECK_HIP(hipFree(p->cplx_buf));
    SMI_CHECK_HIP(hipFree(p->ghat));
    p->real_buf = p->cplx_buf = p->ghat = nullptr;
    SMI_CHECK_HIP(hipMalloc(&p->row_real, 2 * (size_t)p->ny_local * p->nx * es));
    SMI_CHECK_HIP(hipMalloc(&p->row_spec, (size_t)p->ny_local * (p->nx + 1) * 2 * es));
    SMI_CHECK_HIP(hipMalloc(&p->xchg, (size_t)P * p->ny_local * p->KL * 2 * es));
    SMI_CHECK_HIP(hipMalloc(&p->col_spec, ncols * 2 * es));
    int nrow[1] = {2 * p->nx}, ncol[1] = {2 * p->ny};
    SMI_CHECK_FFT2(hipfftPlanMany(&p->row_fwd, 1, nrow, nullptr, 1, 0, nullptr, 1, 0, f32 ? HIPFFT_R2C : HIPFFT_D2Z,
                                  p->ny_local));
    SMI_CHECK_FFT2(hipfftPlanMany(&p->row_inv, 1, nrow, nullptr, 1, 0, nullptr, 1, 0, f32 ? HIPFFT_C2R : HIPFFT_Z2D,
                                  p->ny_local));
    SMI_CHECK_FFT2(hipfftPlanMany(&p->col, 1, ncol, ncol, p->KL, 1, ncol, p->KL, 1, f32 ? HIPFFT_C2C : HIPFFT_Z2Z,
                                  p->KL));
    SMI_CHECK_FFT2(hipfftSetStream(p->row_fwd, c->stream));
    SMI_CHECK_FFT2(hipfftSetStream(p->row_inv, c->stream));
    SMI_CHECK_FFT2(hipfftSetStream(p->col, c->stream));
  }
  SMI_CHECK_HIP(hipStreamSynchronize(c->stream));
  *out = p;
  return 0;
}

extern "C" int smi2d_poisson_destroy(smi_poisson2d* p) {
  if (!p) return 0;
  if (p->ctx) hipStreamSynchronize(p->ctx->stream);  // solves and captured steps still in flight use what is freed here
  if (p->fwd) hipfftDestroy(p->fwd);
  if (p->inv) hipfftDestroy(p->inv);
  hipFree(p->real_buf);
  hipFree(p->cplx_buf);
  hipFree(p->ghat);
  if (p->row_fwd) hipfftDestroy(p->row_fwd);
  if (p->row_inv) hipfftDestroy(p->row_inv);
  if (p->col) hipfftDestroy(p->col);
  hipFree(p->row_real);
  hipFree(p->row_spec);
  hipFree(p->xchg);
  hipFree(p->col_spec);
  hipFree(p->ghat_cols);
  if (p->step_exec) hipGraphExecDestroy(p->step_exec);
  if (p->step_graph) hipGraphDestroy(p->step_graph);
  if (p->step_coef) hipFree(p->step_coef);
  delete p;
  return 0;
}

// Several ranks (slabs of rows): the transform is divided instead of gathered.  Every rank transforms its own rows
// (zero padded in x only: the zero rows of the doubled plane are never formed), an all-to-all turns rows into
// spectral columns (KL of the nx + 1 per rank, all ny rows of the plane), the columns are zero padded to 2ny,
// transformed, multiplied by this rank's columns of the Green's spectrum and transformed back, and the mirror image
// brings the first ny rows home.  This replaces the reference's MPIDomainDoublingCommunicator2D + PFFT pair
// (UnboundedPoissonSolverMPI2D.py:120-153, 156-352) with two all-to-alls of (ny / P) x (nx + 1) complex values.
template <typename T>
static int solve_divided(smi_poisson2d* p, void* solution, const void* rhs) {
  smi_ctx* c = p->ctx;
  const int P = c->nranks, nyl = p->ny_local, nx = p->nx, KL = p->KL;
  const long long nblk = (long long)P * nyl * KL, nrows = 2LL * nyl * nx;
  const size_t per_peer = (size_t)nyl * KL * 2 * sizeof(T);
  constexpr bool f32 = sizeof(T) == 4;
  k2_pad_rows<T><<<nblocks(nrows), kB, 0, c->stream>>>((T*)p->row_real, (const T*)rhs, nyl, nx, c->g, c->px);
  if (f32)
    SMI_CHECK_FFT2(hipfftExecR2C(p->row_fwd, (hipfftReal*)p->row_real, (hipfftComplex*)p->row_spec));
  else
    SMI_CHECK_FFT2(hipfftExecD2Z(p->row_fwd, (hipfftDoubleReal*)p->row_real, (hipfftDoubleComplex*)p->row_spec));
  k2_col_blocks<T, true><<<nblocks(nblk), kB, 0, c->stream>>>((T*)p->row_spec, (T*)p->xchg, nyl, nx, KL, P);
  SMI_LAUNCH_CHECK();
  // block s of the result = rows s nyl .. of the plane: the first ny rows of the column array, in order
  SMI_CHECK(smi_alltoall_impl(c, p->xchg, p->col_spec, per_peer, c->stream));
  T* pad = (T*)p->col_spec + 2 * (size_t)p->ny * KL;
  k2_fill_zero<T><<<nblocks(2LL * p->ny * KL), kB, 0, c->stream>>>(pad, 2LL * p->ny * KL);
  if (f32) {
    SMI_CHECK_FFT2(hipfftExecC2C(p->col, (hipfftComplex*)p->col_spec, (hipfftComplex*)p->col_spec, HIPFFT_FORWARD));
  } else {
    SMI_CHECK_FFT2(hipfftExecZ2Z(p->col, (hipfftDoubleComplex*)p->col_spec, (hipfftDoubleComplex*)p->col_spec,
                                 HIPFFT_FORWARD));
  }
  k2_cmul<T><<<nblocks(2LL * p->ny * KL), kB, 0, c->stream>>>((T*)p->col_spec, (const T*)p->ghat_cols, 2LL * p->ny * KL);
  if (f32) {
    SMI_CHECK_FFT2(hipfftExecC2C(p->col, (hipfftComplex*)p->col_spec, (hipfftComplex*)p->col_spec, HIPFFT_BACKWARD));
  } else {
    SMI_CHECK_FFT2(hipfftExecZ2Z(p->col, (hipfftDoubleComplex*)p->col_spec, (hipfftDoubleComplex*)p->col_spec,
                                 HIPFFT_BACKWARD));
  }
  SMI_LAUNCH_CHECK();
  SMI_CHECK(smi_alltoall_impl(c, p->col_spec, p->xchg, per_peer, c->stream));
  k2_col_blocks<T, false><<<nblocks(nblk), kB, 0, c->stream>>>((T*)p->row_spec, (T*)p->xchg, nyl, nx, KL, P);
  if (f32)
    SMI_CHECK_FFT2(hipfftExecC2R(p->row_inv, (hipfftComplex*)p->row_spec, (hipfftReal*)p->row_real));
  else
    SMI_CHECK_FFT2(hipfftExecZ2D(p->row_inv, (hipfftDoubleComplex*)p->row_spec, (hipfftDoubleReal*)p->row_real));
  k2_extract<T><<<nblocks((long long)nyl * nx), kB, 0, c->stream>>>((T*)solution, (const T*)p->row_real, nyl, nx, c->g,
                                                                   c->px, 0);
  SMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int smi2d_poisson_solve(smi_poisson2d* p, void* solution, const void* rhs) {
  SMI_REQUIRE(p && solution && rhs, "smi2d_poisson_solve: bad argument");
  smi_ctx* c = p->ctx;
  if (c->nranks > 1) {
    SMI_DISPATCH(c, { SMI_CHECK(solve_divided<T>(p, solution, rhs)); });
    return 0;
  }
  const long long nreal = 4LL * p->ny * p->nx, ncplx = 2LL * p->ny * (p->nx + 1);
  const long long nlocal = (long long)p->ny_local * p->nx;
  SMI_DISPATCH(c, {
    k2_pad_load<T><<<nblocks(nreal), kB, 0, c->stream>>>((T*)p->real_buf, (const T*)rhs, p->ny, p->nx, c->g, c->px);
    SMI_CHECK((fft_forward<T>(p, p->real_buf, p->cplx_buf)));
    k2_cmul<T><<<nblocks(ncplx), kB, 0, c->stream>>>((T*)p->cplx_buf, (const T*)p->ghat, ncplx);
    if (sizeof(T) == 4)
      SMI_CHECK_FFT2(hipfftExecC2R(p->inv, (hipfftComplex*)p->cplx_buf, (hipfftReal*)p->real_buf));
    else
      SMI_CHECK_FFT2(hipfftExecZ2D(p->inv, (hipfftDoubleComplex*)p->cplx_buf, (hipfftDoubleReal*)p->real_buf));
    k2_extract<T><<<nblocks(nlocal), kB, 0, c->stream>>>((T*)solution, (const T*)p->real_buf, p->ny_local, p->nx, c->g,
                                                         c->px, p->row0);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

// ---- immersed boundary in 2D (EulerianLagrangianGridCommunicatorMPI2D.py:116-520) -------------
// Sixteen lanes per Lagrangian node: the 4 x 4 delta-function window, lane = (dy << 2) | dx.
// Lagrangian data are float64, the Eulerian grid is the context's real type.
namespace {

constexpr int kW2 = 2;  // interp_kernel_width (the reference asserts 2)

__device__ __forceinline__ double py_floordiv2(double a, double b) {
  double mod = fmod(a, b);
  double div = (a - mod) / b;
  if (mod != 0.0 && ((b < 0.0) != (mod < 0.0))) div -= 1.0;
  double fl;
  if (div != 0.0) {
    fl = floor(div);
    if (div - fl > 0.5) fl += 1.0;
  } else {
    fl = copysign(0.0, a / b);
  }
  return fl;
}

struct Ib2 {
  int py, px, g;
  long long n;
  double dx, shift;
  long long lsx, lsy;  // mpi_local_substart_coord_shift = first cell of this rank's block - ghost (x, y)
};

Ib2 make_ib2(const smi_ctx* c, double dx, double shift) {
  return Ib2{c->py, c->px, c->g, (long long)c->py * c->px, dx, shift, -(long long)c->g,
             (long long)c->rank_y * c->ny - c->g};
}

struct Delta2 {
  double c_cos, half_pi, c_pes;
};
// constants typed as the reference's njit closures type them (real_t values used in float64 arithmetic)
Delta2 make_delta2(const smi_ctx* c, double dx) {
  Delta2 d;
  d.c_cos = (0.25 / dx) * (0.25 / dx);
  d.half_pi = 0.5 * M_PI;
  d.c_pes = (0.125 / dx) * (0.125 / dx);
  if (c->dtype == SMI_F32) {
    d.c_cos = (double)(float)d.c_cos;
    d.half_pi = (double)(float)d.half_pi;
    d.c_pes = -d.c_pes;  // flags rounding of the Peskin weights to single precision
  }
  return d;
}

__device__ __forceinline__ double peskin2_1d(double r) {
  double a = (r < 1.0) ? (3.0 - 2 * r + sqrt(fabs(1 + 4 * r - 4 * r * r))) : 0.0;
  double b = (r >= 1.0 && r < 2.0) ? (5.0 - 2 * r - sqrt(fabs(-7 + 12 * r - 4 * r * r))) : 0.0;
  return a + b;
}

__global__ void __launch_bounds__(256) k2_ib_support(double* __restrict__ support, long long* __restrict__ index,
                                                     const double* __restrict__ pos, long long m, Ib2 G) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long node = gid >> 4;
  if (node >= m) return;
  const int lane = (int)(gid & 15), ox = (lane & 3) - kW2 + 1, oy = (lane >> 2) - kW2 + 1;
  const double X = pos[node], Y = pos[m + node];
  const long long ix = (long long)py_floordiv2(X - G.shift, G.dx) - G.lsx;
  const long long iy = (long long)py_floordiv2(Y - G.shift, G.dx) - G.lsy;
  if (lane == 0) {
    index[node] = ix;
    index[m + node] = iy;
  }
  const long long cell = (long long)lane * m + node;  // support[c][dy][dx][node]
  support[cell] = (double)(ix + ox + G.lsx) * G.dx + G.shift - X;
  support[16 * m + cell] = (double)(iy + oy + G.lsy) * G.dx + G.shift - Y;
}

__global__ void __launch_bounds__(256) k2_ib_weights(double* __restrict__ weights, double* __restrict__ support,
                                                     long long m, double dx, int kernel_type, Delta2 dc) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 16 * m) return;
  double sx = support[i], sy = support[16 * m + i], w;
  if (kernel_type == SMI_DELTA_COSINE) {
    sx /= dx;
    sy /= dx;
    w = dc.c_cos * (1.0 + cos(dc.half_pi * sx)) * (1.0 + cos(dc.half_pi * sy));
  } else {
    sx = fabs(sx) / dx;
    sy = fabs(sy) / dx;
    w = fabs(dc.c_pes) * peskin2_1d(sx) * peskin2_1d(sy);
    if (dc.c_pes < 0.0) w = (double)(float)w;
  }
  support[i] = sx;  // the reference mutates the support array in place
  support[16 * m + i] = sy;
  weights[i] = w;
}

__device__ __forceinline__ bool window2(const Ib2& G, long long ix, long long iy, int lane, long long& off) {
  const long long x = ix + (lane & 3) - kW2 + 1, y = iy + (lane >> 2) - kW2 + 1;
  off = y * G.px + x;
  return x >= 0 && x < G.px && y >= 0 && y < G.py;
}

template <typename T>
__global__ void __launch_bounds__(256) k2_ib_interp(double* __restrict__ lag, const T* __restrict__ eul,
                                                    const double* __restrict__ weights,
                                                    const long long* __restrict__ index, long long m, int ncomp,
                                                    Ib2 G, double dx_pow) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long node = gid >> 4;
  const int lane = (int)(gid & 15);
  const bool live = node < m;  // every lane takes part in the shuffles
  long long off = 0;
  const bool ok = live && window2(G, index[node], index[m + node], lane, off);
  const double w = live ? weights[(long long)lane * m + node] : 0.0;
  for (int c = 0; c < ncomp; ++c) {
    double v = ok ? (double)eul[c * G.n + off] * w : 0.0;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
    if (live && lane == 0) lag[c * m + node] = v * dx_pow;
  }
}

template <typename T>
__global__ void __launch_bounds__(256) k2_ib_spread(T* __restrict__ eul, const double* __restrict__ lag,
                                                    const double* __restrict__ weights,
                                                    const long long* __restrict__ index, long long m, int ncomp,
                                                    Ib2 G) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long node = gid >> 4;
  if (node >= m) return;
  const int lane = (int)(gid & 15);
  long long off;
  if (!window2(G, index[node], index[m + node], lane, off)) return;
  const double w = weights[(long long)lane * m + node];
  for (int c = 0; c < ncomp; ++c) atomicAdd(eul + c * G.n + off, (T)(lag[c * m + node] * w));
}

template <typename T>
__global__ void __launch_bounds__(kB) k2_clear_ghosts(T* __restrict__ f, G2 G, int ncomp) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G.n * ncomp) return;
  const long long r = i % G.n;
  const int y = (int)(r / G.px), x = (int)(r - (long long)y * G.px);
  if (x < G.g || x >= G.px - G.g || y < G.g || y >= G.py - G.g) f[i] = T(0);
}

}  // namespace

extern "C" int smi2d_ib_support(smi_ctx* c, void* support, void* nearest_index, const void* lag_positions, int64_t m,
                                double dx, double coord_shift) {
  SMI_CHECK(require_2d(c, "smi2d_ib_support"));
  SMI_REQUIRE(support && nearest_index && lag_positions && m >= 0, "smi2d_ib_support: bad argument");
  if (m == 0) return 0;
  k2_ib_support<<<nblocks(16 * m), 256, 0, c->stream>>>((double*)support, (long long*)nearest_index,
                                                       (const double*)lag_positions, m, make_ib2(c, dx, coord_shift));
  SMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int smi2d_ib_weights(smi_ctx* c, void* weights, void* support, int64_t m, double dx, int kernel_type) {
  SMI_CHECK(require_2d(c, "smi2d_ib_weights"));
  SMI_REQUIRE(weights && support && m >= 0, "smi2d_ib_weights: bad argument");
  SMI_REQUIRE(kernel_type == SMI_DELTA_COSINE || kernel_type == SMI_DELTA_PESKIN,
              "Invalid interpolation kernel type. Currently supported types are 'cosine' and 'peskin'.");
  if (m == 0) return 0;
  k2_ib_weights<<<nblocks(16 * m), 256, 0, c->stream>>>((double*)weights, (double*)support, m, dx, kernel_type,
                                                       make_delta2(c, dx));
  SMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int smi2d_ib_interpolate(smi_ctx* c, void* lag_field, const void* eul_field, const void* weights,
                                    const void* nearest_index, int64_t m, int ncomp, double dx) {
  SMI_CHECK(require_2d(c, "smi2d_ib_interpolate"));
  SMI_REQUIRE(lag_field && eul_field && weights && nearest_index && m >= 0 && ncomp >= 1,
              "smi2d_ib_interpolate: bad argument");
  SMI_REQUIRE(c->g >= kW2, "ghost size (%d) needs to be >= interp kernel width (%d)", c->g, kW2);
  if (m == 0) return 0;
  Ib2 G = make_ib2(c, dx, 0.0);
  SMI_DISPATCH(c, {
    k2_ib_interp<T><<<nblocks(16 * m), 256, 0, c->stream>>>((double*)lag_field, (const T*)eul_field,
                                                          (const double*)weights, (const long long*)nearest_index, m,
                                                          ncomp, G, dx * dx);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

// Spreading followed by the ghost sum: ghost rows are added to the neighbouring ranks' edge rows, then every ghost
// cell is cleared (on a single rank every neighbour is PROC_NULL and only the clearing remains)
extern "C" int smi2d_ib_spread(smi_ctx* c, void* eul_field, const void* lag_field, const void* weights,
                               const void* nearest_index, int64_t m, int ncomp, int ghost_sum) {
  SMI_CHECK(require_2d(c, "smi2d_ib_spread"));
  SMI_REQUIRE(eul_field && lag_field && weights && nearest_index && m >= 0 && ncomp >= 1,
              "smi2d_ib_spread: bad argument");
  SMI_REQUIRE(c->g >= kW2, "ghost size (%d) needs to be >= interp kernel width (%d)", c->g, kW2);
  Ib2 G = make_ib2(c, 1.0, 0.0);
  G2 P = geo2(c);
  SMI_DISPATCH(c, {
    if (m > 0)
      k2_ib_spread<T><<<nblocks(16 * m), 256, 0, c->stream>>>((T*)eul_field, (const double*)lag_field,
                                                            (const double*)weights, (const long long*)nearest_index,
                                                            m, ncomp, G);
  });
  SMI_LAUNCH_CHECK();
  if (ghost_sum) {
    SMI_CHECK(ghost_sum2(c, eul_field, ncomp));
    SMI_DISPATCH(c, { k2_clear_ghosts<T><<<nblocks(P.n * ncomp), kB, 0, c->stream>>>((T*)eul_field, P, ncomp); });
    SMI_LAUNCH_CHECK();
  }
  return 0;
}

// MPIGhostSumCommunicator2D.ghost_sum / clear_ghost_cells on their own
// (EulerianLagrangianGridCommunicatorMPI2D.py:521-652): the ghost rows are added to the neighbouring ranks' edge rows,
// then every ghost cell is zeroed; clear_only != 0 skips the sum
extern "C" int smi2d_ghost_sum(smi_ctx* c, void* field, int ncomp, int clear_only) {
  SMI_CHECK(require_2d(c, "smi2d_ghost_sum"));
  SMI_REQUIRE(field && ncomp >= 1 && ncomp <= 2, "smi2d_ghost_sum: bad argument");
  if (!clear_only) SMI_CHECK(ghost_sum2(c, field, ncomp));
  G2 P = geo2(c);
  SMI_DISPATCH(c, { k2_clear_ghosts<T><<<nblocks(P.n * ncomp), kB, 0, c->stream>>>((T*)field, P, ncomp); });
  SMI_LAUNCH_CHECK();
  return 0;
}

// MPIGhostCommunicator2D.exchange_scalar_field_init / exchange_vector_field_init (utils/mpi_utils_2d.py:286-420)
extern "C" int smi2d_halo_exchange(smi_ctx* c, void* field, int ncomp) {
  SMI_CHECK(require_2d(c, "smi2d_halo_exchange"));
  SMI_REQUIRE(field && ncomp >= 1, "smi2d_halo_exchange: bad argument");
  return halo2(c, field, ncomp);
}

// One 2D Navier-Stokes step sequenced inside the library (flow_simulators_mpi_2d.py:260-298): the
// forcing update, ENO3 advection, diffusion, penalisation, Poisson solve, curl and free stream,
// one host call instead of seven.  The 256^2 step is bound by launch overhead (some fifteen kernels of a few
// microseconds each), so on a single rank the sequence is captured into a HIP graph once and replayed: the arrays
// and the penalisation tables are the same every step, and what changes with the time step - four prefactors and the
// free stream - is read by the kernels from a small device array that a one-lane kernel refreshes before each
// replay.  SMI_2D_NO_GRAPH=1 keeps the launch-by-launch path.
namespace {
enum { CF_FORCING, CF_ADVECT, CF_DIFFUSE, CF_CURL, CF_FS_U, CF_FS_V, CF_COUNT };
struct CoefVals {
  double v[CF_COUNT];
};
__global__ void k2_write_coefs(double* __restrict__ dst, CoefVals vals) {
  if (threadIdx.x < CF_COUNT) dst[threadIdx.x] = vals.v[threadIdx.x];
}

// the launches of one step; dev != nullptr: coefficients come from that device array (capture / replay)
template <typename T>
int step2_launches(smi_ctx* c, smi_poisson2d* p, T* w, T* vel, T* psi, T* buf, T* forcing, const CoefVals& val,
                   const double* dev, int penalty_width, const double* fac_lo, const double* fac_hi, bool has_fs) {
  auto cf = [&](int i) { return Coef<T>{(T)val.v[i], dev ? dev + i : nullptr}; };
  const G2 G = geo2(c);
  const int nb = nblocks(G.n);
  if (forcing) {
    SMI_CHECK(halo2(c, forcing, 2));
    k2_update_vorticity<T><<<nb, kB, 0, c->stream>>>(w, forcing, cf(CF_FORCING), G);
  }
  SMI_CHECK(halo2(c, w, 1));
  SMI_CHECK(halo2(c, vel, 2));
  k2_advection_flux<T><<<nb, kB, 0, c->stream>>>(buf, w, vel, cf(CF_ADVECT), G);
  k2_sum<T><<<nb, kB, 0, c->stream>>>(w, w, buf, G.n);
  SMI_CHECK(halo2(c, w, 1));
  k2_diffusion<T, false><<<nb, kB, 0, c->stream>>>(buf, w, cf(CF_DIFFUSE), G);
  k2_sum<T><<<nb, kB, 0, c->stream>>>(w, w, buf, G.n);
  SMI_LAUNCH_CHECK();
  if (penalty_width > 0) SMI_CHECK(smi2d_penalise_boundary(c, w, penalty_width, fac_lo, fac_hi));
  SMI_CHECK(smi2d_poisson_solve(p, psi, w));
  SMI_CHECK(halo2(c, psi, 1));
  if (has_fs)
    k2_outplane_curl<T, true><<<nb, kB, 0, c->stream>>>(vel, psi, cf(CF_CURL), G, cf(CF_FS_U), cf(CF_FS_V));
  else
    k2_outplane_curl<T, false><<<nb, kB, 0, c->stream>>>(vel, psi, cf(CF_CURL), G);
  SMI_LAUNCH_CHECK();
  if (forcing) {
    const double z2[2] = {0.0, 0.0};
    SMI_CHECK(set_add2(c, forcing, forcing, 2, z2, 0));
  }
  return 0;
}

void drop_step_graph(smi_poisson2d* p) {
  // replays of the old graph may still be running (a rare event: the arrays of a simulator do not change)
  if (p->step_exec && p->ctx) hipStreamSynchronize(p->ctx->stream);
  if (p->step_exec) hipGraphExecDestroy(p->step_exec);
  if (p->step_graph) hipGraphDestroy(p->step_graph);
  p->step_exec = nullptr;
  p->step_graph = nullptr;
  if (p->step_state > 0) p->step_state = 0;
}
}  // namespace

extern "C" int smi2d_flow_step_ns(smi_ctx* c, smi_poisson2d* p, void* vorticity, void* velocity, void* stream_func,
                                  void* buffer_scalar, void* forcing, double dt, double nu, double dx,
                                  int penalty_width, const double* fac_lo, const double* fac_hi,
                                  const double* free_stream) {
  SMI_REQUIRE(c && p && vorticity && velocity && stream_func && buffer_scalar, "smi2d_flow_step_ns: null argument");
  SMI_REQUIRE(p->ctx == c, "smi2d_flow_step_ns: the Poisson solver belongs to another context");
  SMI_REQUIRE(c->g >= 2, "smi2d_flow_step_ns: the advection stencil needs ghost_size >= 2 (got %d)", c->g);
  SMI_REQUIRE(penalty_width == 0 || (fac_lo && fac_hi), "smi2d_flow_step_ns: penalisation tables missing");
  SMI_REQUIRE(penalty_width >= 0 && c->g + penalty_width <= 16, "smi2d_flow_step_ns: bad penalty zone width %d",
              penalty_width);
  // prefactors rounded to the real type exactly as the reference does with real_t(...)
  auto r = [&](double v) { return c->dtype == SMI_F32 ? (double)(float)v : v; };
  CoefVals val;
  val.v[CF_FORCING] = r(dt / (2 * dx));
  val.v[CF_ADVECT] = -r(dt / dx);
  val.v[CF_DIFFUSE] = r(nu * dt / dx / dx);
  val.v[CF_CURL] = r(0.5 / dx);
  val.v[CF_FS_U] = free_stream ? r(free_stream[0]) : 0.0;
  val.v[CF_FS_V] = free_stream ? r(free_stream[1]) : 0.0;
  const bool has_fs = free_stream != nullptr;
  const int nfac = 2 * (c->g + penalty_width);  // entries of each table: x band, then y band
  auto eager = [&]() -> int {
    ++p->eager_steps;
    SMI_DISPATCH(c, {
      return step2_launches<T>(c, p, (T*)vorticity, (T*)velocity, (T*)stream_func, (T*)buffer_scalar, (T*)forcing, val,
                               nullptr, penalty_width, fac_lo, fac_hi, has_fs);
    });
    return 0;
  };
  static const bool no_graph = getenv("SMI_2D_NO_GRAPH") != nullptr;
  const bool profiling = c->prof && c->prof->on;
  if (no_graph || c->nranks > 1 || profiling || p->step_state < 0 || p->eager_steps < 2) return eager();

  const void* key[5] = {vorticity, velocity, stream_func, buffer_scalar, forcing};
  bool same = p->step_state == 1 && memcmp(key, p->step_key, sizeof(key)) == 0 && p->step_width == penalty_width &&
              p->step_has_fs == (int)has_fs;
  if (same && penalty_width > 0)
    same = memcmp(p->step_fac[0], fac_lo, nfac * sizeof(double)) == 0 &&
           memcmp(p->step_fac[1], fac_hi, nfac * sizeof(double)) == 0;
  if (!same) {
    drop_step_graph(p);
    if (!p->step_coef) SMI_CHECK_HIP(hipMalloc((void**)&p->step_coef, sizeof(double) * 8));
    SMI_CHECK_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    int rc = 0;
    SMI_DISPATCH(c, {
      rc = step2_launches<T>(c, p, (T*)vorticity, (T*)velocity, (T*)stream_func, (T*)buffer_scalar, (T*)forcing, val,
                             p->step_coef, penalty_width, fac_lo, fac_hi, has_fs);
    });
    hipError_t e = hipStreamEndCapture(c->stream, &p->step_graph);  // always: leaves capture mode
    if (rc == 0 && e == hipSuccess) e = hipGraphInstantiate(&p->step_exec, p->step_graph, nullptr, nullptr, 0);
    if (rc != 0 || e != hipSuccess) {
      fprintf(stderr, "libsophtmi: capturing the 2D step failed (%s); steps run launch by launch\n",
              rc ? smi_last_error() : hipGetErrorString(e));
      (void)hipGetLastError();
      drop_step_graph(p);
      p->step_state = -1;
      return eager();
    }
    memcpy(p->step_key, key, sizeof(key));
    p->step_width = penalty_width;
    p->step_has_fs = has_fs;
    if (penalty_width > 0) {
      memcpy(p->step_fac[0], fac_lo, nfac * sizeof(double));
      memcpy(p->step_fac[1], fac_hi, nfac * sizeof(double));
    }
    p->step_state = 1;
  }
  k2_write_coefs<<<1, 64, 0, c->stream>>>(p->step_coef, val);
  SMI_LAUNCH_CHECK();
  SMI_CHECK_HIP(hipGraphLaunch(p->step_exec, c->stream));
  return 0;
}
