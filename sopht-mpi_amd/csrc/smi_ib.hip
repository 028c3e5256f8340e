// Immersed-boundary operators of libsophtmi (gfx950).
//
// One 64-lane wavefront per Lagrangian node: the 4 x 4 x 4 delta-function window is
// exactly one wave, lane = (dz << 4) | (dy << 2) | dx.  Eulerian -> Lagrangian
// interpolation is a wave-wide shuffle reduction in double; Lagrangian -> Eulerian
// spreading is one float atomic per lane (rows of four consecutive cells).
// Lagrangian data are float64 and the Eulerian grid is the context's real type, as in
// the reference (VirtualBoundaryForcingMPI.py:78-81).
#include "smi_common.h"
#include "smi_prof.h"

namespace {
constexpr int kW = 2;          // interp_kernel_width; the reference asserts 2 for both kernels
constexpr int kWin = 2 * kW;   // 4
constexpr int kWaves = 4;      // nodes per 256-thread workgroup

// Python / numpy floor division of doubles (npy_divmod): exact floor of the true quotient.
__device__ __forceinline__ double py_floordiv(double a, double b) {
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

struct IbGeo {
  int pz, py, px;
  long long plane, vol;
  double dx, shift;
  long long lsx, lsy, lsz;  // mpi_local_substart_coord_shift = substart - ghost (x, y, z)
};

IbGeo make_ibgeo(const smi_ctx* c, double dx, double shift, const int64_t* substart_xyz) {
  IbGeo G;
  G.pz = c->pz;
  G.py = c->py;
  G.px = c->px;
  G.plane = (long long)c->plane;
  G.vol = (long long)c->vol;
  G.dx = dx;
  G.shift = shift;
  G.lsx = substart_xyz[0] - c->g;
  G.lsy = substart_xyz[1] - c->g;
  G.lsz = substart_xyz[2] - c->g;
  return G;
}

// EulerianLagrangianGridCommunicatorMPI3D.py:156-158
__device__ __forceinline__ long long nearest_idx(double X, const IbGeo& G, long long ls) {
  return (long long)py_floordiv(X - G.shift, G.dx) - ls;
}
// :161-176, one window cell
__device__ __forceinline__ double support_val(long long idx, int off, long long ls, double X, const IbGeo& G) {
  return (double)(idx + off + ls) * G.dx + G.shift - X;
}

__device__ __forceinline__ double cosine_1d(double s_over_dx, double half_pi) { return 1.0 + cos(half_pi * s_over_dx); }
__device__ __forceinline__ double peskin_1d(double r) {
  double a = (r < 1.0) ? (3.0 - 2 * r + sqrt(fabs(1 + 4 * r - 4 * r * r))) : 0.0;
  double b = (r >= 1.0 && r < 2.0) ? (5.0 - 2 * r - sqrt(fabs(-7 + 12 * r - 4 * r * r))) : 0.0;
  return a + b;
}

// weight of this lane's window cell, computed from the three supports (already / dx for
// cosine, |.| / dx for Peskin)
__device__ __forceinline__ double delta_weight(double sx, double sy, double sz, int kernel_type, double c_cos,
                                               double half_pi, double c_pes) {
  if (kernel_type == SMI_DELTA_COSINE)
    return c_cos * cosine_1d(sx, half_pi) * cosine_1d(sy, half_pi) * cosine_1d(sz, half_pi);
  // the reference rounds the Peskin weights to the Eulerian precision (.astype(real_t), :587);
  // c_pes < 0 flags single precision
  double w = fabs(c_pes) * peskin_1d(sx) * peskin_1d(sy) * peskin_1d(sz);
  return c_pes < 0.0 ? (double)(float)w : w;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

struct DeltaConst {
  double c_cos, half_pi, c_pes;
};
// constants as the reference's njit closures type them: real_t((0.25/dx)**3) and
// real_t(0.5*pi) are Eulerian-precision values used in float64 arithmetic (:464-467)
DeltaConst make_delta_const(const smi_ctx* c, double dx) {
  DeltaConst d;
  double cc = (0.25 / dx) * (0.25 / dx) * (0.25 / dx);
  double hp = 0.5 * M_PI;
  if (c->dtype == SMI_F32) {
    cc = (double)(float)cc;
    hp = (double)(float)hp;
  }
  d.c_cos = cc;
  d.half_pi = hp;
  d.c_pes = (0.125 / dx) * (0.125 / dx) * (0.125 / dx);
  if (c->dtype == SMI_F32) d.c_pes = -d.c_pes;
  return d;
}

// ---------------------------------------------------------------- materialised operators
__global__ void __launch_bounds__(256) k_ib_support(double* __restrict__ support, long long* __restrict__ index,
                                                    const double* __restrict__ pos, long long m, IbGeo G) {
  long long node = (long long)blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (node >= m) return;
  int lane = threadIdx.x & 63;
  int ox = (lane & 3) - kW + 1, oy = ((lane >> 2) & 3) - kW + 1, oz = (lane >> 4) - kW + 1;
  double X = pos[node], Y = pos[m + node], Z = pos[2 * m + node];
  long long ix = nearest_idx(X, G, G.lsx), iy = nearest_idx(Y, G, G.lsy), iz = nearest_idx(Z, G, G.lsz);
  if (lane == 0) {
    index[node] = ix;
    index[m + node] = iy;
    index[2 * m + node] = iz;
  }
  // support[c][dz][dy][dx][node]
  long long cell = (long long)lane * m + node;
  support[cell] = support_val(ix, ox, G.lsx, X, G);
  support[64 * m + cell] = support_val(iy, oy, G.lsy, Y, G);
  support[128 * m + cell] = support_val(iz, oz, G.lsz, Z, G);
}

__global__ void __launch_bounds__(256) k_ib_weights(double* __restrict__ weights, double* __restrict__ support,
                                                    long long m, double dx, int kernel_type, DeltaConst dc) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over 64 * m window cells
  if (i >= 64 * m) return;
  double sx = support[i], sy = support[64 * m + i], sz = support[128 * m + i];
  if (kernel_type == SMI_DELTA_COSINE) {
    sx /= dx;
    sy /= dx;
    sz /= dx;
  } else {
    sx = fabs(sx) / dx;
    sy = fabs(sy) / dx;
    sz = fabs(sz) / dx;
  }
  // the reference mutates the support array in place (:462, :501-503)
  support[i] = sx;
  support[64 * m + i] = sy;
  support[128 * m + i] = sz;
  weights[i] = delta_weight(sx, sy, sz, kernel_type, dc.c_cos, dc.half_pi, dc.c_pes);
}

__device__ __forceinline__ bool window_cell(const IbGeo& G, long long ix, long long iy, long long iz, int lane,
                                            long long& off) {
  long long x = ix + (lane & 3) - kW + 1, y = iy + ((lane >> 2) & 3) - kW + 1, z = iz + (lane >> 4) - kW + 1;
  off = z * G.plane + y * G.px + x;
  return x >= 0 && x < G.px && y >= 0 && y < G.py && z >= 0 && z < G.pz;
}

template <typename T>
__global__ void __launch_bounds__(256) k_ib_interp(double* __restrict__ lag, const T* __restrict__ eul,
                                                   const double* __restrict__ weights,
                                                   const long long* __restrict__ index, long long m, int ncomp,
                                                   IbGeo G, double dx_pow) {
  long long node = (long long)blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (node >= m) return;
  int lane = threadIdx.x & 63;
  long long off;
  bool ok = window_cell(G, index[node], index[m + node], index[2 * m + node], lane, off);
  double w = weights[(long long)lane * m + node];
  for (int c = 0; c < ncomp; ++c) {
    double v = ok ? (double)eul[c * G.vol + off] * w : 0.0;
    v = wave_sum(v);
    if (lane == 0) lag[c * m + node] = v * dx_pow;
  }
}

template <typename T>
__global__ void __launch_bounds__(256) k_ib_spread(T* __restrict__ eul, const double* __restrict__ lag,
                                                   const double* __restrict__ weights,
                                                   const long long* __restrict__ index, long long m, int ncomp,
                                                   IbGeo G) {
  long long node = (long long)blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (node >= m) return;
  int lane = threadIdx.x & 63;
  long long off;
  bool ok = window_cell(G, index[node], index[m + node], index[2 * m + node], lane, off);
  double w = weights[(long long)lane * m + node];
  if (!ok) return;
  for (int c = 0; c < ncomp; ++c) atomicAdd(eul + c * G.vol + off, (T)(lag[c * m + node] * w));
}

// ---------------------------------------------------------------- fused operators
__device__ __forceinline__ double lane_weight(const IbGeo& G, double X, double Y, double Z, long long ix,
                                              long long iy, long long iz, int lane, int kernel_type,
                                              const DeltaConst& dc) {
  int ox = (lane & 3) - kW + 1, oy = ((lane >> 2) & 3) - kW + 1, oz = (lane >> 4) - kW + 1;
  double sx = support_val(ix, ox, G.lsx, X, G), sy = support_val(iy, oy, G.lsy, Y, G),
         sz = support_val(iz, oz, G.lsz, Z, G);
  if (kernel_type == SMI_DELTA_COSINE) {
    sx /= G.dx;
    sy /= G.dx;
    sz /= G.dx;
  } else {
    sx = fabs(sx) / G.dx;
    sy = fabs(sy) / G.dx;
    sz = fabs(sz) / G.dx;
  }
  return delta_weight(sx, sy, sz, kernel_type, dc.c_cos, dc.half_pi, dc.c_pes);
}

template <typename T>
__global__ void __launch_bounds__(256) k_ib_forcing_fused(double* __restrict__ forcing, double* __restrict__ vmis,
                                                          double* __restrict__ flowvel,
                                                          const double* __restrict__ pmis,
                                                          const double* __restrict__ pos,
                                                          const double* __restrict__ vel, const T* __restrict__ eul,
                                                          long long m, IbGeo G, double dx_pow, double stiffness,
                                                          double damping, int kernel_type, DeltaConst dc) {
  long long node = (long long)blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (node >= m) return;
  int lane = threadIdx.x & 63;
  double X = pos[node], Y = pos[m + node], Z = pos[2 * m + node];
  long long ix = nearest_idx(X, G, G.lsx), iy = nearest_idx(Y, G, G.lsy), iz = nearest_idx(Z, G, G.lsz);
  double w = lane_weight(G, X, Y, Z, ix, iy, iz, lane, kernel_type, dc);
  long long off;
  bool ok = window_cell(G, ix, iy, iz, lane, off);
  for (int c = 0; c < 3; ++c) {
    double v = ok ? (double)eul[c * G.vol + off] * w : 0.0;
    v = wave_sum(v) * dx_pow;
    if (lane == 0) {
      double dv = v - vel[c * m + node];
      flowvel[c * m + node] = v;
      vmis[c * m + node] = dv;
      forcing[c * m + node] = stiffness * pmis[c * m + node] + damping * dv;
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256) k_ib_spread_fused(T* __restrict__ eul, const double* __restrict__ forcing,
                                                         const double* __restrict__ pos, long long m, IbGeo G,
                                                         int kernel_type, DeltaConst dc) {
  long long node = (long long)blockIdx.x * kWaves + (threadIdx.x >> 6);
  if (node >= m) return;
  int lane = threadIdx.x & 63;
  double X = pos[node], Y = pos[m + node], Z = pos[2 * m + node];
  long long ix = nearest_idx(X, G, G.lsx), iy = nearest_idx(Y, G, G.lsy), iz = nearest_idx(Z, G, G.lsz);
  double w = lane_weight(G, X, Y, Z, ix, iy, iz, lane, kernel_type, dc);
  long long off;
  if (!window_cell(G, ix, iy, iz, lane, off)) return;
  for (int c = 0; c < 3; ++c) atomicAdd(eul + c * G.vol + off, (T)(forcing[c * m + node] * w));
}

// Node-ordered spreading (smi_set_ordered_spread): one wave per component walks the nodes in index order and adds
// each node's 4^3 window with a load, an add in double and a store, like the reference's serial loop over nodes
// (EulerianLagrangianGridCommunicatorMPI3D.py:404-425: field[window] += force * weights, one node after the other),
// so overlapping windows accumulate in node order with one rounding to the field type per node and the result does
// not depend on how the hardware schedules atomics.  The loads and stores are device-scope atomic accesses (they
// bypass the per-CU cache).  Windows of consecutive nodes overlap, so one lane's load for node n + 1 may hit the
// address another lane stored for node n: a wavefront-scope fence at the top of every iteration makes the wave wait
// until the stores of the previous node have been acknowledged before it issues the next loads.  A few microseconds
// per node: for parity runs, not for production.
// WEIGHTS: materialised weights / index arrays (smi_ib_spread) instead of positions (smi_ib_spread_fused).
template <typename T, bool WEIGHTS>
__global__ void __launch_bounds__(64) k_ib_spread_ordered(T* __restrict__ eul, const double* __restrict__ forcing,
                                                          const double* __restrict__ pos,
                                                          const double* __restrict__ weights,
                                                          const long long* __restrict__ index, long long m, IbGeo G,
                                                          int kernel_type, DeltaConst dc) {
  const int c = blockIdx.x, lane = threadIdx.x;
  T* field = eul + (long long)c * G.vol;
  for (long long node = 0; node < m; ++node) {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    long long ix, iy, iz;
    double w;
    if (WEIGHTS) {
      ix = index[node], iy = index[m + node], iz = index[2 * m + node];
      w = weights[(long long)lane * m + node];
    } else {
      const double X = pos[node], Y = pos[m + node], Z = pos[2 * m + node];
      ix = nearest_idx(X, G, G.lsx), iy = nearest_idx(Y, G, G.lsy), iz = nearest_idx(Z, G, G.lsz);
      w = lane_weight(G, X, Y, Z, ix, iy, iz, lane, kernel_type, dc);
    }
    long long off;
    if (!window_cell(G, ix, iy, iz, lane, off)) continue;
    const double add = forcing[c * m + node] * w;
    const T old = __hip_atomic_load(field + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(field + off, (T)((double)old + add), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void __launch_bounds__(256) k_axpy_f64(double* __restrict__ y, const double* __restrict__ x, long long n,
                                                  double a) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = y[i] + a * x[i];
}

int node_blocks(long long m) { return (int)((m + kWaves - 1) / kWaves); }
}  // namespace

extern "C" int smi_ib_support(smi_ctx* c, void* support, void* nearest_index, const void* lag_positions, int64_t m,
                              double dx, double coord_shift, const int64_t* substart_xyz) {
  SMI_REQUIRE(c && support && nearest_index && lag_positions && substart_xyz && m >= 0, "smi_ib_support: bad argument");
  SMI_REQUIRE(c->g >= kW, "ghost size (%d) needs to be >= interp kernel width (%d)", c->g, kW);
  if (m == 0) return 0;
  IbGeo G = make_ibgeo(c, dx, coord_shift, substart_xyz);
  k_ib_support<<<node_blocks(m), 256, 0, c->stream>>>((double*)support, (long long*)nearest_index,
                                                     (const double*)lag_positions, m, G);
  SMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int smi_ib_weights(smi_ctx* c, void* weights, void* support, int64_t m, double dx, int kernel_type) {
  SMI_REQUIRE(c && weights && support && m >= 0, "smi_ib_weights: bad argument");
  SMI_REQUIRE(kernel_type == SMI_DELTA_COSINE || kernel_type == SMI_DELTA_PESKIN,
              "Invalid interpolation kernel type. Currently supported types are 'cosine' and 'peskin'.");
  if (m == 0) return 0;
  DeltaConst dc = make_delta_const(c, dx);
  long long n = 64 * (long long)m;
  k_ib_weights<<<(int)((n + 255) / 256), 256, 0, c->stream>>>((double*)weights, (double*)support, m, dx, kernel_type,
                                                             dc);
  SMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int smi_ib_interpolate(smi_ctx* c, void* lag_field, const void* eul_field, const void* weights,
                                  const void* nearest_index, int64_t m, int ncomp, double dx) {
  SMI_REQUIRE(c && lag_field && eul_field && weights && nearest_index && m >= 0, "smi_ib_interpolate: bad argument");
  SMI_REQUIRE(ncomp == 1 || ncomp == 3, "invalid number of components for interpolation!");
  if (m == 0) return 0;
  int64_t zero3[3] = {0, 0, 0};
  IbGeo G = make_ibgeo(c, dx, 0.0, zero3);
  double dx_pow = dx * dx * dx;
  SMI_DISPATCH(c, {
    k_ib_interp<T><<<node_blocks(m), 256, 0, c->stream>>>((double*)lag_field, (const T*)eul_field,
                                                         (const double*)weights, (const long long*)nearest_index, m,
                                                         ncomp, G, dx_pow);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int smi_ib_spread(smi_ctx* c, void* eul_field, const void* lag_field, const void* weights,
                             const void* nearest_index, int64_t m, int ncomp) {
  SMI_REQUIRE(c && lag_field && eul_field && weights && nearest_index && m >= 0, "smi_ib_spread: bad argument");
  SMI_REQUIRE(ncomp == 1 || ncomp == 3, "invalid number of components for interpolation!");
  if (m == 0) return 0;
  int64_t zero3[3] = {0, 0, 0};
  IbGeo G = make_ibgeo(c, 1.0, 0.0, zero3);
  if (c->ordered_spread) {
    SMI_DISPATCH(c, {
      k_ib_spread_ordered<T, true><<<ncomp, 64, 0, c->stream>>>((T*)eul_field, (const double*)lag_field, nullptr,
                                                               (const double*)weights,
                                                               (const long long*)nearest_index, m, G, 0, DeltaConst{});
    });
    SMI_LAUNCH_CHECK();
    return 0;
  }
  SMI_DISPATCH(c, {
    k_ib_spread<T><<<node_blocks(m), 256, 0, c->stream>>>((T*)eul_field, (const double*)lag_field,
                                                         (const double*)weights, (const long long*)nearest_index, m,
                                                         ncomp, G);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int smi_ib_forcing_fused(smi_ctx* c, void* lag_forcing, void* lag_velocity_mismatch, void* lag_flow_velocity,
                                    const void* lag_position_mismatch, const void* lag_positions,
                                    const void* lag_velocities, const void* eul_velocity, int64_t m, double dx,
                                    double coord_shift, const int64_t* substart_xyz, double stiffness, double damping,
                                    int kernel_type) {
  SMI_REQUIRE(c && lag_forcing && lag_velocity_mismatch && lag_flow_velocity && lag_position_mismatch &&
                  lag_positions && lag_velocities && eul_velocity && substart_xyz && m >= 0,
              "smi_ib_forcing_fused: bad argument");
  SMI_REQUIRE(c->g >= kW, "ghost size (%d) needs to be >= interp kernel width (%d)", c->g, kW);
  SMI_REQUIRE(kernel_type == SMI_DELTA_COSINE || kernel_type == SMI_DELTA_PESKIN,
              "Invalid interpolation kernel type. Currently supported types are 'cosine' and 'peskin'.");
  if (m == 0) return 0;
  IbGeo G = make_ibgeo(c, dx, coord_shift, substart_xyz);
  DeltaConst dc = make_delta_const(c, dx);
  // per node: 64 window cells x 3 components read, 10 doubles of Lagrangian data
  SmiProfScope prof(c, PROF_IB_FORCING, (double)m * (64.0 * 3 * smi_elem_size(c) + 80.0));
  SMI_DISPATCH(c, {
    k_ib_forcing_fused<T><<<node_blocks(m), 256, 0, c->stream>>>(
        (double*)lag_forcing, (double*)lag_velocity_mismatch, (double*)lag_flow_velocity,
        (const double*)lag_position_mismatch, (const double*)lag_positions, (const double*)lag_velocities,
        (const T*)eul_velocity, m, G, dx * dx * dx, stiffness, damping, kernel_type, dc);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int smi_ib_spread_fused(smi_ctx* c, void* eul_forcing, const void* lag_forcing, const void* lag_positions,
                                   int64_t m, double dx, double coord_shift, const int64_t* substart_xyz,
                                   int kernel_type) {
  SMI_REQUIRE(c && eul_forcing && lag_forcing && lag_positions && substart_xyz && m >= 0,
              "smi_ib_spread_fused: bad argument");
  SMI_REQUIRE(kernel_type == SMI_DELTA_COSINE || kernel_type == SMI_DELTA_PESKIN,
              "Invalid interpolation kernel type. Currently supported types are 'cosine' and 'peskin'.");
  if (m == 0) return 0;
  IbGeo G = make_ibgeo(c, dx, coord_shift, substart_xyz);
  DeltaConst dc = make_delta_const(c, dx);
  // per node: 64 window cells x 3 components read-modify-written
  SmiProfScope prof(c, PROF_IB_SPREAD, (double)m * (64.0 * 3 * 2 * smi_elem_size(c) + 48.0));
  if (c->ordered_spread) {
    SMI_DISPATCH(c, {
      k_ib_spread_ordered<T, false><<<3, 64, 0, c->stream>>>((T*)eul_forcing, (const double*)lag_forcing,
                                                            (const double*)lag_positions, nullptr, nullptr, m, G,
                                                            kernel_type, dc);
    });
    SMI_LAUNCH_CHECK();
    return 0;
  }
  SMI_DISPATCH(c, {
    k_ib_spread_fused<T><<<node_blocks(m), 256, 0, c->stream>>>((T*)eul_forcing, (const double*)lag_forcing,
                                                               (const double*)lag_positions, m, G, kernel_type, dc);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int smi_set_ordered_spread(smi_ctx* c, int on) {
  SMI_REQUIRE(c, "smi_set_ordered_spread: null context");
  c->ordered_spread = on != 0;
  return 0;
}

extern "C" int smi_ib_mismatch_step(smi_ctx* c, void* position_mismatch, const void* velocity_mismatch, int64_t n,
                                    double dt) {
  SMI_REQUIRE(c && position_mismatch && velocity_mismatch && n >= 0, "smi_ib_mismatch_step: bad argument");
  if (n == 0) return 0;
  k_axpy_f64<<<(int)((n + 255) / 256), 256, 0, c->stream>>>((double*)position_mismatch,
                                                           (const double*)velocity_mismatch, n, dt);
  SMI_LAUNCH_CHECK();
  return 0;
}

// out = a x + b y on n doubles: each of the three pointwise kernels of VirtualBoundaryForcingMPI.py:278-331 on its own
// (velocity mismatch: 1, flow, -1, body; position mismatch: 1, itself, dt, velocity mismatch; penalty force: k, position
// mismatch, c, velocity mismatch).  a x is formed first and b y added to it in one fused step.
__global__ void __launch_bounds__(256) k_axpby_f64(double* __restrict__ out, double a, const double* __restrict__ x,
                                                   double b, const double* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = fma(b, y[i], a * x[i]);
}

extern "C" int smi_ib_axpby(smi_ctx* c, void* out, double a, const void* x, double b, const void* y, int64_t n) {
  SMI_REQUIRE(c && out && x && y && n >= 0, "smi_ib_axpby: bad argument");
  if (n == 0) return 0;
  k_axpby_f64<<<(int)((n + 255) / 256), 256, 0, c->stream>>>((double*)out, a, (const double*)x, b, (const double*)y, n);
  SMI_LAUNCH_CHECK();
  return 0;
}

// VirtualBoundaryForcingMPI.py:278-331, the three pointwise kernels in one: velocity mismatch = interpolated flow
// velocity - body velocity, force = k * position mismatch + c * velocity mismatch.  n = dim * nodes doubles.
__global__ void __launch_bounds__(256) k_penalty_force(double* __restrict__ force, double* __restrict__ vel_mismatch,
                                                       const double* __restrict__ flow_vel,
                                                       const double* __restrict__ body_vel,
                                                       const double* __restrict__ pos_mismatch, int64_t n, double k,
                                                       double cdamp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double dv = flow_vel[i] - body_vel[i];
  vel_mismatch[i] = dv;
  force[i] = k * pos_mismatch[i] + cdamp * dv;
}

extern "C" int smi_ib_penalty_force(smi_ctx* c, void* lag_forcing, void* velocity_mismatch, const void* flow_velocity,
                                    const void* body_velocity, const void* position_mismatch, int64_t n,
                                    double stiffness, double damping) {
  SMI_REQUIRE(c && lag_forcing && velocity_mismatch && flow_velocity && body_velocity && position_mismatch && n >= 0,
              "smi_ib_penalty_force: bad argument");
  if (n == 0) return 0;
  k_penalty_force<<<(int)((n + 255) / 256), 256, 0, c->stream>>>(
      (double*)lag_forcing, (double*)velocity_mismatch, (const double*)flow_velocity, (const double*)body_velocity,
      (const double*)position_mismatch, n, stiffness, damping);
  SMI_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- forcing grids of framed bodies on the device
// A rigid sphere, or the surface rings of a Cosserat rod, is a set of nodes each tied to one FRAME (the sphere; a rod
// element): node = centre + Q^T (scale * local offset), velocity = v + (Q^T omega) x arm.  The host sends the frames
// (a few doubles per element) and the device forms the (3, M) node arrays where the interaction kernels read them,
// instead of the host forming them in numpy and uploading 6 M doubles per interaction.
// Frame record, 19 doubles: centre (3), directors Q row by row (9; rows = material axes in lab coordinates, the
// PyElastica convention), velocity of the centre (3), angular velocity in the material frame (3), scale (1).
constexpr int kFrameDoubles = 19;

__global__ void __launch_bounds__(256) k_frame_nodes(double* __restrict__ pos, double* __restrict__ vel,
                                                     double* __restrict__ arm, const double* __restrict__ local,
                                                     const int* __restrict__ frame_of_node,
                                                     const double* __restrict__ frames, int64_t m) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= m) return;
  const double* F = frames + (int64_t)kFrameDoubles * frame_of_node[n];
  const double s = F[18];
  const double l0 = s * local[n], l1 = s * local[m + n], l2 = s * local[2 * m + n];
  double a[3], w[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    a[i] = F[3 + i] * l0 + F[6 + i] * l1 + F[9 + i] * l2;             // Q^T l
    w[i] = F[3 + i] * F[15] + F[6 + i] * F[16] + F[9 + i] * F[17];    // Q^T omega
  }
  pos[n] = F[0] + a[0];
  pos[m + n] = F[1] + a[1];
  pos[2 * m + n] = F[2] + a[2];
  vel[n] = F[12] + (w[1] * a[2] - w[2] * a[1]);
  vel[m + n] = F[13] + (w[2] * a[0] - w[0] * a[2]);
  vel[2 * m + n] = F[14] + (w[0] * a[1] - w[1] * a[0]);
  arm[n] = a[0];
  arm[m + n] = a[1];
  arm[2 * m + n] = a[2];
}

extern "C" int smi_ib_frame_nodes(smi_ctx* c, void* positions, void* velocities, void* arms, const void* local_offsets,
                                  const void* frame_of_node, const void* frames, int n_frames, int64_t m) {
  SMI_REQUIRE(c && positions && velocities && arms && local_offsets && frame_of_node && frames && n_frames >= 1 &&
                  m >= 0,
              "smi_ib_frame_nodes: bad argument");
  if (m == 0) return 0;
  k_frame_nodes<<<(int)((m + 255) / 256), 256, 0, c->stream>>>((double*)positions, (double*)velocities, (double*)arms,
                                                               (const double*)local_offsets, (const int*)frame_of_node,
                                                               (const double*)frames, m);
  SMI_LAUNCH_CHECK();
  return 0;
}

// Per frame: sum of the node forces and of arm x force (lab frame), six doubles.  One workgroup per frame walks the
// frame's nodes (contiguous: frame_start[e] .. frame_start[e + 1] - 1) and reduces in a fixed order, so the result is
// reproducible.  transfer_forcing_from_grid_to_body of the forcing grids is built from these sums on the host.
__global__ void __launch_bounds__(256) k_frame_loads(double* __restrict__ loads, const double* __restrict__ force,
                                                     const double* __restrict__ arm,
                                                     const int* __restrict__ frame_start, int64_t m) {
  __shared__ double part[6][256];
  const int e = blockIdx.x, t = threadIdx.x;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (int n = frame_start[e] + t; n < frame_start[e + 1]; n += 256) {
    const double f0 = force[n], f1 = force[m + n], f2 = force[2 * m + n];
    const double a0 = arm[n], a1 = arm[m + n], a2 = arm[2 * m + n];
    acc[0] += f0;
    acc[1] += f1;
    acc[2] += f2;
    acc[3] += a1 * f2 - a2 * f1;
    acc[4] += a2 * f0 - a0 * f2;
    acc[5] += a0 * f1 - a1 * f0;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) part[k][t] = acc[k];
  __syncthreads();
  for (int half = 128; half >= 1; half >>= 1) {
    if (t < half) {
#pragma unroll
      for (int k = 0; k < 6; ++k) part[k][t] += part[k][t + half];
    }
    __syncthreads();
  }
  if (t < 6) loads[6 * e + t] = part[t][0];
}

extern "C" int smi_ib_frame_loads(smi_ctx* c, void* loads, const void* lag_forcing, const void* arms,
                                  const void* frame_start, int n_frames, int64_t m) {
  SMI_REQUIRE(c && loads && lag_forcing && arms && frame_start && n_frames >= 1 && m >= 0,
              "smi_ib_frame_loads: bad argument");
  k_frame_loads<<<n_frames, 256, 0, c->stream>>>((double*)loads, (const double*)lag_forcing, (const double*)arms,
                                                 (const int*)frame_start, m);
  SMI_LAUNCH_CHECK();
  return 0;
}
