// Pruned unbounded Poisson solve (SMI_POISSON_PRUNED), gfx950.
//
// The reference zero-pads the right-hand side to (2nz, 2ny, 2nx), transforms, multiplies by
// the Green's-function spectrum and transforms back, keeping one octant
// (UnboundedPoissonSolverMPI3D.py:133-167).  Seven eighths of that doubled array are
// zeros going in and discarded coming out.  Here the padding is never stored:
//
//   a 2L-point DFT of a sequence whose upper half is zero splits into two L-point DFTs,
//       X[2m]   = DFT_L(a)[m],        X[2m+1] = DFT_L(a[n] e^{-i pi n / L})[m],
//   and the first L outputs of a 2L-point inverse DFT are
//       a[n] = IDFT_L(X[even])[n] + e^{+i pi n / L} IDFT_L(X[odd])[n].
//
// Five kernels per solve, all batched 1D FFTs held in LDS (one tile = TS sequences of L
// points; Stockham radix-8/4/2 stages, eight points per work item in registers):
//   x_fwd   real rows (read straight from the ghosted field) -> nx+1 complex per row
//   y_fwd   columns ny -> 2ny
//   z_conv  columns nz -> 2nz, times the Green's spectrum, 2nz -> nz   (one kernel, in place)
//   y_inv   columns 2ny -> ny
//   x_inv   complex rows -> real rows written straight into the ghosted solution field
// Spectral indices are stored split by parity ([even | odd]); the Green's table is laid out
// in the same order, folded by its even symmetry and pre-multiplied by dx^3 and by the
// backward normalisation, so it is real and (nz+1)(ny+1)(nx+1) in size.
//
// Slabs: after x_fwd each rank holds its z planes for all kx; one all-to-all hands every
// rank a contiguous kx range for ALL z (before any y / z padding exists), y and z passes
// run locally, a second all-to-all goes back.  x_fwd writes and x_inv reads the exchange
// layout directly, so there is no pack / unpack pass.
#include <vector>

#include "smi_common.h"
#include "smi_poisson.h"
#include "smi_prof.h"

namespace {

template <typename T>
struct alignas(2 * sizeof(T)) Cx {
  T x, y;
};
template <typename T>
__device__ __forceinline__ Cx<T> operator+(Cx<T> a, Cx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T>
__device__ __forceinline__ Cx<T> operator-(Cx<T> a, Cx<T> b) { return {a.x - b.x, a.y - b.y}; }
template <typename T>
__device__ __forceinline__ Cx<T> operator*(Cx<T> a, Cx<T> b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
template <typename T>
__device__ __forceinline__ Cx<T> scale(Cx<T> a, T s) { return {a.x * s, a.y * s}; }
template <typename T>
__device__ __forceinline__ Cx<T> conj(Cx<T> a) { return {a.x, -a.y}; }
// multiply by SIGN * i  (SIGN = -1: forward transform, +1: inverse)
template <int SIGN, typename T>
__device__ __forceinline__ Cx<T> mul_si(Cx<T> a) {
  return SIGN < 0 ? Cx<T>{a.y, -a.x} : Cx<T>{-a.y, a.x};
}

struct FftPlan {
  int L, L8, nstages;
  int radix[5];
};

FftPlan make_plan(int L) {
  FftPlan p;
  p.L = L;
  p.L8 = L / 8;
  p.nstages = 0;
  int lg = 0;
  while ((1 << lg) < L) ++lg;
  while (lg >= 3) {
    p.radix[p.nstages++] = 8;
    lg -= 3;
  }
  if (lg == 2) p.radix[p.nstages++] = 4;
  if (lg == 1) p.radix[p.nstages++] = 2;
  return p;
}

// tw4[k] = exp(-2 pi i k / (4L)), k in [0, 4L); SIGN > 0 conjugates.
template <int SIGN, typename T>
__device__ __forceinline__ Cx<T> twl(const Cx<T>* __restrict__ tw4, int idx) {
  Cx<T> w = tw4[idx];
  if (SIGN > 0) w.y = -w.y;
  return w;
}

template <int SIGN, typename T>
__device__ __forceinline__ void dft4(Cx<T>& x0, Cx<T>& x1, Cx<T>& x2, Cx<T>& x3) {
  Cx<T> t0 = x0 + x2, t1 = x0 - x2, t2 = x1 + x3, t3 = mul_si<SIGN>(x1 - x3);
  x0 = t0 + t2;
  x1 = t1 + t3;
  x2 = t0 - t2;
  x3 = t1 - t3;
}

template <int SIGN, typename T>
__device__ __forceinline__ void dft8(Cx<T> (&v)[8]) {
  Cx<T> e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
  dft4<SIGN>(e0, e1, e2, e3);
  dft4<SIGN>(o0, o1, o2, o3);
  const T h = (T)0.70710678118654752440;
  // w8^k, w8 = exp(SIGN * 2 pi i / 8)
  Cx<T> t1 = o1 * Cx<T>{h, SIGN * h};
  Cx<T> t2 = mul_si<SIGN>(o2);
  Cx<T> t3 = o3 * Cx<T>{-h, SIGN * h};
  v[0] = e0 + o0;
  v[4] = e0 - o0;
  v[1] = e1 + t1;
  v[5] = e1 - t1;
  v[2] = e2 + t2;
  v[6] = e2 - t2;
  v[3] = e3 + t3;
  v[7] = e3 - t3;
}

// One Stockham stage for the work item that owns elements u + m * L/8 (m = 0..7).
// Inputs in v; outputs back in v, pos[m] = their position in the stage's output sequence.
// Radix-R butterfly j takes inputs j + r * L/R, multiplies input r by w_{Ns R}^{r k}
// (k = j mod Ns) and writes (j - k) R + k + r Ns.  With R < 8 the item runs 8/R butterflies
// j = u + q L/8, which own exactly the same eight elements.
template <int SIGN, typename T>
__device__ __forceinline__ void butterflies(Cx<T> (&v)[8], int (&pos)[8], int u, int R, int Ns, const FftPlan& P,
                                            const Cx<T>* __restrict__ tw4) {
  if (R == 8) {
    const int k = u & (Ns - 1);
    if (Ns > 1) {
      const int step = (P.L / (Ns * 8)) * 4 * k;
      Cx<T> w1 = twl<SIGN>(tw4, step), w2 = twl<SIGN>(tw4, 2 * step), w4 = twl<SIGN>(tw4, 4 * step);
      Cx<T> w3 = w1 * w2, w5 = w4 * w1, w6 = w4 * w2, w7 = w4 * w3;
      v[1] = v[1] * w1;
      v[2] = v[2] * w2;
      v[3] = v[3] * w3;
      v[4] = v[4] * w4;
      v[5] = v[5] * w5;
      v[6] = v[6] * w6;
      v[7] = v[7] * w7;
    }
    dft8<SIGN>(v);
    const int j0 = (u - k) * 8 + k;
#pragma unroll
    for (int r = 0; r < 8; ++r) pos[r] = j0 + r * Ns;
  } else if (R == 4) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int j = u + q * P.L8, k = j & (Ns - 1);
      if (Ns > 1) {
        const int step = (P.L / (Ns * 4)) * 4 * k;
        Cx<T> w1 = twl<SIGN>(tw4, step), w2 = twl<SIGN>(tw4, 2 * step);
        Cx<T> w3 = w1 * w2;
        v[q + 2] = v[q + 2] * w1;
        v[q + 4] = v[q + 4] * w2;
        v[q + 6] = v[q + 6] * w3;
      }
      dft4<SIGN>(v[q], v[q + 2], v[q + 4], v[q + 6]);
      const int j0 = (j - k) * 4 + k;
#pragma unroll
      for (int r = 0; r < 4; ++r) pos[q + 2 * r] = j0 + r * Ns;
    }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = u + q * P.L8, k = j & (Ns - 1);
      Cx<T> b = v[q + 4];
      if (Ns > 1) b = b * twl<SIGN>(tw4, (P.L / (Ns * 2)) * 4 * k);
      Cx<T> a = v[q];
      v[q] = a + b;
      v[q + 4] = a - b;
      const int j0 = (j - k) * 2 + k;
      pos[q] = j0;
      pos[q + 4] = j0 + Ns;
    }
  }
}

constexpr int kThreads = 256;      // pointwise helper kernels
constexpr int kMaxThreads = 1024;  // FFT kernels: one work item per thread, TS * L / 8 threads

// The work item of this thread: sequence t (fastest over lanes) and element group u.
struct Items {
  int ts_mask, ts_shift;
  __device__ __forceinline__ int t(int) const { return threadIdx.x & ts_mask; }
  __device__ __forceinline__ int u(int) const { return threadIdx.x >> ts_shift; }
};

// All stages of one batched FFT.  LDS tile layout: lds[pos * TSP + t].
// FIRST_FROM_LDS: stage 0 inputs are read from the tile (the caller filled it and must NOT
// have synchronised) instead of being supplied in v.  On return v / pos hold the last
// stage's outputs; every read of the tile has completed (barrier), so the caller may
// overwrite it.
template <int SIGN, bool FIRST_FROM_LDS, int BPT, typename T>
__device__ __forceinline__ void fft_run(Cx<T> (&v)[BPT][8], int (&pos)[BPT][8], const FftPlan& P, Cx<T>* lds,
                                        int TSP, const Items& it, const Cx<T>* __restrict__ tw4) {
  int Ns = 1;
  for (int s = 0; s < P.nstages; ++s) {
    if (s > 0 || FIRST_FROM_LDS) {
      if (s > 0) {
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
          const int t = it.t(i);
#pragma unroll
          for (int m = 0; m < 8; ++m) lds[pos[i][m] * TSP + t] = v[i][m];
        }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < BPT; ++i) {
        const int t = it.t(i), u = it.u(i);
#pragma unroll
        for (int m = 0; m < 8; ++m) v[i][m] = lds[(u + m * P.L8) * TSP + t];
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < BPT; ++i) butterflies<SIGN>(v[i], pos[i], it.u(i), P.radix[s], Ns, P, tw4);
    Ns *= P.radix[s];
  }
}

// Geometry of the spectral buffers.
//   bufA (exchange layout) : [rank P][comp nc][z nzl][y ny][KXP]      x_fwd out, x_inv in
//   bufT (after all-to-all): [slab P][comp nc][z nzl][y ny][KXP]      y_fwd in, y_inv out
//   bufB                   : [comp nc][z nz][ky' 2ny][KXP]            z_conv in place
//   green                  : [kz folded nz+1][ky folded ny+1][KXP]    real
// KXL = spectral columns per rank (kx in parity-split order j = 0..nx, cut into P ranges),
// KXP = row pitch (KXL rounded up to 16 columns).
struct PGeo {
  int nz, ny, nx, nzl, P, nc, KXL, KXP;
};

// ------------------------------------------------------------------------- y forward
template <typename T, int BPT>
__global__ void __launch_bounds__(kMaxThreads) k_fft_y_fwd(const Cx<T>* __restrict__ in, Cx<T>* __restrict__ out,
                                                        PGeo G, FftPlan P, const Cx<T>* __restrict__ tw4, int ts_shift) {
  extern __shared__ char smem[];
  Cx<T>* lds = reinterpret_cast<Cx<T>*>(smem);
  const int TS = 1 << ts_shift;
  const Items it{TS - 1, ts_shift};
  const int c = blockIdx.z, z = blockIdx.y, kx0 = blockIdx.x * TS;
  const Cx<T>* src = in + ((((size_t)(z / G.nzl) * G.nc + c) * G.nzl + (z % G.nzl)) * G.ny) * (size_t)G.KXP;
  Cx<T>* dst = out + (((size_t)c * G.nz + z) * (2 * G.ny)) * (size_t)G.KXP;
  Cx<T> v[BPT][8], a0[BPT][8];
  int pos[BPT][8];
  // the column is read once and kept in registers for both parities
#pragma unroll
  for (int i = 0; i < BPT; ++i) {
    const int col = kx0 + it.t(i), u = it.u(i);
    const bool ok = col < G.KXL;
#pragma unroll
    for (int m = 0; m < 8; ++m) a0[i][m] = ok ? src[(size_t)(u + m * P.L8) * G.KXP + col] : Cx<T>{0, 0};
  }
  for (int parity = 0; parity < 2; ++parity) {
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      const int u = it.u(i);
#pragma unroll
      for (int m = 0; m < 8; ++m) v[i][m] = parity ? a0[i][m] * tw4[2 * (u + m * P.L8)] : a0[i][m];
    }
    fft_run<-1, false, BPT>(v, pos, P, lds, TS, it, tw4);
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      const int col = kx0 + it.t(i);
      if (col < G.KXL) {
#pragma unroll
        for (int m = 0; m < 8; ++m) dst[(size_t)(parity * P.L + pos[i][m]) * G.KXP + col] = v[i][m];
      }
    }
  }
}

// ------------------------------------------------------------------------- y inverse
template <typename T, int BPT>
__global__ void __launch_bounds__(kMaxThreads) k_fft_y_inv(const Cx<T>* __restrict__ in, Cx<T>* __restrict__ out,
                                                        PGeo G, FftPlan P, const Cx<T>* __restrict__ tw4, int ts_shift) {
  extern __shared__ char smem[];
  Cx<T>* lds = reinterpret_cast<Cx<T>*>(smem);
  const int TS = 1 << ts_shift;
  const Items it{TS - 1, ts_shift};
  const int c = blockIdx.z, z = blockIdx.y, kx0 = blockIdx.x * TS;
  const Cx<T>* src = in + (((size_t)c * G.nz + z) * (2 * G.ny)) * (size_t)G.KXP;
  Cx<T>* dst = out + ((((size_t)(z / G.nzl) * G.nc + c) * G.nzl + (z % G.nzl)) * G.ny) * (size_t)G.KXP;
  Cx<T> v[BPT][8], res[BPT][8];
  int pos[BPT][8];
  for (int parity = 0; parity < 2; ++parity) {
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      const int col = kx0 + it.t(i), u = it.u(i);
      const bool ok = col < G.KXL;
#pragma unroll
      for (int m = 0; m < 8; ++m)
        v[i][m] = ok ? src[(size_t)(parity * P.L + u + m * P.L8) * G.KXP + col] : Cx<T>{0, 0};
    }
    fft_run<+1, false, BPT>(v, pos, P, lds, TS, it, tw4);
    if (parity == 0) {
#pragma unroll
      for (int i = 0; i < BPT; ++i)
#pragma unroll
        for (int m = 0; m < 8; ++m) res[i][m] = v[i][m];
    } else {
#pragma unroll
      for (int i = 0; i < BPT; ++i) {
        const int col = kx0 + it.t(i);
        if (col < G.KXL) {
#pragma unroll
          for (int m = 0; m < 8; ++m) {
            const int n = pos[i][m];
            dst[(size_t)n * G.KXP + col] = res[i][m] + conj(tw4[2 * n]) * v[i][m];
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------- z convolution
// forward nz -> 2nz, multiply by the folded Green's table, inverse 2nz -> nz, in place.
template <typename T, int BPT>
__global__ void __launch_bounds__(kMaxThreads) k_fft_z_conv(Cx<T>* __restrict__ buf, const T* __restrict__ green, PGeo G,
                                                         FftPlan P, const Cx<T>* __restrict__ tw4, int ts_shift) {
  extern __shared__ char smem[];
  Cx<T>* lds = reinterpret_cast<Cx<T>*>(smem);
  const int TS = 1 << ts_shift;
  const Items it{TS - 1, ts_shift};
  const int c = blockIdx.z, kyp = blockIdx.y, kx0 = blockIdx.x * TS;
  const size_t zstride = (size_t)2 * G.ny * G.KXP;
  Cx<T>* base = buf + ((size_t)c * G.nz * 2 * G.ny + kyp) * (size_t)G.KXP;
  const int ky = 2 * (kyp % G.ny) + kyp / G.ny;
  const int kyc = ky <= G.ny ? ky : 2 * G.ny - ky;
  const T* gbase = green + (size_t)kyc * G.KXP;
  const size_t gstride = (size_t)(G.ny + 1) * G.KXP;
  Cx<T> v[BPT][8], res[BPT][8], a0[BPT][8];
  int pos[BPT][8];
  // the column is read once and kept in registers for both parities
#pragma unroll
  for (int i = 0; i < BPT; ++i) {
    const int col = kx0 + it.t(i), u = it.u(i);
    const bool ok = col < G.KXL;
#pragma unroll
    for (int m = 0; m < 8; ++m) a0[i][m] = ok ? base[(size_t)(u + m * P.L8) * zstride + col] : Cx<T>{0, 0};
  }
  for (int pz = 0; pz < 2; ++pz) {
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      const int u = it.u(i);
#pragma unroll
      for (int m = 0; m < 8; ++m) v[i][m] = pz ? a0[i][m] * tw4[2 * (u + m * P.L8)] : a0[i][m];
    }
    fft_run<-1, false, BPT>(v, pos, P, lds, TS, it, tw4);
    // spectrum at kz = 2 pos + pz, times the Green's table; into the tile in natural order
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      const int t = it.t(i), col = kx0 + t;
      const bool ok = col < G.KXL;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int kz = 2 * pos[i][m] + pz;
        const int kzc = kz <= G.nz ? kz : 2 * G.nz - kz;
        const T gval = ok ? gbase[(size_t)kzc * gstride + col] : T(0);
        lds[pos[i][m] * TS + t] = scale(v[i][m], gval);
      }
    }
    fft_run<+1, true, BPT>(v, pos, P, lds, TS, it, tw4);
    if (pz == 0) {
#pragma unroll
      for (int i = 0; i < BPT; ++i)
#pragma unroll
        for (int m = 0; m < 8; ++m) res[i][m] = v[i][m];
    } else {
#pragma unroll
      for (int i = 0; i < BPT; ++i) {
        const int col = kx0 + it.t(i);
        if (col < G.KXL) {
#pragma unroll
          for (int m = 0; m < 8; ++m) {
            const int n = pos[i][m];
            base[(size_t)n * zstride + col] = res[i][m] + conj(tw4[2 * n]) * v[i][m];
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------- x forward
// Real rows of 2L = nx points -> nx + 1 complex, through an L-point complex FFT of the
// packed row z[n] = x[2n] + i x[2n+1] (zero padded to 2L, hence again an even / odd pair
// of L-point transforms) and the real-FFT untangle
//   X[k] = (Z[k] + conj Z[M-k]) / 2 - (i/2) e^{-2 pi i k / 2M} (Z[k] - conj Z[M-k]),  M = nx.
// Even k only need the even transform, odd k only the odd one.  Output column order:
// j = 0..L holds kx = 2j, j = L+1..2L holds kx = 2(j-L-1)+1.
struct FieldGeo {
  int g, px;
  long long plane, vol;
};

template <typename T, int BPT>
__global__ void __launch_bounds__(kMaxThreads) k_fft_x_fwd(const T* __restrict__ field, Cx<T>* __restrict__ out, PGeo G,
                                                        FieldGeo F, FftPlan P, const Cx<T>* __restrict__ tw4,
                                                        int ts_shift) {
  extern __shared__ char smem[];
  Cx<T>* lds = reinterpret_cast<Cx<T>*>(smem);
  const int TS = 1 << ts_shift, TSP = TS + 1;
  const Items it{TS - 1, ts_shift};
  const int c = blockIdx.z, zl = blockIdx.y, y0 = blockIdx.x * TS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int L = P.L;
  const T* fbase = field + (size_t)c * F.vol + (size_t)(zl + F.g) * F.plane + F.g;
  Cx<T> v[BPT][8];
  int pos[BPT][8];
  for (int parity = 0; parity < 2; ++parity) {
    for (int r = wave; r < TS; r += nwaves) {
      const int y = y0 + r;
      const T* row = fbase + (size_t)(y + F.g) * F.px;
      for (int n = lane; n < L; n += 64) {
        Cx<T> a{0, 0};
        if (y < G.ny) {
          a.x = row[2 * n];
          a.y = row[2 * n + 1];
        }
        if (parity) a = a * tw4[2 * n];
        lds[n * TSP + r] = a;
      }
    }
    fft_run<-1, true, BPT>(v, pos, P, lds, TSP, it, tw4);
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
      const int t = it.t(i);
#pragma unroll
      for (int m = 0; m < 8; ++m) lds[pos[i][m] * TSP + t] = v[i][m];
    }
    __syncthreads();
    const int cnt = L + 1 - parity;
    for (int r = wave; r < TS; r += nwaves) {
      const int y = y0 + r;
      if (y >= G.ny) continue;
      for (int m = lane; m < cnt; m += 64) {
        Cx<T> a, b, w;
        if (parity == 0) {
          a = lds[(m & (L - 1)) * TSP + r];
          b = conj(lds[((L - m) & (L - 1)) * TSP + r]);
          w = tw4[2 * m];
        } else {
          a = lds[m * TSP + r];
          b = conj(lds[(L - 1 - m) * TSP + r]);
          w = tw4[2 * m + 1];
        }
        Cx<T> s = a + b, d = w * (a - b);
        Cx<T> X{(T)0.5 * (s.x + d.y), (T)0.5 * (s.y - d.x)};  // s/2 - (i/2) d
        const int j = parity ? L + 1 + m : m;
        const int rk = j / G.KXL, jl = j - rk * G.KXL;
        out[((((size_t)rk * G.nc + c) * G.nzl + zl) * G.ny + y) * (size_t)G.KXP + jl] = X;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------- x inverse
template <typename T, int BPT>
__global__ void __launch_bounds__(kMaxThreads) k_fft_x_inv(const Cx<T>* __restrict__ in, T* __restrict__ field, PGeo G,
                                                        FieldGeo F, FftPlan P, const Cx<T>* __restrict__ tw4,
                                                        int ts_shift) {
  extern __shared__ char smem[];
  Cx<T>* lds = reinterpret_cast<Cx<T>*>(smem);
  const int TS = 1 << ts_shift, TSP = TS + 1;
  const Items it{TS - 1, ts_shift};
  const int c = blockIdx.z, zl = blockIdx.y, y0 = blockIdx.x * TS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int L = P.L;
  T* fbase = field + (size_t)c * F.vol + (size_t)(zl + F.g) * F.plane + F.g;
  Cx<T> v[BPT][8], res[BPT][8];
  int pos[BPT][8];
  for (int parity = 0; parity < 2; ++parity) {
    const int cnt = L + 1 - parity;
    for (int r = wave; r < TS; r += nwaves) {
      const int y = y0 + r;
      for (int m = lane; m < cnt; m += 64) {
        Cx<T> a{0, 0};
        if (y < G.ny) {
          const int j = parity ? L + 1 + m : m;
          const int rk = j / G.KXL, jl = j - rk * G.KXL;
          a = in[((((size_t)rk * G.nc + c) * G.nzl + zl) * G.ny + y) * (size_t)G.KXP + jl];
        }
        lds[m * TSP + r] = a;
      }
    }
    __syncthreads();
    // Z[k] = E[k] + i O[k], E = (X[k] + conj X[M-k]) / 2, O = (X[k] - conj X[M-k]) / 2 * e^{+2 pi i k / 2M};
    // pairs (k, M-k) share their two inputs, so each lane rewrites one pair in place.
    const int npair = parity ? L / 2 : L / 2 + 1;
    for (int r = wave; r < TS; r += nwaves) {
      for (int m = lane; m < npair; m += 64) {
        const int m2 = parity ? L - 1 - m : L - m;  // partner index in this parity block
        Cx<T> xa = lds[m * TSP + r], xb = lds[m2 * TSP + r];
        const int k1 = parity ? 2 * m + 1 : 2 * m, k2 = parity ? 2 * m2 + 1 : 2 * m2;
        Cx<T> s1 = xa + conj(xb), d1 = conj(tw4[k1]) * (xa - conj(xb));
        Cx<T> z1{(T)0.5 * (s1.x - d1.y), (T)0.5 * (s1.y + d1.x)};  // s/2 + (i/2) d
        lds[m * TSP + r] = z1;
        if (m2 != m && m2 < L) {
          Cx<T> s2 = xb + conj(xa), d2 = conj(tw4[k2]) * (xb - conj(xa));
          Cx<T> z2{(T)0.5 * (s2.x - d2.y), (T)0.5 * (s2.y + d2.x)};
          lds[m2 * TSP + r] = z2;
        }
      }
    }
    fft_run<+1, true, BPT>(v, pos, P, lds, TSP, it, tw4);
    if (parity == 0) {
#pragma unroll
      for (int i = 0; i < BPT; ++i)
#pragma unroll
        for (int m = 0; m < 8; ++m) res[i][m] = v[i][m];
    } else {
#pragma unroll
      for (int i = 0; i < BPT; ++i) {
        const int t = it.t(i);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const int n = pos[i][m];
          lds[n * TSP + t] = res[i][m] + conj(tw4[2 * n]) * v[i][m];
        }
      }
      __syncthreads();
      for (int r = wave; r < TS; r += nwaves) {
        const int y = y0 + r;
        if (y >= G.ny) continue;
        T* row = fbase + (size_t)(y + F.g) * F.px;
        for (int n = lane; n < L; n += 64) {
          Cx<T> a = lds[n * TSP + r];
          row[2 * n] = a.x;
          row[2 * n + 1] = a.y;
        }
      }
    }
  }
}

// Folded, permuted, real Green's table from the full complex spectrum of the rocFFT path.
template <typename T>
__global__ void __launch_bounds__(kThreads) k_fold_green(T* __restrict__ green, const T* __restrict__ ghat, PGeo G,
                                                         int rank, T norm) {
  const long long total = (long long)(G.nz + 1) * (G.ny + 1) * G.KXP;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int jl = (int)(i % G.KXP);
  long long r = i / G.KXP;
  const int kyc = (int)(r % (G.ny + 1)), kzc = (int)(r / (G.ny + 1));
  const int L = G.nx / 2;
  const int j = rank * G.KXL + jl;
  T val = T(0);
  if (jl < G.KXL && j <= G.nx) {
    const int kx = j <= L ? 2 * j : 2 * (j - L - 1) + 1;
    val = ghat[2 * (((long long)kzc * (2 * G.ny) + kyc) * (G.nx + 1) + kx)] * norm;
  }
  green[i] = val;
}

bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

// Sequences per tile.  Columns want >= 16 (f32) / 8 (f64) so that every row of the tile is
// a whole 128-byte line; one work item (8 points) per thread, 256..1024 threads.
int choose_ts(int L, bool f32, bool rows) {
  int want = rows ? 4 : (f32 ? 16 : 8);
  int ts = 2048 / L > want ? 2048 / L : want;
  if (ts * L / 8 > kMaxThreads) ts = kMaxThreads * 8 / L;
  if (ts < 1) ts = 1;
  return ts;
}

int ilog2(int n) {
  int s = 0;
  while ((1 << s) < n) ++s;
  return s;
}

}  // namespace

int smi_greens_spectrum_rocfft(smi_ctx* c, int gnz, int gny, int gnx, double dx, double x_range, void** ghat_out);
typedef int (*alltoall_fn)(smi_ctx*, const void*, void*, size_t);
int smi_alltoall_impl(smi_ctx* c, const void* send, void* recv, size_t bytes_per_peer);

struct PrunedState {
  PGeo G;
  FftPlan px, py, pz;
  int ts_x, ts_y, ts_z;
  void* tw[3] = {nullptr, nullptr, nullptr};  // x, y, z twiddle tables (4L complex each)
  void* bufA = nullptr;
  void* bufT = nullptr;  // == bufA on a single rank
  void* bufB = nullptr;
  void* green = nullptr;
  size_t green_bytes = 0;
};

template <typename T>
static int upload_twiddles(smi_ctx* c, int L, void** out) {
  std::vector<Cx<T>> h(4 * (size_t)L);
  for (int k = 0; k < 4 * L; ++k) {
    double a = -2.0 * M_PI * (double)k / (4.0 * (double)L);
    h[k].x = (T)cos(a);
    h[k].y = (T)sin(a);
  }
  SMI_CHECK_HIP(hipMalloc(out, h.size() * sizeof(Cx<T>)));
  SMI_CHECK_HIP(hipMemcpy(*out, h.data(), h.size() * sizeof(Cx<T>), hipMemcpyHostToDevice));
  return 0;
}

int smi_pruned_destroy(smi_poisson* p) {
  PrunedState* s = p->pruned;
  if (!s) return 0;
  for (int i = 0; i < 3; ++i) hipFree(s->tw[i]);
  if (s->bufT && s->bufT != s->bufA) hipFree(s->bufT);
  hipFree(s->bufA);
  hipFree(s->bufB);
  hipFree(s->green);
  delete s;
  p->pruned = nullptr;
  return 0;
}

template <typename T, int BPT>
static int set_lds(const void* fn, size_t bytes) {
  SMI_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return 0;
}

int smi_pruned_create(smi_poisson* p) {
  smi_ctx* c = p->ctx;
  SMI_REQUIRE(is_pow2(p->gnx) && is_pow2(p->gny) && is_pow2(p->gnz) && p->gnx >= 16 && p->gny >= 8 && p->gnz >= 8,
              "SMI_POISSON_PRUNED needs power of two grid sizes with nx >= 16, ny >= 8, nz >= 8 (got %d %d %d)",
              p->gnz, p->gny, p->gnx);
  SMI_REQUIRE(p->gnx <= 4096 && p->gny <= 2048 && p->gnz <= 2048, "SMI_POISSON_PRUNED: grid too large");
  PrunedState* s = new PrunedState();
  p->pruned = s;
  PGeo& G = s->G;
  G.nz = p->gnz;
  G.ny = p->gny;
  G.nx = p->gnx;
  G.P = c->nranks;
  G.nzl = c->nz;
  G.nc = 3;
  G.KXL = (G.nx + 1 + G.P - 1) / G.P;
  G.KXP = (G.KXL + 15) / 16 * 16;
  s->px = make_plan(G.nx / 2);
  s->py = make_plan(G.ny);
  s->pz = make_plan(G.nz);
  const bool f32 = c->dtype == SMI_F32;
  s->ts_x = choose_ts(s->px.L, f32, true);
  s->ts_y = choose_ts(s->py.L, f32, false);
  s->ts_z = choose_ts(s->pz.L, f32, false);
  const size_t cs = 2 * smi_elem_size(c);
  const size_t nA = (size_t)G.P * G.nc * G.nzl * G.ny * G.KXP;
  const size_t nB = (size_t)G.nc * G.nz * 2 * G.ny * G.KXP;
  SMI_CHECK_HIP(hipMalloc(&s->bufA, nA * cs));
  SMI_CHECK_HIP(hipMemsetAsync(s->bufA, 0, nA * cs, c->stream));
  if (G.P > 1) {
    SMI_CHECK_HIP(hipMalloc(&s->bufT, nA * cs));
    SMI_CHECK_HIP(hipMemsetAsync(s->bufT, 0, nA * cs, c->stream));
  } else {
    s->bufT = s->bufA;
  }
  SMI_CHECK_HIP(hipMalloc(&s->bufB, nB * cs));
  SMI_CHECK_HIP(hipMemsetAsync(s->bufB, 0, nB * cs, c->stream));
  SMI_DISPATCH(c, {
    SMI_CHECK(upload_twiddles<T>(c, s->px.L, &s->tw[0]));
    SMI_CHECK(upload_twiddles<T>(c, s->py.L, &s->tw[1]));
    SMI_CHECK(upload_twiddles<T>(c, s->pz.L, &s->tw[2]));
  });
  // Green's table: full spectrum once through rocFFT, then fold / permute / normalise
  void* ghat = nullptr;
  SMI_CHECK(smi_greens_spectrum_rocfft(c, G.nz, G.ny, G.nx, p->dx, p->x_range, &ghat));
  s->green_bytes = (size_t)(G.nz + 1) * (G.ny + 1) * G.KXP * smi_elem_size(c);
  SMI_CHECK_HIP(hipMalloc(&s->green, s->green_bytes));
  const long long total = (long long)(G.nz + 1) * (G.ny + 1) * G.KXP;
  const double norm = 1.0 / (4.0 * (double)G.nx * (double)G.ny * (double)G.nz);
  SMI_DISPATCH(c, {
    k_fold_green<T><<<(int)((total + kThreads - 1) / kThreads), kThreads, 0, c->stream>>>((T*)s->green, (const T*)ghat,
                                                                                         G, c->rank, (T)norm);
  });
  SMI_LAUNCH_CHECK();
  SMI_CHECK_HIP(hipStreamSynchronize(c->stream));
  SMI_CHECK_HIP(hipFree(ghat));
  return 0;
}

size_t smi_pruned_greens_size(smi_poisson* p) { return p->pruned ? p->pruned->green_bytes : 0; }

int smi_pruned_greens_download(smi_poisson* p, void* host) {
  SMI_REQUIRE(p->pruned, "pruned state missing");
  return smi_memcpy_d2h(p->ctx, host, p->pruned->green, p->pruned->green_bytes);
}

template <typename T, int BPT>
static int launch_cols(smi_ctx* c, PrunedState* s, int which, int nc) {
  // which: 0 = y_fwd, 1 = z_conv, 2 = y_inv
  PGeo G = s->G;
  G.nc = nc;
  const FftPlan& P = which == 1 ? s->pz : s->py;
  const int TS = which == 1 ? s->ts_z : s->ts_y, ts_shift = ilog2(TS), nthreads = TS * P.L / 8;
  const size_t lds_bytes = (size_t)P.L * TS * sizeof(Cx<T>);
  dim3 grid((G.KXL + TS - 1) / TS, which == 1 ? 2 * G.ny : G.nz, nc);
  const Cx<T>* tw = (const Cx<T>*)s->tw[which == 1 ? 2 : 1];
  if (which == 0) {
    SMI_CHECK((set_lds<T, BPT>((const void*)k_fft_y_fwd<T, BPT>, lds_bytes)));
    SmiProfScope prof(c, PROF_FFT_Y_FWD, 6.0 * nc * smi_S(c));
    k_fft_y_fwd<T, BPT><<<grid, nthreads, lds_bytes, c->stream>>>((const Cx<T>*)s->bufT, (Cx<T>*)s->bufB, G, P, tw,
                                                                 ts_shift);
  } else if (which == 1) {
    SMI_CHECK((set_lds<T, BPT>((const void*)k_fft_z_conv<T, BPT>, lds_bytes)));
    SmiProfScope prof(c, PROF_FFT_Z_CONV, 9.0 * nc * smi_S(c));
    k_fft_z_conv<T, BPT><<<grid, nthreads, lds_bytes, c->stream>>>((Cx<T>*)s->bufB, (const T*)s->green, G, P, tw,
                                                                  ts_shift);
  } else {
    SMI_CHECK((set_lds<T, BPT>((const void*)k_fft_y_inv<T, BPT>, lds_bytes)));
    SmiProfScope prof(c, PROF_FFT_Y_INV, 6.0 * nc * smi_S(c));
    k_fft_y_inv<T, BPT><<<grid, nthreads, lds_bytes, c->stream>>>((const Cx<T>*)s->bufB, (Cx<T>*)s->bufT, G, P, tw,
                                                                 ts_shift);
  }
  SMI_LAUNCH_CHECK();
  return 0;
}

template <typename T, int BPT>
static int launch_rows(smi_ctx* c, PrunedState* s, bool fwd, int nc, void* field) {
  PGeo G = s->G;
  G.nc = nc;
  const FftPlan& P = s->px;
  const int TS = s->ts_x, ts_shift = ilog2(TS), nthreads = TS * P.L / 8;
  const size_t lds_bytes = (size_t)(P.L + 1) * (TS + 1) * sizeof(Cx<T>);
  FieldGeo F{c->g, c->px, (long long)c->plane, (long long)c->vol};
  dim3 grid((G.ny + TS - 1) / TS, G.nzl, nc);
  const Cx<T>* tw = (const Cx<T>*)s->tw[0];
  if (fwd) {
    SMI_CHECK((set_lds<T, BPT>((const void*)k_fft_x_fwd<T, BPT>, lds_bytes)));
    SmiProfScope prof(c, PROF_FFT_X_FWD, 3.0 * nc * smi_S(c));
    k_fft_x_fwd<T, BPT><<<grid, nthreads, lds_bytes, c->stream>>>((const T*)field, (Cx<T>*)s->bufA, G, F, P, tw,
                                                                 ts_shift);
  } else {
    SMI_CHECK((set_lds<T, BPT>((const void*)k_fft_x_inv<T, BPT>, lds_bytes)));
    SmiProfScope prof(c, PROF_FFT_X_INV, 3.0 * nc * smi_S(c));
    k_fft_x_inv<T, BPT><<<grid, nthreads, lds_bytes, c->stream>>>((const Cx<T>*)s->bufA, (T*)field, G, F, P, tw,
                                                                 ts_shift);
  }
  SMI_LAUNCH_CHECK();
  return 0;
}

#define BPT_SWITCH(bpt, ...)            \
  switch (bpt) {                        \
    case 1: { constexpr int BPT = 1; __VA_ARGS__ } break; \
    default: { constexpr int BPT = 1; __VA_ARGS__ } break; \
  }

int smi_pruned_solve(smi_poisson* p, void* solution, const void* rhs, int ncomp) {
  smi_ctx* c = p->ctx;
  PrunedState* s = p->pruned;
  SMI_REQUIRE(s, "pruned state missing");
  SMI_REQUIRE(ncomp == 1 || ncomp == 3, "smi_pruned_solve: ncomp must be 1 or 3");
  const PGeo& G = s->G;
  const size_t cs = 2 * smi_elem_size(c);
  const size_t peer_bytes = (size_t)ncomp * G.nzl * G.ny * G.KXP * cs;
  SMI_DISPATCH(c, {
    BPT_SWITCH(1, SMI_CHECK((launch_rows<T, BPT>(c, s, true, ncomp, (void*)rhs)));)
    if (G.P > 1) {
      SmiProfScope prof(c, PROF_TRANSPOSE, (double)peer_bytes * G.P);
      SMI_CHECK(smi_alltoall_impl(c, s->bufA, s->bufT, peer_bytes));
    }
    BPT_SWITCH(1, SMI_CHECK((launch_cols<T, BPT>(c, s, 0, ncomp)));)
    BPT_SWITCH(1, SMI_CHECK((launch_cols<T, BPT>(c, s, 1, ncomp)));)
    BPT_SWITCH(1, SMI_CHECK((launch_cols<T, BPT>(c, s, 2, ncomp)));)
    if (G.P > 1) {
      SmiProfScope prof(c, PROF_TRANSPOSE, (double)peer_bytes * G.P);
      SMI_CHECK(smi_alltoall_impl(c, s->bufT, s->bufA, peer_bytes));
    }
    BPT_SWITCH(1, SMI_CHECK((launch_rows<T, BPT>(c, s, false, ncomp, solution)));)
  });
  return 0;
}
