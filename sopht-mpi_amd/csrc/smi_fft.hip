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

// Compile-time plan of an L = 2^LG point transform: radix-8 stages, then one radix-4 or
// radix-2 stage for the remainder.  Every index below folds to constants.
template <int LG>
struct Plan {
  static constexpr int L = 1 << LG, L8 = L / 8, N8 = LG / 3, REM = LG % 3, NST = N8 + (REM ? 1 : 0);
  static constexpr int radix(int s) { return s < N8 ? 8 : (1 << REM); }
  static constexpr int ns(int s) {  // product of the radices before stage s
    int n = 1;
    for (int i = 0; i < s; ++i) n *= radix(i);
    return n;
  }
};

// Position of output m of stage S for the work item that owns elements u + m L/8.
// Radix-R butterfly j takes inputs j + r L/R, multiplies input r by w_{Ns R}^{r k} (k = j mod
// Ns) and writes (j - k) R + k + r Ns.  With R < 8 the item runs 8/R butterflies
// j = u + q L/8, which own exactly the same eight elements (m = q + (8/R) r).
template <int LG, int S>
__device__ __forceinline__ int stage_pos(int u, int m) {
  constexpr int R = Plan<LG>::radix(S), Ns = Plan<LG>::ns(S), L8 = Plan<LG>::L8;
  if (R == 8) {
    const int k = u & (Ns - 1);
    return (u - k) * 8 + k + m * Ns;
  } else if (R == 4) {
    const int j = u + (m & 1) * L8, k = j & (Ns - 1);
    return (j - k) * 4 + k + (m >> 1) * Ns;
  } else {
    const int j = u + (m & 3) * L8, k = j & (Ns - 1);
    return (j - k) * 2 + k + (m >> 2) * Ns;
  }
}

// Per-thread stage twiddles (forward sign), loaded once per kernel from
// tw4[k] = exp(-2 pi i k / (4L)).  Radix 8 keeps w, w^2, w^4 and forms the other powers
// with four complex products.
template <typename T, int LG>
struct Twiddles {
  Cx<T> w[Plan<LG>::NST > 1 ? Plan<LG>::NST : 1][4];
};

template <typename T, int LG, int S>
__device__ __forceinline__ void load_stage_tw(Twiddles<T, LG>& tw, int u, const Cx<T>* __restrict__ tw4) {
  if constexpr (S < Plan<LG>::NST) {
    if constexpr (S > 0) {
      constexpr int R = Plan<LG>::radix(S), Ns = Plan<LG>::ns(S), L = Plan<LG>::L, L8 = Plan<LG>::L8;
      if (R == 8) {
        const int step = (L / (Ns * 8)) * 4 * (u & (Ns - 1));
        tw.w[S][0] = tw4[step];
        tw.w[S][1] = tw4[2 * step];
        tw.w[S][2] = tw4[4 * step];
      } else if (R == 4) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int step = (L / Ns) * ((u + q * L8) & (Ns - 1));
          tw.w[S][2 * q] = tw4[step];
          tw.w[S][2 * q + 1] = tw4[2 * step];
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) tw.w[S][q] = tw4[(2 * L / Ns) * ((u + q * L8) & (Ns - 1))];
      }
    }
    load_stage_tw<T, LG, S + 1>(tw, u, tw4);
  }
}

template <int SIGN, typename T>
__device__ __forceinline__ Cx<T> sgn(Cx<T> w) {
  return SIGN > 0 ? Cx<T>{w.x, -w.y} : w;
}

template <int SIGN, int LG, int S, typename T>
__device__ __forceinline__ void butterflies(Cx<T> (&v)[8], const Twiddles<T, LG>& tw) {
  constexpr int R = Plan<LG>::radix(S);
  if constexpr (R == 8) {
    if constexpr (S > 0) {
      const Cx<T> w1 = sgn<SIGN>(tw.w[S][0]), w2 = sgn<SIGN>(tw.w[S][1]), w4 = sgn<SIGN>(tw.w[S][2]);
      const Cx<T> w3 = w1 * w2, w5 = w4 * w1, w6 = w4 * w2, w7 = w4 * w3;
      v[1] = v[1] * w1;
      v[2] = v[2] * w2;
      v[3] = v[3] * w3;
      v[4] = v[4] * w4;
      v[5] = v[5] * w5;
      v[6] = v[6] * w6;
      v[7] = v[7] * w7;
    }
    dft8<SIGN>(v);
  } else if constexpr (R == 4) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if constexpr (S > 0) {
        const Cx<T> w1 = sgn<SIGN>(tw.w[S][2 * q]), w2 = sgn<SIGN>(tw.w[S][2 * q + 1]);
        v[q + 2] = v[q + 2] * w1;
        v[q + 4] = v[q + 4] * w2;
        v[q + 6] = v[q + 6] * (w1 * w2);
      }
      dft4<SIGN>(v[q], v[q + 2], v[q + 4], v[q + 6]);
    }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      Cx<T> b = v[q + 4];
      if constexpr (S > 0) b = b * sgn<SIGN>(tw.w[S][q]);
      const Cx<T> a = v[q];
      v[q] = a + b;
      v[q + 4] = a - b;
    }
  }
}

constexpr int kThreads = 256;      // pointwise helper kernels
constexpr int kMaxThreads = 1024;  // FFT kernels: one work item (8 points) per thread

#ifndef SMI_FFT_COLS_F32
#define SMI_FFT_COLS_F32 16
#endif
// Sequences per tile.  Column passes want >= 16 (f32) / 8 (f64) so that every row of the tile
// is a whole 128-byte line; TS * L / 8 threads, at most 1024.
constexpr int tile_seqs(int L, bool f32, bool rows) {
  int want = rows ? 4 : (f32 ? SMI_FFT_COLS_F32 : 8);
  int ts = 2048 / L > want ? 2048 / L : want;
  if (ts * L / 8 > kMaxThreads) ts = kMaxThreads * 8 / L;
  return ts < 1 ? 1 : ts;
}

// All stages of one batched FFT, stage S onward.  LDS tile layout: lds[pos * TSP + t].
// FIRST_FROM_LDS: stage 0 inputs are read from the tile (the caller filled it and must NOT
// have synchronised) instead of being supplied in v.  On return v holds the last stage's
// outputs (output m at position stage_pos<LG, NST-1>(u, m)); every read of the tile has
// completed (barrier), so the caller may overwrite it.
template <int SIGN, bool FIRST_FROM_LDS, int LG, int TSP, int S, typename T>
__device__ __forceinline__ void fft_run_from(Cx<T> (&v)[8], Cx<T>* lds, int t, int u, const Twiddles<T, LG>& tw) {
  if constexpr (S < Plan<LG>::NST) {
    if constexpr (S > 0 || FIRST_FROM_LDS) {
      if constexpr (S > 0) {
#pragma unroll
        for (int m = 0; m < 8; ++m) lds[stage_pos<LG, S - 1>(u, m) * TSP + t] = v[m];
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < 8; ++m) v[m] = lds[(u + m * Plan<LG>::L8) * TSP + t];
      __syncthreads();
    }
    butterflies<SIGN, LG, S>(v, tw);
    fft_run_from<SIGN, FIRST_FROM_LDS, LG, TSP, S + 1>(v, lds, t, u, tw);
  }
}
template <int SIGN, bool FIRST_FROM_LDS, int LG, int TSP, typename T>
__device__ __forceinline__ void fft_run(Cx<T> (&v)[8], Cx<T>* lds, int t, int u, const Twiddles<T, LG>& tw) {
  fft_run_from<SIGN, FIRST_FROM_LDS, LG, TSP, 0>(v, lds, t, u, tw);
}
template <int LG>
__device__ __forceinline__ int out_pos(int u, int m) {
  return stage_pos<LG, Plan<LG>::NST - 1>(u, m);
}

// Geometry of the spectral buffers.
//   bufA (exchange layout) : [rank P][comp nc][z nzl][y ny][KXP]      x_fwd out, x_inv in
//   bufT (after all-to-all): [slab P][comp nc][z nzl][y ny][KXP]      y_fwd in, y_inv out
//   bufB                   : [comp nc][z / ZB][ky' 2ny][z % ZB][KXP]   z_conv in place; blocked in z so
//                            that neither a y tile (all ky' of one z) nor a z tile (all z of one
//                            ky') strides over more than a few dozen pages
//   green                  : [ky folded ny+1][kz folded nz+1][KXP]    real
// KXL = spectral columns per rank (kx in parity-split order j = 0..nx, cut into P ranges),
// KXP = row pitch (KXL rounded up to 16 columns).
struct PGeo {
  int nz, ny, nx, nzl, P, nc, KXL, KXP, zb_shift;
};

// element offset of bufB row (c, z, ky')
__device__ __forceinline__ size_t rowB(const PGeo& G, int c, int z, int kyp) {
  const int ZB = 1 << G.zb_shift;
  return ((((size_t)c * (G.nz >> G.zb_shift) + (z >> G.zb_shift)) * (2 * G.ny) + kyp) * ZB + (z & (ZB - 1))) *
         (size_t)G.KXP;
}

#define FFT_KERNEL_PROLOGUE(ROWS)                                                  \
  constexpr int L = Plan<LG>::L, L8 = Plan<LG>::L8;                                \
  constexpr int TS = tile_seqs(L, sizeof(T) == 4, ROWS), NT = TS * L / 8;          \
  extern __shared__ char smem[];                                                   \
  Cx<T>* lds = reinterpret_cast<Cx<T>*>(smem);                                     \
  const int t = threadIdx.x & (TS - 1), u = threadIdx.x / TS;                      \
  Twiddles<T, LG> tw;                                                              \
  load_stage_tw<T, LG, 0>(tw, u, tw4);                                             \
  (void)L8; (void)NT;

// ------------------------------------------------------------------------- y forward
template <typename T, int LG>
__global__ void __launch_bounds__(tile_seqs(1 << LG, sizeof(T) == 4, false) * (1 << LG) / 8)
    k_fft_y_fwd(const Cx<T>* __restrict__ in, Cx<T>* __restrict__ out, PGeo G, const Cx<T>* __restrict__ tw4) {
  FFT_KERNEL_PROLOGUE(false)
  const int c = blockIdx.z, z = blockIdx.y, col = blockIdx.x * TS + t;
  const bool ok = col < G.KXL;
  const Cx<T>* src = in + ((((size_t)(z / G.nzl) * G.nc + c) * G.nzl + (z % G.nzl)) * G.ny) * (size_t)G.KXP + col;
  Cx<T>* dst = out + rowB(G, c, z, 0) + col;
  const size_t kystride = (size_t)G.KXP << G.zb_shift;
  Cx<T> v[8], a0[8];
  // the column is read once and kept in registers for both parities
#pragma unroll
  for (int m = 0; m < 8; ++m) a0[m] = ok ? src[(size_t)(u + m * L8) * G.KXP] : Cx<T>{0, 0};
#pragma unroll
  for (int parity = 0; parity < 2; ++parity) {
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = parity ? a0[m] * tw4[2 * (u + m * L8)] : a0[m];
    fft_run<-1, false, LG, TS>(v, lds, t, u, tw);
    if (ok) {
#pragma unroll
      for (int m = 0; m < 8; ++m) dst[(size_t)(parity * L + out_pos<LG>(u, m)) * kystride] = v[m];
    }
  }
}

// ------------------------------------------------------------------------- y inverse
template <typename T, int LG>
__global__ void __launch_bounds__(tile_seqs(1 << LG, sizeof(T) == 4, false) * (1 << LG) / 8)
    k_fft_y_inv(const Cx<T>* __restrict__ in, Cx<T>* __restrict__ out, PGeo G, const Cx<T>* __restrict__ tw4) {
  FFT_KERNEL_PROLOGUE(false)
  const int c = blockIdx.z, z = blockIdx.y, col = blockIdx.x * TS + t;
  const bool ok = col < G.KXL;
  const Cx<T>* src = in + rowB(G, c, z, 0) + col;
  const size_t kystride = (size_t)G.KXP << G.zb_shift;
  Cx<T>* dst = out + ((((size_t)(z / G.nzl) * G.nc + c) * G.nzl + (z % G.nzl)) * G.ny) * (size_t)G.KXP + col;
  Cx<T> v[8], res[8];
#pragma unroll
  for (int parity = 0; parity < 2; ++parity) {
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = ok ? src[(size_t)(parity * L + u + m * L8) * kystride] : Cx<T>{0, 0};
    fft_run<+1, false, LG, TS>(v, lds, t, u, tw);
    if (parity == 0) {
#pragma unroll
      for (int m = 0; m < 8; ++m) res[m] = v[m];
    } else if (ok) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int n = out_pos<LG>(u, m);
        dst[(size_t)n * G.KXP] = res[m] + conj(tw4[2 * n]) * v[m];
      }
    }
  }
}

// ------------------------------------------------------------------------- z convolution
// forward nz -> 2nz, multiply by the folded Green's table, inverse 2nz -> nz, in place.
template <typename T, int LG>
__global__ void __launch_bounds__(tile_seqs(1 << LG, sizeof(T) == 4, false) * (1 << LG) / 8)
    k_fft_z_conv(Cx<T>* __restrict__ buf, const T* __restrict__ green, PGeo G, const Cx<T>* __restrict__ tw4) {
  FFT_KERNEL_PROLOGUE(false)
  const int c = blockIdx.z, kyp = blockIdx.y, col = blockIdx.x * TS + t;
  const bool ok = col < G.KXL;
  Cx<T>* base = buf + col;
  auto zoff = [&](int z) { return rowB(G, c, z, kyp); };
  const int ky = 2 * (kyp % G.ny) + kyp / G.ny;
  const int kyc = ky <= G.ny ? ky : 2 * G.ny - ky;
  const T* gbase = green + (size_t)kyc * (G.nz + 1) * G.KXP + col;  // all kz of this ky are contiguous rows
  const size_t gstride = (size_t)G.KXP;
  Cx<T> v[8], res[8], a0[8];
  // the column is read once and kept in registers for both parities
#pragma unroll
  for (int m = 0; m < 8; ++m) a0[m] = ok ? base[zoff(u + m * L8)] : Cx<T>{0, 0};
#pragma unroll
  for (int pz = 0; pz < 2; ++pz) {
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = pz ? a0[m] * tw4[2 * (u + m * L8)] : a0[m];
    fft_run<-1, false, LG, TS>(v, lds, t, u, tw);
    // spectrum at kz = 2 pos + pz, times the Green's table; into the tile in natural order
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int pos = out_pos<LG>(u, m);
      const int kz = 2 * pos + pz;
      const int kzc = kz <= G.nz ? kz : 2 * G.nz - kz;
      const T gval = ok ? gbase[(size_t)kzc * gstride] : T(0);
      lds[pos * TS + t] = scale(v[m], gval);
    }
    fft_run<+1, true, LG, TS>(v, lds, t, u, tw);
    if (pz == 0) {
#pragma unroll
      for (int m = 0; m < 8; ++m) res[m] = v[m];
    } else if (ok) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int n = out_pos<LG>(u, m);
        base[zoff(n)] = res[m] + conj(tw4[2 * n]) * v[m];
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

template <typename T, int LG>
__global__ void __launch_bounds__(tile_seqs(1 << LG, sizeof(T) == 4, true) * (1 << LG) / 8)
    k_fft_x_fwd(const T* __restrict__ field, Cx<T>* __restrict__ out, PGeo G, FieldGeo F,
                const Cx<T>* __restrict__ tw4) {
  FFT_KERNEL_PROLOGUE(true)
  constexpr int TSP = TS + 1, NW = NT / 64 > 0 ? NT / 64 : 1;
  const int c = blockIdx.z, zl = blockIdx.y, y0 = blockIdx.x * TS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* fbase = field + (size_t)c * F.vol + (size_t)(zl + F.g) * F.plane + F.g;
  Cx<T> v[8];
  for (int parity = 0; parity < 2; ++parity) {
    for (int r = wave; r < TS; r += NW) {
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
    fft_run<-1, true, LG, TSP>(v, lds, t, u, tw);
#pragma unroll
    for (int m = 0; m < 8; ++m) lds[out_pos<LG>(u, m) * TSP + t] = v[m];
    __syncthreads();
    const int cnt = L + 1 - parity;
    for (int r = wave; r < TS; r += NW) {
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
template <typename T, int LG>
__global__ void __launch_bounds__(tile_seqs(1 << LG, sizeof(T) == 4, true) * (1 << LG) / 8)
    k_fft_x_inv(const Cx<T>* __restrict__ in, T* __restrict__ field, PGeo G, FieldGeo F,
                const Cx<T>* __restrict__ tw4) {
  FFT_KERNEL_PROLOGUE(true)
  constexpr int TSP = TS + 1, NW = NT / 64 > 0 ? NT / 64 : 1;
  const int c = blockIdx.z, zl = blockIdx.y, y0 = blockIdx.x * TS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T* fbase = field + (size_t)c * F.vol + (size_t)(zl + F.g) * F.plane + F.g;
  Cx<T> v[8], res[8];
  for (int parity = 0; parity < 2; ++parity) {
    const int cnt = L + 1 - parity;
    for (int r = wave; r < TS; r += NW) {
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
    for (int r = wave; r < TS; r += NW) {
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
    fft_run<+1, true, LG, TSP>(v, lds, t, u, tw);
    if (parity == 0) {
#pragma unroll
      for (int m = 0; m < 8; ++m) res[m] = v[m];
    } else {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int n = out_pos<LG>(u, m);
        lds[n * TSP + t] = res[m] + conj(tw4[2 * n]) * v[m];
      }
      __syncthreads();
      for (int r = wave; r < TS; r += NW) {
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
  const int kzc = (int)(r % (G.nz + 1)), kyc = (int)(r / (G.nz + 1));
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
  int lgx, lgy, lgz;  // log2 of the transform lengths nx/2, ny, nz
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
  G.zb_shift = G.nz >= 16 ? 4 : ilog2(G.nz);
  s->lgx = ilog2(G.nx / 2);
  s->lgy = ilog2(G.ny);
  s->lgz = ilog2(G.nz);
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
    SMI_CHECK(upload_twiddles<T>(c, G.nx / 2, &s->tw[0]));
    SMI_CHECK(upload_twiddles<T>(c, G.ny, &s->tw[1]));
    SMI_CHECK(upload_twiddles<T>(c, G.nz, &s->tw[2]));
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

template <typename T, int LG>
static int launch_cols(smi_ctx* c, PrunedState* s, int which, int nc) {
  // which: 0 = y_fwd, 1 = z_conv, 2 = y_inv
  PGeo G = s->G;
  G.nc = nc;
  constexpr int L = 1 << LG, TS = tile_seqs(L, sizeof(T) == 4, false), NT = TS * L / 8;
  const size_t lds_bytes = (size_t)L * TS * sizeof(Cx<T>);
  dim3 grid((G.KXL + TS - 1) / TS, which == 1 ? 2 * G.ny : G.nz, nc);
  const Cx<T>* tw = (const Cx<T>*)s->tw[which == 1 ? 2 : 1];
  if (which == 0) {
    SMI_CHECK(set_lds((const void*)k_fft_y_fwd<T, LG>, lds_bytes));
    SmiProfScope prof(c, PROF_FFT_Y_FWD, 6.0 * nc * smi_S(c));
    k_fft_y_fwd<T, LG><<<grid, NT, lds_bytes, c->stream>>>((const Cx<T>*)s->bufT, (Cx<T>*)s->bufB, G, tw);
  } else if (which == 1) {
    SMI_CHECK(set_lds((const void*)k_fft_z_conv<T, LG>, lds_bytes));
    SmiProfScope prof(c, PROF_FFT_Z_CONV, 9.0 * nc * smi_S(c));
    k_fft_z_conv<T, LG><<<grid, NT, lds_bytes, c->stream>>>((Cx<T>*)s->bufB, (const T*)s->green, G, tw);
  } else {
    SMI_CHECK(set_lds((const void*)k_fft_y_inv<T, LG>, lds_bytes));
    SmiProfScope prof(c, PROF_FFT_Y_INV, 6.0 * nc * smi_S(c));
    k_fft_y_inv<T, LG><<<grid, NT, lds_bytes, c->stream>>>((const Cx<T>*)s->bufB, (Cx<T>*)s->bufT, G, tw);
  }
  SMI_LAUNCH_CHECK();
  return 0;
}

template <typename T, int LG>
static int launch_rows(smi_ctx* c, PrunedState* s, bool fwd, int nc, void* field) {
  PGeo G = s->G;
  G.nc = nc;
  constexpr int L = 1 << LG, TS = tile_seqs(L, sizeof(T) == 4, true), NT = TS * L / 8;
  const size_t lds_bytes = (size_t)(L + 1) * (TS + 1) * sizeof(Cx<T>);
  FieldGeo F{c->g, c->px, (long long)c->plane, (long long)c->vol};
  dim3 grid((G.ny + TS - 1) / TS, G.nzl, nc);
  const Cx<T>* tw = (const Cx<T>*)s->tw[0];
  if (fwd) {
    SMI_CHECK(set_lds((const void*)k_fft_x_fwd<T, LG>, lds_bytes));
    SmiProfScope prof(c, PROF_FFT_X_FWD, 3.0 * nc * smi_S(c));
    k_fft_x_fwd<T, LG><<<grid, NT, lds_bytes, c->stream>>>((const T*)field, (Cx<T>*)s->bufA, G, F, tw);
  } else {
    SMI_CHECK(set_lds((const void*)k_fft_x_inv<T, LG>, lds_bytes));
    SmiProfScope prof(c, PROF_FFT_X_INV, 3.0 * nc * smi_S(c));
    k_fft_x_inv<T, LG><<<grid, NT, lds_bytes, c->stream>>>((const Cx<T>*)s->bufA, (T*)field, G, F, tw);
  }
  SMI_LAUNCH_CHECK();
  return 0;
}

#define LG_CASE(n, ...) case n: { constexpr int LG = n; __VA_ARGS__ } break;
#define LG_SWITCH(lg, ...)                                                                      \
  switch (lg) {                                                                                 \
    LG_CASE(3, __VA_ARGS__) LG_CASE(4, __VA_ARGS__) LG_CASE(5, __VA_ARGS__) LG_CASE(6, __VA_ARGS__) \
    LG_CASE(7, __VA_ARGS__) LG_CASE(8, __VA_ARGS__) LG_CASE(9, __VA_ARGS__) LG_CASE(10, __VA_ARGS__) \
    LG_CASE(11, __VA_ARGS__)                                                                    \
    default: smi_set_error("unsupported transform length 2^%d", lg); return 1;                  \
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
    LG_SWITCH(s->lgx, SMI_CHECK((launch_rows<T, LG>(c, s, true, ncomp, (void*)rhs)));)
    if (G.P > 1) {
      SmiProfScope prof(c, PROF_TRANSPOSE, (double)peer_bytes * G.P);
      SMI_CHECK(smi_alltoall_impl(c, s->bufA, s->bufT, peer_bytes));
    }
    LG_SWITCH(s->lgy, SMI_CHECK((launch_cols<T, LG>(c, s, 0, ncomp)));)
    LG_SWITCH(s->lgz, SMI_CHECK((launch_cols<T, LG>(c, s, 1, ncomp)));)
    LG_SWITCH(s->lgy, SMI_CHECK((launch_cols<T, LG>(c, s, 2, ncomp)));)
    if (G.P > 1) {
      SmiProfScope prof(c, PROF_TRANSPOSE, (double)peer_bytes * G.P);
      SMI_CHECK(smi_alltoall_impl(c, s->bufT, s->bufA, peer_bytes));
    }
    LG_SWITCH(s->lgx, SMI_CHECK((launch_rows<T, LG>(c, s, false, ncomp, solution)));)
  });
  return 0;
}
