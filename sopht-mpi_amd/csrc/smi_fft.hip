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
// That is how the y and z axes are handled.  Along x, where the rows are real, the 2nx-point transform is taken at
// the half-sample-shifted frequencies k + 1/2 instead: a real row then has exactly nx independent complex
// coefficients (those at even k), all obtained from ONE nx-point transform of the row times e^{-i pi n / 2nx}, and
// the Green's function enters through the shifted transform of its negacyclic extension (see k_fft_x_fwd).
//
// Five kernels per solve, all batched 1D FFTs held in LDS (one tile = TS sequences of L
// points; Stockham radix-8/4/2 stages, eight points per work item in registers):
//   x_fwd   real rows (read straight from the ghosted field) -> nx complex per row
//   y_fwd   columns ny -> 2ny
//   z_conv  columns nz -> 2nz, times the Green's spectrum, 2nz -> nz   (one kernel, in place)
//   y_inv   columns 2ny -> ny
//   x_inv   complex rows -> real rows written straight into the ghosted solution field
// The y and z spectral indices are stored split by parity ([even | odd]); the Green's table is laid out in the
// same order, folded by its even symmetry in y and z and pre-multiplied by dx^3 and by the backward
// normalisation, so it is real and (nz+1)(ny+1) nx in size.
//
// Slabs: after x_fwd each rank holds its z planes for all kx; one all-to-all hands every
// rank a contiguous kx range for ALL z (before any y / z padding exists), y and z passes
// run locally, a second all-to-all goes back.  x_fwd writes and x_inv reads the exchange
// layout directly, so there is no pack / unpack pass.
#include <cstdlib>
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
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
// multiply by SIGN * i  (SIGN = -1: forward transform, +1: inverse)
template <int SIGN, typename T>
__device__ __forceinline__ Cx<T> mul_si(Cx<T> a) {
  return SIGN < 0 ? Cx<T>{a.y, -a.x} : Cx<T>{-a.y, a.x};
}

// Stores of the spectral passes: every output line is written once and not read again before the
// next pass has streamed gigabytes through the caches.
#ifndef SMI_FFT_NT_STORE
#define SMI_FFT_NT_STORE 1
#endif
template <typename T>
__device__ __forceinline__ void stream_store(Cx<T>* p, Cx<T> v) {
#if SMI_FFT_NT_STORE
  typedef T vec2 __attribute__((ext_vector_type(2)));
  vec2 w = {v.x, v.y};
  __builtin_nontemporal_store(w, reinterpret_cast<vec2*>(p));
#else
  *p = v;
#endif
}

// ... and the column passes read every input line once (the row pass k_fft_x_inv reads each line
// twice, through its mirror columns, and was 10 % slower with streaming loads).  Measured at
// 512^3: y_inv 2.10 -> 2.01 ms, z_conv 4.00 -> 3.96 ms, y_fwd 2.01 -> 1.99 ms.
#ifndef SMI_FFT_NT_LOAD
#define SMI_FFT_NT_LOAD 1
#endif
template <typename T>
__device__ __forceinline__ Cx<T> stream_load(const Cx<T>* p) {
#if SMI_FFT_NT_LOAD
  typedef T vec2 __attribute__((ext_vector_type(2)));
  const vec2 w = __builtin_nontemporal_load(reinterpret_cast<const vec2*>(p));
  return Cx<T>{w.x, w.y};
#else
  return *p;
#endif
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
  // w8^k, w8 = exp(SIGN * 2 pi i / 8): o1 w8 = h p1, o3 w8^3 = h p3, folded into the final
  // additions as fused multiply-adds (the products have two uses each, which the compiler's
  // contraction leaves alone)
  const Cx<T> p1 = SIGN < 0 ? Cx<T>{o1.x + o1.y, o1.y - o1.x} : Cx<T>{o1.x - o1.y, o1.y + o1.x};
  const Cx<T> p3 = SIGN < 0 ? Cx<T>{o3.y - o3.x, -o3.x - o3.y} : Cx<T>{-o3.x - o3.y, o3.x - o3.y};
  const Cx<T> t2 = mul_si<SIGN>(o2);
  v[0] = e0 + o0;
  v[4] = e0 - o0;
  v[1] = Cx<T>{fma_(h, p1.x, e1.x), fma_(h, p1.y, e1.y)};
  v[5] = Cx<T>{fma_(-h, p1.x, e1.x), fma_(-h, p1.y, e1.y)};
  v[2] = e2 + t2;
  v[6] = e2 - t2;
  v[3] = Cx<T>{fma_(h, p3.x, e3.x), fma_(h, p3.y, e3.y)};
  v[7] = Cx<T>{fma_(-h, p3.x, e3.x), fma_(-h, p3.y, e3.y)};
}

// Compile-time plan of an L point transform.  The template key LG is log2(L) for powers of
// two (radix-8 stages, then one radix-4 or radix-2 stage for the remainder) and L itself for
// L = 64 R with R = 3, 5, 6, 10, 12 (192, 320, 384, 640, 768): three stages of radix 8, R, 8.  Every index
// below folds to constants.
template <int LG>
struct Plan {
  static constexpr bool POW2 = LG < 32;
  static constexpr int L = POW2 ? (1 << (POW2 ? LG : 0)) : LG, L8 = L / 8;
  static constexpr int RM = POW2 ? 0 : L / 64;  // middle radix of the 8 x R x 8 plans
  static_assert(POW2 || (L % 64 == 0 && (RM == 3 || RM == 5 || RM == 6 || RM == 10 || RM == 12)),
                "unsupported transform length");
  static constexpr int N8 = POW2 ? LG / 3 : 0, REM = POW2 ? LG % 3 : 0, NST = POW2 ? N8 + (REM ? 1 : 0) : 3;
  // threads per sequence: one per work item of 8 points; the 64 items of the middle stage of a
  // mixed plan need 64 lanes, or 32 that take MI = 2 items each when those are three points wide
  static constexpr int TPS = POW2 ? L8 : (L8 > 64 ? L8 : (RM == 3 ? 32 : 64));
  static constexpr int MI = POW2 ? 1 : (TPS < 64 ? 64 / TPS : 1);
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
// Plans of four stages (L >= 1024) would hold ten complex stage twiddles per lane on top of the
// transform's own registers; they keep them in an LDS table instead, indexed by k = j mod Ns:
// radix 8 -> (w, w^2, w^4), radix 4 -> (w, w^2), radix 2 -> w, with w = w_{Ns R}^k.
template <int LG>
struct TwTab {
  static constexpr bool USE = Plan<LG>::POW2 && LG >= 10;
  static constexpr int per(int s) { return Plan<LG>::radix(s) == 8 ? 3 : (Plan<LG>::radix(s) == 4 ? 2 : 1); }
  static constexpr int off(int s) {  // first entry of stage s
    int o = 0;
    for (int i = 1; i < s; ++i) o += per(i) * Plan<LG>::ns(i);
    return o;
  }
  static constexpr int SIZE = USE ? off(Plan<LG>::NST) : 0;
  // index into tw4[k] = exp(-2 pi i k / 4L) of w_{Ns R}: 4 L / (Ns R)
  static constexpr int unit(int s) { return 4 * Plan<LG>::L / (Plan<LG>::ns(s) * Plan<LG>::radix(s)); }
};

template <typename T, int LG>
struct Twiddles {
  Cx<T> w[Plan<LG>::NST > 1 ? Plan<LG>::NST : 1][4];
  const Cx<T>* mid;  // mixed plans: w_{8R}^{q k} = mid[q * mid_step], read from the table at each use
  int mid_step;
  const Cx<T>* tab;  // TwTab plans: the LDS table
};

// Fills the LDS table; the first barrier of the first transform publishes it (stage 0 has no twiddles).
template <typename T, int LG, int S>
__device__ __forceinline__ void fill_tw_tab(Cx<T>* tab, const Cx<T>* __restrict__ tw4) {
  if constexpr (TwTab<LG>::USE && S < Plan<LG>::NST) {
    constexpr int per = TwTab<LG>::per(S), n = per * Plan<LG>::ns(S);
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
      const int k = e / per, j = e - k * per;
      tab[TwTab<LG>::off(S) + e] = tw4[(TwTab<LG>::unit(S) * k) << j];
    }
    fill_tw_tab<T, LG, S + 1>(tab, tw4);
  }
}

template <typename T, int LG, int S>
__device__ __forceinline__ void load_stage_tw(Twiddles<T, LG>& tw, int u, const Cx<T>* __restrict__ tw4) {
  if constexpr (!Plan<LG>::POW2) {
    // middle stage (Ns = 8): w_{8R}^{q k} = tw4[32 q k], k = j mod 8; last stage (Ns = L/8): w_L^u and
    // its 2nd and 4th powers (idle lanes u >= L/8 read a valid entry)
    const int ul = u < Plan<LG>::L8 ? u : 0;
    tw.mid = tw4;
    tw.mid_step = 32 * (u & 7);
    tw.w[2][0] = tw4[4 * ul];
    tw.w[2][1] = tw4[8 * ul];
    tw.w[2][2] = tw4[16 * ul];
  } else if constexpr (TwTab<LG>::USE) {
    // stage twiddles are read from the LDS table at each use
  } else if constexpr (S < Plan<LG>::NST) {
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

// stage twiddle j of butterfly item q (registers, or the LDS table of the four-stage plans)
template <int LG, int S, typename T>
__device__ __forceinline__ Cx<T> stage_tw(const Twiddles<T, LG>& tw, int u, int q, int j) {
  if constexpr (TwTab<LG>::USE) {
    constexpr int Ns = Plan<LG>::ns(S), per = TwTab<LG>::per(S);
    const int k = (u + q * Plan<LG>::L8) & (Ns - 1);
    return tw.tab[TwTab<LG>::off(S) + per * k + j];
  } else {
    constexpr int R = Plan<LG>::radix(S);
    return tw.w[S][R == 8 ? j : (R == 4 ? 2 * q + j : q)];
  }
}

template <int SIGN, int LG, int S, typename T>
__device__ __forceinline__ void butterflies(Cx<T> (&v)[8], const Twiddles<T, LG>& tw, int u) {
  constexpr int R = Plan<LG>::radix(S);
  if constexpr (R == 8) {
    if constexpr (S > 0) {
      const Cx<T> w1 = sgn<SIGN>(stage_tw<LG, S>(tw, u, 0, 0)), w2 = sgn<SIGN>(stage_tw<LG, S>(tw, u, 0, 1)),
                  w4 = sgn<SIGN>(stage_tw<LG, S>(tw, u, 0, 2));
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
        const Cx<T> w1 = sgn<SIGN>(stage_tw<LG, S>(tw, u, q, 0)), w2 = sgn<SIGN>(stage_tw<LG, S>(tw, u, q, 1));
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
      if constexpr (S > 0) b = b * sgn<SIGN>(stage_tw<LG, S>(tw, u, q, 0));
      const Cx<T> a = v[q];
      v[q] = a + b;
      v[q + 4] = a - b;
    }
  }
}

constexpr int kThreads = 256;      // pointwise helper kernels
constexpr int kMaxThreads = 1024;  // FFT kernels: one work item (8 points) per thread

#ifndef SMI_FFT_ZCOLS
#define SMI_FFT_ZCOLS 0
#endif
// mixed plans with a middle radix above this run the z convolution in its lean form (see k_fft_z_conv).  Measured with
// 6 (the 192-, 320- and 384-point plans in the full form, which then spills or exceeds 128 registers): z convolution
// 2.42 against 1.95 ms at 384^3, 1.29 against 1.19 ms at 320^3, 0.29 against 0.22 ms at 192^3, so every mixed plan is lean
#ifndef SMI_FFT_LEAN_ABOVE
#define SMI_FFT_LEAN_ABOVE 0
#endif
// Row tiles one workgroup of the x kernels walks over (the twiddle set-up is then paid once for all of them) and
// rows per tile.  Measured at 512^3 (512-point plan): two tiles against one take x_fwd from 0.95 to 0.93 ms and x_inv
// from 0.98 to 0.94 ms; eight rows per tile instead of four, or one or two rows and four or eight tiles, change nothing
// or cost 1-5 %.  The mixed plans of up to 64 lanes per sequence, whose radix-8 stages leave a quarter or more of those
// lanes idle, want the opposite: ONE wave per workgroup (a single row, two for the 32-lane 192-point plan; the barriers
// of its exchanges then synchronise nobody) walking over four (forward) or eight (inverse) rows.  Measured at 384^3
// against eight rows per tile and one tile: x_fwd 0.61 -> 0.46 ms, x_inv 0.56 -> 0.38 ms (four rows x two tiles: 0.49 /
// 0.49; two rows x four tiles: 0.45 / 0.45); at 320^3 0.33 / 0.34 -> 0.28 / 0.22; at 192^3 0.069 / 0.083 -> 0.051 / 0.061.
// The 640- and 768-point plans (80 and 96 lanes per sequence, so a row is not one wave) keep four rows and one tile:
// single rows measured 2.96 / 2.52 against 2.48 / 2.57 ms at 640^3 and 4.24 / 5.28 against 5.15 / 4.75 ms at 768^3.
#ifndef SMI_FFT_XT_FWD
#define SMI_FFT_XT_FWD 2
#endif
#ifndef SMI_FFT_XT_INV
#define SMI_FFT_XT_INV 2
#endif
#ifndef SMI_FFT_XT_FWD_MIXED
#define SMI_FFT_XT_FWD_MIXED 4
#endif
#ifndef SMI_FFT_XT_INV_MIXED
#define SMI_FFT_XT_INV_MIXED 8
#endif
#ifndef SMI_FFT_XROWS
#define SMI_FFT_XROWS 4
#endif
template <int LG>
constexpr int x_tiles(bool fwd) {
  if (!Plan<LG>::POW2) return Plan<LG>::TPS <= 64 ? (fwd ? SMI_FFT_XT_FWD_MIXED : SMI_FFT_XT_INV_MIXED) : 1;
  return LG >= 8 ? (fwd ? SMI_FFT_XT_FWD : SMI_FFT_XT_INV) : 1;
}
#ifndef SMI_FFT_COLS_F32
#define SMI_FFT_COLS_F32 16
#endif
// Sequences per tile.  Column passes want >= 16 (f32) / 8 (f64) so that every row of the tile
// is a whole 128-byte line; TS * L / 8 threads, at most 1024.
constexpr int tile_seqs(int L, int tps, bool f32, bool rows, int want_cols = 0, bool pow2 = true) {
  if (rows && !pow2 && tps <= 64) return 64 / tps;  // one wave per row tile
  int want = rows ? SMI_FFT_XROWS : (want_cols ? want_cols : (f32 ? SMI_FFT_COLS_F32 : 8));
  int ts = want;
  while (ts * L < 2048) ts *= 2;
  // double precision kernels hold twice the registers per lane: 512 lanes keep them out of scratch
  const int max_threads = f32 ? kMaxThreads : kMaxThreads / 2;
  while (ts > 1 && ts * tps > max_threads) ts /= 2;
  return ts;
}

// LDS tile layouts.  Column passes: lds[pos * TSP + t], lanes run along the sequences of the
// tile.  Row passes: one padded row per sequence, lds[t * PITCH + pos + pos / 8], lanes run
// along the work items of a sequence; the pos / 8 skew spreads the radix-8 strided writes of
// the autosort stages over the banks.
// TPS_: lanes per sequence of the kernel that uses the layout (the mixed plans deal their middle-stage items by it).
template <int TSP, int TPS_>
struct ColLayout {
  static constexpr int TPS = TPS_;
  static constexpr __host__ __device__ __forceinline__ int idx(int pos, int t) { return pos * TSP + t; }
};
template <int PITCH, int TPS_>
struct RowLayout {
  static constexpr int TPS = TPS_;
  static constexpr __host__ __device__ __forceinline__ int idx(int pos, int t) { return t * PITCH + pos + (pos >> 3); }
};

// Hands out the tile for the next exchange.  With two tiles (PP) consecutive exchanges
// alternate between them, so a wave may start writing the next exchange while slower waves
// still read the previous one: one barrier per exchange.  With one tile a second barrier
// after the reads protects it.
template <typename T, bool PP>
struct TileSel {
  Cx<T>* base;
  Cx<T>* cur;
  int tile;   // elements per tile
  int phase;
  __device__ __forceinline__ TileSel(Cx<T>* lds, int tile_elems) : base(lds), cur(lds), tile(tile_elems), phase(0) {}
  __device__ __forceinline__ Cx<T>* next() {
    cur = base + ((PP && phase) ? tile : 0);
    phase ^= 1;
    return cur;
  }
  __device__ __forceinline__ void after_read() {
    if (!PP) __syncthreads();
  }
};

// All stages of one batched FFT, stage S onward.  FIRST_FROM_LDS: stage 0 inputs are read
// from the tile the caller obtained with sel.next() and filled (the caller must NOT have
// synchronised) instead of being supplied in v.  On return v[m] holds element u + m L/8 of
// the transform (the last stage of every plan lands in natural order).
template <int SIGN, bool FIRST_FROM_LDS, int LG, typename LAY, int S, typename T, bool PP>
__device__ __forceinline__ void fft_run_from(Cx<T> (&v)[8], TileSel<T, PP>& sel, int t, int u,
                                             const Twiddles<T, LG>& tw) {
  if constexpr (S < Plan<LG>::NST) {
    if constexpr (S > 0 || FIRST_FROM_LDS) {
      if constexpr (S > 0) {
        Cx<T>* w = sel.next();
#pragma unroll
        for (int m = 0; m < 8; ++m) w[LAY::idx(stage_pos<LG, S - 1>(u, m), t)] = v[m];
      }
      __syncthreads();
      const Cx<T>* r = sel.cur;
#pragma unroll
      for (int m = 0; m < 8; ++m) v[m] = r[LAY::idx(u + m * Plan<LG>::L8, t)];
      sel.after_read();
    }
    butterflies<SIGN, LG, S>(v, tw, u);
    fft_run_from<SIGN, FIRST_FROM_LDS, LG, LAY, S + 1>(v, sel, t, u, tw);
  }
}
template <int SIGN, bool FIRST_FROM_LDS, int LG, typename LAY, typename T, bool PP>
__device__ __forceinline__ void fft_run(Cx<T> (&v)[8], TileSel<T, PP>& sel, int t, int u, const Twiddles<T, LG>& tw);

// ---- mixed 8 x R x 8 plans ------------------------------------------------------------
template <int SIGN, typename T>
__device__ __forceinline__ void dft3(Cx<T>& a, Cx<T>& b, Cx<T>& c) {
  const T h = (T)(SIGN * 0.86602540378443864676);  // SIGN sin(2 pi / 3)
  const Cx<T> t = b + c, d = b - c, m{a.x - (T)0.5 * t.x, a.y - (T)0.5 * t.y};
  const Cx<T> id{-h * d.y, h * d.x};  // i h d
  a = a + t;
  b = m + id;
  c = m - id;
}

template <int SIGN, typename T>
__device__ __forceinline__ void dft5(Cx<T>& x0, Cx<T>& x1, Cx<T>& x2, Cx<T>& x3, Cx<T>& x4) {
  const T c1 = (T)0.30901699437494742410, c2 = (T)-0.80901699437494742410;              // cos(2 pi / 5), cos(4 pi / 5)
  const T s1 = (T)(SIGN * 0.95105651629515357212), s2 = (T)(SIGN * 0.58778525229247312917);  // SIGN sin(...)
  const Cx<T> t1 = x1 + x4, t2 = x2 + x3, t3 = x1 - x4, t4 = x2 - x3;
  const Cx<T> a1{x0.x + c1 * t1.x + c2 * t2.x, x0.y + c1 * t1.y + c2 * t2.y};
  const Cx<T> a2{x0.x + c2 * t1.x + c1 * t2.x, x0.y + c2 * t1.y + c1 * t2.y};
  const Cx<T> b1{s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y}, b2{s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y};
  const Cx<T> ib1{-b1.y, b1.x}, ib2{-b2.y, b2.x};  // i b
  x0 = x0 + t1 + t2;
  x1 = a1 + ib1;
  x4 = a1 - ib1;
  x2 = a2 + ib2;
  x3 = a2 - ib2;
}

// x[0..R-1] -> DFT_R in place (natural order in, natural order out), R = 3, 5, 6, 10, 12
template <int SIGN, int R, typename T>
__device__ __forceinline__ void dft_mid(Cx<T> (&x)[12]) {
  if constexpr (R == 3) {
    dft3<SIGN>(x[0], x[1], x[2]);
  } else if constexpr (R == 5) {
    dft5<SIGN>(x[0], x[1], x[2], x[3], x[4]);
  } else if constexpr (R == 10) {
    // n = 2 n1 + n2, k = k1 + 5 k2
    Cx<T> e[5] = {x[0], x[2], x[4], x[6], x[8]}, o[5] = {x[1], x[3], x[5], x[7], x[9]};
    dft5<SIGN>(e[0], e[1], e[2], e[3], e[4]);
    dft5<SIGN>(o[0], o[1], o[2], o[3], o[4]);
    const T s = (T)SIGN;
    o[1] = o[1] * Cx<T>{(T)0.80901699437494742410, s * (T)0.58778525229247312917};    // w10
    o[2] = o[2] * Cx<T>{(T)0.30901699437494742410, s * (T)0.95105651629515357212};    // w10^2
    o[3] = o[3] * Cx<T>{(T)-0.30901699437494742410, s * (T)0.95105651629515357212};   // w10^3
    o[4] = o[4] * Cx<T>{(T)-0.80901699437494742410, s * (T)0.58778525229247312917};   // w10^4
#pragma unroll
    for (int k1 = 0; k1 < 5; ++k1) {
      x[k1] = e[k1] + o[k1];
      x[k1 + 5] = e[k1] - o[k1];
    }
  } else if constexpr (R == 6) {
    // n = 2 n1 + n2, k = k1 + 3 k2
    Cx<T> e0 = x[0], e1 = x[2], e2 = x[4], o0 = x[1], o1 = x[3], o2 = x[5];
    dft3<SIGN>(e0, e1, e2);
    dft3<SIGN>(o0, o1, o2);
    const T h = (T)(SIGN * 0.86602540378443864676);
    o1 = o1 * Cx<T>{(T)0.5, h};    // w6
    o2 = o2 * Cx<T>{(T)-0.5, h};   // w6^2
    x[0] = e0 + o0;
    x[3] = e0 - o0;
    x[1] = e1 + o1;
    x[4] = e1 - o1;
    x[2] = e2 + o2;
    x[5] = e2 - o2;
  } else {
    // n = 4 n1 + n2, k = k1 + 3 k2: four 3-point transforms, twiddles w12^{n2 k1}, three 4-point ones
    Cx<T> a[4][3];
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) {
      a[n2][0] = x[n2];
      a[n2][1] = x[4 + n2];
      a[n2][2] = x[8 + n2];
      dft3<SIGN>(a[n2][0], a[n2][1], a[n2][2]);
    }
    const T s = (T)SIGN, c30 = (T)0.86602540378443864676;
    a[1][1] = a[1][1] * Cx<T>{c30, s * (T)0.5};          // w12
    a[1][2] = a[1][2] * Cx<T>{(T)0.5, s * c30};          // w12^2
    a[2][1] = a[2][1] * Cx<T>{(T)0.5, s * c30};          // w12^2
    a[2][2] = a[2][2] * Cx<T>{(T)-0.5, s * c30};         // w12^4
    a[3][1] = mul_si<SIGN>(a[3][1]);                     // w12^3 = SIGN i
    a[3][2] = Cx<T>{-a[3][2].x, -a[3][2].y};             // w12^6 = -1
#pragma unroll
    for (int k1 = 0; k1 < 3; ++k1) {
      dft4<SIGN>(a[0][k1], a[1][k1], a[2][k1], a[3][k1]);
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) x[k1 + 3 * k2] = a[k2][k1];
    }
  }
}

// Radix-8 stage on work items u < L/8 (no twiddles: Ns = 1), exchange, radix-R stage on
// items j < 64 (item j owns j + 64 r, twiddle w_{8R}^{r (j mod 8)}, writes (j - k) R + k + 8 q),
// exchange, radix-8 stage with twiddles w_L^{m u}.  Every lane takes both barriers.
// The 64 middle-stage items of a sequence are dealt to its LAY::TPS lanes: lane u takes items u, u + TPS, ... below
// 64 (one each with 64 or more lanes; two each with 32; with L/8 = 40 or 48 lanes - no lane idle in the radix-8
// stages - the first 24 or 16 lanes take a second one).
template <int SIGN, int LG, typename LAY, typename T, bool PP>
__device__ __forceinline__ void fft_run_mixed(Cx<T> (&v)[8], TileSel<T, PP>& sel, int t, int u,
                                              const Twiddles<T, LG>& tw) {
  constexpr int L8 = Plan<LG>::L8, R = Plan<LG>::RM, TPS = LAY::TPS;
  constexpr int MI = TPS < 64 ? (64 + TPS - 1) / TPS : 1;
  static_assert(MI * R <= 12, "middle-stage items of a lane must fit x[]");
  static_assert(TPS >= 64 || TPS % 8 == 0, "items of one lane must share their middle-stage twiddles");
  const bool a8 = u < L8;
  if (a8) dft8<SIGN>(v);
  Cx<T>* w = sel.next();
  if (a8) {
#pragma unroll
    for (int m = 0; m < 8; ++m) w[LAY::idx(u * 8 + m, t)] = v[m];
  }
  __syncthreads();
  Cx<T> x[12];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    if (u + i * TPS < 64) {
      const Cx<T>* r = sel.cur;
#pragma unroll
      for (int q = 0; q < R; ++q) x[i * R + q] = r[LAY::idx(u + i * TPS + 64 * q, t)];
    }
  }
  sel.after_read();
  w = sel.next();
  {
    // table values rather than running powers: eleven chained products would cost accuracy
    // (items u and u + TPS share u mod 8, hence their twiddles)
    const int k = u & 7;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if (u + i * TPS >= 64) continue;
      Cx<T> y[12];
      y[0] = x[i * R];
#pragma unroll
      for (int q = 1; q < R; ++q) y[q] = x[i * R + q] * sgn<SIGN>(tw.mid[q * tw.mid_step]);
      dft_mid<SIGN, R>(y);
      const int j = u + i * TPS;
#pragma unroll
      for (int q = 0; q < R; ++q) w[LAY::idx((j - k) * R + k + 8 * q, t)] = y[q];
    }
  }
  __syncthreads();
  if (a8) {
    const Cx<T>* r = sel.cur;
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = r[LAY::idx(u + m * L8, t)];
  }
  sel.after_read();
  if (a8) {
    const Cx<T> w1 = sgn<SIGN>(tw.w[2][0]), w2 = sgn<SIGN>(tw.w[2][1]), w4 = sgn<SIGN>(tw.w[2][2]);
    const Cx<T> w3 = w1 * w2, w5 = w4 * w1, w6 = w4 * w2, w7 = w4 * w3;
    v[1] = v[1] * w1;
    v[2] = v[2] * w2;
    v[3] = v[3] * w3;
    v[4] = v[4] * w4;
    v[5] = v[5] * w5;
    v[6] = v[6] * w6;
    v[7] = v[7] * w7;
    dft8<SIGN>(v);
  }
}

template <int SIGN, bool FIRST_FROM_LDS, int LG, typename LAY, typename T, bool PP>
__device__ __forceinline__ void fft_run(Cx<T> (&v)[8], TileSel<T, PP>& sel, int t, int u, const Twiddles<T, LG>& tw) {
  if constexpr (Plan<LG>::POW2) {
    fft_run_from<SIGN, FIRST_FROM_LDS, LG, LAY, 0>(v, sel, t, u, tw);
  } else {
    static_assert(!FIRST_FROM_LDS, "mixed plans take their input in registers");
    fft_run_mixed<SIGN, LG, LAY>(v, sel, t, u, tw);
  }
}

// w * exp(-i pi m / 8): the zero-padding twiddle exp(-i pi n / L) of element n = u + m L/8 is
// the per-thread factor exp(-i pi u / L) times this constant, so one table load serves all
// eight elements of a work item.
template <typename T>
__device__ __forceinline__ Cx<T> mul_c16(Cx<T> w, int m) {
  constexpr double c[8] = {1.0, 0.92387953251128675613, 0.70710678118654752440, 0.38268343236508977173,
                           0.0, -0.38268343236508977173, -0.70710678118654752440, -0.92387953251128675613};
  constexpr double sn[8] = {0.0, 0.38268343236508977173, 0.70710678118654752440, 0.92387953251128675613,
                            1.0, 0.92387953251128675613, 0.70710678118654752440, 0.38268343236508977173};
  if (m == 0) return w;
  if (m == 4) return Cx<T>{w.y, -w.x};
  const T cc = (T)c[m], ss = (T)sn[m];
  return Cx<T>{w.x * cc + w.y * ss, w.y * cc - w.x * ss};
}

constexpr size_t kLdsBudget = 128 * 1024;  // of 160 KB per CU

// Geometry of the spectral buffers.
//   bufA (exchange layout) : [comp][rank P][z nzl][y ny][KXA]         x_fwd out, x_inv in
//   bufT (after all-to-all): [comp][slab P][z nzl][y ny][KXA]         y_fwd in, y_inv out
//                            = [comp][z nz][y ny][KXA]; one component's rank blocks are
//                            contiguous, so every component is exchanged on its own
//   bufB                   : [comp nc][z / ZB][ky' 2ny][z % ZB][KXP]   z_conv in place; blocked in z so
//                            that neither a y tile (all ky' of one z) nor a z tile (all z of one
//                            ky') strides over more than a few dozen pages
//   green                  : [ky folded ny+1][kz folded nz+1][KXP]    real
// KXL = spectral columns per rank (column j holds the shifted frequency 2 j + 1/2, j = 0..nx-1, cut into P ranges),
// KXP = row pitch of bufB and of the Green's table (KXL rounded up to 16 columns: whole 128-byte
// lines, plus one more tile where that would be a power-of-two pitch).  KXA = row pitch of the exchange buffers: KXP on one rank; across ranks KXL rounded up to
// 64 bytes, because every pad column travels through both all-to-alls (19 % of their bytes at 8 ranks
// with whole lines) while rows on arbitrary 8-byte offsets make the y kernels write scattered partial
// lines (see smi_pruned_create).  The y kernels then read rows of bufT past their end (the
// next row's first columns stand in for the pad columns of a tile: finite values that only reach
// pad columns of bufB) and store only columns below KXA; what is left of the partial-line cost is why
// a single rank keeps the padded pitch.
struct PGeo {
  int nz, ny, nx, nzl, P, nc, KXL, KXP, KXA, zb_shift;
  int c0;  // first component of this launch (components c0 .. c0 + nc - 1)
  int colbase;  // column kernels: first spectral column of this launch (its tiles start there)
  int nct, n2, n3;  // column kernels: tiles per row, second and third extent of the launch
  int zg;           // y kernels: consecutive planes one workgroup walks over
  int zl0;          // x kernels: first local plane of this launch (grid y counts planes from there)
  // y kernels: planes of this launch.  yc_n = 0: all nz of them.  Otherwise the planes one z chunk of a transpose
  // delivers or takes: local planes [yc_lo, yc_lo + yc_n) of every rank's slab, i.e. z = s nzl + yc_lo + i for
  // s < P, i < yc_n, P yc_n planes in all, numbered s yc_n + i
  int yc_lo, yc_n;
  // column kernels: the third extent of the launch runs over the components, one per workgroup, instead of every
  // workgroup walking over all of them (small grids: more, shorter workgroups)
  int csplit;
};

// The column kernels run on a one-dimensional grid whose blocks are dealt to the eight XCDs
// round-robin by linear id.  A tile of TS columns narrower than a 128-byte line shares every
// line with its PF - 1 neighbours; those get consecutive slots on the SAME XCD, so that its L2
// fetches and writes back each line once.  Returns false for the blocks that pad the grid.
template <int PF>
__device__ __forceinline__ bool col_tile(const PGeo& G, int& ct, int& j, int& k) {
  unsigned q = blockIdx.x;
  if constexpr (PF > 1) {
    const unsigned xcd = q & 7, slot = q >> 3;
    q = ((slot / PF) * 8 + xcd) * PF + slot % PF;
  }
  ct = (int)(q % (unsigned)G.nct);
  const unsigned r = q / (unsigned)G.nct;
  j = (int)(r % (unsigned)G.n2);
  k = (int)(r / (unsigned)G.n2);
  return k < G.n3;
}

// element offset of bufB row (c, z, ky')
__device__ __forceinline__ size_t rowB(const PGeo& G, int c, int z, int kyp) {
  const int ZB = 1 << G.zb_shift;
  return ((((size_t)c * (G.nz >> G.zb_shift) + (z >> G.zb_shift)) * (2 * G.ny) + kyp) * ZB + (z & (ZB - 1))) *
         (size_t)G.KXP;
}

// DENSE (column kernels of the mixed plans with 32 < L/8 < 64, i.e. 320 and 384 points): L/8 lanes per sequence
// instead of 64, so no wave idles through the radix-8 stages and the workgroup (16 L/8 lanes: 640 or 768) may hold
// 168 registers per lane instead of 128.
template <typename T, int LG, bool ROWS, int WANT_COLS = 0, bool DENSE = false>
struct TileCfg {
  static constexpr int L = Plan<LG>::L, L8 = Plan<LG>::L8;
  static constexpr bool DENSE_ON = DENSE && !ROWS && !Plan<LG>::POW2 && L8 > 32 && L8 < 64;
  static constexpr int TPS = DENSE_ON ? L8 : Plan<LG>::TPS;
  static constexpr int TS = tile_seqs(L, TPS, sizeof(T) == 4, ROWS, WANT_COLS, Plan<LG>::POW2), NT = TS * TPS;
  static constexpr int PITCH = L + L / 8 + 1;                            // row layout
  // waves per SIMD the register allocation must allow: single precision column kernels of 512
  // lanes keep to 128 registers, like those of 1024, so that two of them fit a CU
  static constexpr int WPE = sizeof(T) == 4 && !ROWS && TS * Plan<LG>::TPS >= 512 && Plan<LG>::POW2 ? 4 : 1;
  static constexpr int ELEMS = ROWS ? TS * PITCH : L * TS;               // one tile
  static constexpr bool PP = 2 * ELEMS * sizeof(Cx<T>) <= kLdsBudget;    // room for two tiles
  static constexpr size_t BYTES = ((PP ? 2 : 1) * ELEMS + TwTab<LG>::SIZE) * sizeof(Cx<T>);  // tiles, twiddle table
  using Lay = typename std::conditional<ROWS, RowLayout<PITCH, TPS>, ColLayout<TS, TPS>>::type;
  // column tiles per 128-byte line
  static constexpr int PF = (!ROWS && TS * sizeof(Cx<T>) < 128) ? 128 / (TS * (int)sizeof(Cx<T>)) : 1;
};

#define FFT_KERNEL_PROLOGUE(ROWS)          \
  using DefaultCfg_ = TileCfg<T, LG, ROWS>; \
  FFT_KERNEL_PROLOGUE_T(ROWS, DefaultCfg_)
#define FFT_KERNEL_PROLOGUE_T(ROWS, CFG)                                               \
  using Cfg = CFG;                                                                     \
  using Lay = typename Cfg::Lay;                                                       \
  constexpr int L = Cfg::L, L8 = Cfg::L8, TS = Cfg::TS;                                \
  extern __shared__ char smem[];                                                       \
  TileSel<T, Cfg::PP> sel(reinterpret_cast<Cx<T>*>(smem), Cfg::ELEMS);                 \
  const int t = ROWS ? threadIdx.x / Cfg::TPS : threadIdx.x & (TS - 1);                \
  const int u = ROWS ? threadIdx.x % Cfg::TPS : threadIdx.x / TS;                      \
  const bool item = Plan<LG>::POW2 || u < L8; /* lanes beyond the work items only serve the middle stage */ \
  Twiddles<T, LG> tw;                                                                  \
  tw.tab = reinterpret_cast<Cx<T>*>(smem) + (Cfg::PP ? 2 : 1) * Cfg::ELEMS;            \
  fill_tw_tab<T, LG, 1>(const_cast<Cx<T>*>(tw.tab), tw4);                              \
  load_stage_tw<T, LG, 0>(tw, u, tw4);                                                 \
  const Cx<T> wu = tw4[2 * u]; /* exp(-i pi u / L) */                                  \
  (void)L; (void)wu; (void)item;

// Addressing in the column kernels: every global access is (workgroup-uniform 64-bit
// pointer that absorbs the element index m) [32-bit per-thread offset].  The per-thread part
// is computed once; no per-access 64-bit multiplies are left in the vector pipeline.  The
// create call checks that one (component, tile) slice stays below 2^32 bytes.
//
// Tiles of at most 16 columns always lie inside the padded row pitch, so their pad columns
// are transformed along with the rest (finite garbage that nothing reads) and no lane is
// predicated off.

// Tile shape of the z convolution: that of the y kernels unless SMI_FFT_ZCOLS asks for another
// width (8-column tiles, i.e. two 512-lane workgroups per CU, measured slower at every length).
// 1: the z convolution of the 320- and 384-point plans on dense tiles (see TileCfg) in its full form (Green's values
// of both parities held across tiles, next column requested two transforms ahead: 163 / 152 registers, no scratch);
// 2: dense tiles, lean form; 0: 64 lanes per sequence, lean form.  Measured z convolution at 384^3: 1.75 / 1.83 / 1.92
// ms; at 320^3: 1.13 / 1.19 / 1.21 ms.
#ifndef SMI_FFT_ZDENSE
#define SMI_FFT_ZDENSE 1
#endif
template <typename T, int LG, int WC = SMI_FFT_ZCOLS>
using ZCfg = TileCfg<T, LG, false, WC, SMI_FFT_ZDENSE != 0>;

// ------------------------------------------------------------------------- y forward
// The y kernels walk over G.zg consecutive planes and the G.nc components of one column tile
// (three tiles per workgroup in the launches below) and request the next tile's column while
// the current one is transformed: with a single workgroup per CU nothing else would cover
// that latency.
struct YWalk {
  int z0, zg, nc, ntile;
  int c0;
  __device__ __forceinline__ YWalk(const PGeo& G, int zgroup, int k3) {
    const int planes = G.yc_n ? G.P * G.yc_n : G.nz;
    z0 = zgroup * G.zg;
    zg = min(G.zg, planes - z0);
    nc = G.csplit ? 1 : G.nc;
    c0 = G.csplit ? G.c0 + k3 : G.c0;
    ntile = zg * nc;
  }
  __device__ __forceinline__ void tile(const PGeo& G, int it, int& c, int& z) const {
    const int k = it / zg;
    c = c0 + k;
    z = z0 + (it - k * zg);
    if (G.yc_n) {  // number within the chunk -> plane
      const int sl = z / G.yc_n;
      z = sl * G.nzl + G.yc_lo + (z - sl * G.yc_n);
    }
  }
};

template <typename T, int LG, int WC = 0>
__global__ void __launch_bounds__((TileCfg<T, LG, false, WC>::NT), (TileCfg<T, LG, false, WC>::WPE))
    k_fft_y_fwd(const Cx<T>* __restrict__ in, Cx<T>* __restrict__ out, PGeo G, const Cx<T>* __restrict__ tw4) {
  using YC_ = TileCfg<T, LG, false, WC>;
  FFT_KERNEL_PROLOGUE_T(false, YC_)
  int ct, zgroup, k3;
  if (!col_tile<Cfg::PF>(G, ct, zgroup, k3)) return;
  const YWalk W(G, zgroup, k3);
  const int col = G.colbase + ct * TS + t;
  const bool ok = (TS <= 16 || col < G.KXL) && item;
  const unsigned kystride = (unsigned)G.KXP << G.zb_shift;
  const unsigned off_in = (unsigned)u * G.KXA + col, off_out = (unsigned)u * kystride + col;
  auto src_of = [&](int it) {
    int c, z;
    W.tile(G, it, c, z);
    return in + (((size_t)c * G.nz + z) * G.ny) * (size_t)G.KXA;
  };
  auto dst_of = [&](int it) {
    int c, z;
    W.tile(G, it, c, z);
    return out + rowB(G, c, z, 0);
  };
  Cx<T> v[8], a0[8];
  // the column is read once and kept in registers for both parities
  {
    const Cx<T>* src = src_of(0);
#pragma unroll
    for (int m = 0; m < 8; ++m) a0[m] = ok ? stream_load(&(src + (size_t)(m * L8) * G.KXA)[off_in]) : Cx<T>{0, 0};
  }
  for (int it = 0; it < W.ntile; ++it) {
    Cx<T>* dst = dst_of(it);
    const Cx<T>* nsrc = src_of(it + 1 < W.ntile ? it + 1 : it);
#pragma unroll
    for (int parity = 0; parity < 2; ++parity) {
#pragma unroll
      for (int m = 0; m < 8; ++m) v[m] = parity ? a0[m] * mul_c16(wu, m) : a0[m];
      if (parity == 1 && it + 1 < W.ntile) {
#pragma unroll
        for (int m = 0; m < 8; ++m) a0[m] = ok ? stream_load(&(nsrc + (size_t)(m * L8) * G.KXA)[off_in]) : Cx<T>{0, 0};
      }
      fft_run<-1, false, LG, Lay>(v, sel, t, u, tw);
      if (ok) {
#pragma unroll
        for (int m = 0; m < 8; ++m) stream_store(&(dst + (size_t)(parity * L + m * L8) * kystride)[off_out], v[m]);
      }
    }
  }
}

// ------------------------------------------------------------------------- y inverse
template <typename T, int LG, int WC = 0>
__global__ void __launch_bounds__((TileCfg<T, LG, false, WC>::NT), (TileCfg<T, LG, false, WC>::WPE))
    k_fft_y_inv(const Cx<T>* __restrict__ in, Cx<T>* __restrict__ out, PGeo G, const Cx<T>* __restrict__ tw4) {
  using YC_ = TileCfg<T, LG, false, WC>;
  FFT_KERNEL_PROLOGUE_T(false, YC_)
  int ct, zgroup, k3;
  if (!col_tile<Cfg::PF>(G, ct, zgroup, k3)) return;
  const YWalk W(G, zgroup, k3);
  const int col = G.colbase + ct * TS + t;
  const bool ok = (TS <= 16 || col < G.KXL) && item;
  const unsigned kystride = (unsigned)G.KXP << G.zb_shift;
  const unsigned off_in = (unsigned)u * kystride + col, off_out = (unsigned)u * G.KXA + col;
  const bool st_ok = ok && col < G.KXA;  // a tile's pad columns lie in the next row of a tightly pitched bufT
  auto src_of = [&](int it) {
    int c, z;
    W.tile(G, it, c, z);
    return in + rowB(G, c, z, 0);
  };
  auto dst_of = [&](int it) {
    int c, z;
    W.tile(G, it, c, z);
    return out + (((size_t)c * G.nz + z) * G.ny) * (size_t)G.KXA;
  };
  // x0 / x1: even / odd half of the spectrum of the tile about to be transformed; each is
  // requested again for the next tile as soon as it has been copied out
  Cx<T> v[8], res[8], x0[8], x1[8];
  {
    const Cx<T>* src = src_of(0);
#pragma unroll
    for (int m = 0; m < 8; ++m) x0[m] = ok ? stream_load(&(src + (size_t)(m * L8) * kystride)[off_in]) : Cx<T>{0, 0};
#pragma unroll
    for (int m = 0; m < 8; ++m) x1[m] = ok ? stream_load(&(src + (size_t)(L + m * L8) * kystride)[off_in]) : Cx<T>{0, 0};
  }
  for (int it = 0; it < W.ntile; ++it) {
    Cx<T>* dst = dst_of(it);
    const bool more = it + 1 < W.ntile;
    const Cx<T>* nsrc = src_of(more ? it + 1 : it);
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = x0[m];
    if (more) {
#pragma unroll
      for (int m = 0; m < 8; ++m) x0[m] = ok ? stream_load(&(nsrc + (size_t)(m * L8) * kystride)[off_in]) : Cx<T>{0, 0};
    }
    fft_run<+1, false, LG, Lay>(v, sel, t, u, tw);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      res[m] = v[m];
      v[m] = x1[m];
    }
    if (more) {
#pragma unroll
      for (int m = 0; m < 8; ++m) x1[m] = ok ? stream_load(&(nsrc + (size_t)(L + m * L8) * kystride)[off_in]) : Cx<T>{0, 0};
    }
    fft_run<+1, false, LG, Lay>(v, sel, t, u, tw);
    if (st_ok) {
#pragma unroll
      for (int m = 0; m < 8; ++m)
        stream_store(&(dst + (size_t)(m * L8) * G.KXA)[off_out], res[m] + conj(mul_c16(wu, m)) * v[m]);
    }
  }
}

// ------------------------------------------------------------------------- z convolution
// forward nz -> 2nz, multiply by the folded Green's table, inverse 2nz -> nz, in place.
// Element m of a work item is plane z = u + m L/8.  The z blocks of bufB are at most L/8
// planes (create enforces it), so z % ZB = u % ZB and z / ZB = u / ZB + m (L/8) / ZB.
// Its spectrum index kz = 2 z + pz folds to kz for m < 4 and to 2 nz - kz for m >= 4.
//
// One workgroup owns a column tile and one folded ky index: it walks over the (at most two)
// ky' rows that fold onto it and over the components, up to six tiles that all use the same
// rows of the Green's table (read from HBM once, from cache afterwards).  The next tile's
// column is requested as soon as the registers of the current one are free (after the odd
// parity pre-twiddle), so its latency hides behind the second half of the arithmetic.
template <typename T, int LG, int WC = SMI_FFT_ZCOLS>
__global__ void __launch_bounds__((ZCfg<T, LG, WC>::NT), (ZCfg<T, LG, WC>::WPE))
    k_fft_z_conv(Cx<T>* __restrict__ buf, const T* __restrict__ green, PGeo G, const Cx<T>* __restrict__ tw4) {
  using ZC_ = ZCfg<T, LG, WC>;
  FFT_KERNEL_PROLOGUE_T(false, ZC_)
  // The mixed plans add their middle-stage operands to the three register sets of this kernel and
  // do not fit 128 registers with the Green's values of both parities held across tiles and the
  // next column requested two transforms ahead: they reload the former per tile and request the
  // latter one transform ahead.
  constexpr bool LEAN = !Plan<LG>::POW2 && Plan<LG>::RM > SMI_FFT_LEAN_ABOVE && !(Cfg::DENSE_ON && SMI_FFT_ZDENSE == 1);
  int ct, kyc, k;
  if (!col_tile<Cfg::PF>(G, ct, kyc, k)) return;
  const int col = G.colbase + ct * TS + t;
  const bool ok = (TS <= 16 || col < G.KXL) && item;
  const int ZB = 1 << G.zb_shift;
  const size_t blk = (size_t)(2 * G.ny) * ZB * G.KXP;  // elements per z block of one component
  const size_t cstride = (size_t)(G.nz >> G.zb_shift) * blk;
  const size_t mstep = (size_t)(L8 >> G.zb_shift) * blk;
  const unsigned off = (unsigned)(u >> G.zb_shift) * (unsigned)blk + (unsigned)(u & (ZB - 1)) * G.KXP + col;
  // ky' rows folding onto kyc: ky = kyc and ky = 2 ny - kyc (the same row for kyc = 0 and ny)
  const int nk = (kyc == 0 || kyc == G.ny) ? 1 : 2;
  const int ncl = G.csplit ? 1 : G.nc, c_lo = G.csplit ? G.c0 + k : G.c0;  // components of this workgroup
  const int ntile = nk * ncl;
  auto tile_base = [&](int it) {
    const int ky = it < ncl ? kyc : 2 * G.ny - kyc;
    const int c = c_lo + (it < ncl ? it : it - ncl);
    const int kyp = (ky & 1) * G.ny + (ky >> 1);
    return buf + (size_t)c * cstride + (size_t)kyp * ZB * G.KXP;
  };
  const T* gbase = green + (size_t)kyc * (G.nz + 1) * G.KXP;  // all kz of this ky are contiguous rows
  Cx<T> v[8], res[8], a0[8];
  Cx<T>* base = tile_base(0);
#pragma unroll
  for (int m = 0; m < 8; ++m) a0[m] = ok ? stream_load(&(base + m * mstep)[off]) : Cx<T>{0, 0};
  for (int it = 0; it < ntile; ++it) {
    Cx<T>* next = tile_base(it + 1 < ntile ? it + 1 : it);
#pragma unroll
    for (int pz = 0; pz < 2; ++pz) {
#pragma unroll
      for (int m = 0; m < 8; ++m) v[m] = pz ? a0[m] * mul_c16(wu, m) : a0[m];
      if (!LEAN && pz == 1 && it + 1 < ntile) {
#pragma unroll
        for (int m = 0; m < 8; ++m) a0[m] = ok ? stream_load(&(next + m * mstep)[off]) : Cx<T>{0, 0};
      }
      // Green's table values for the spectrum at kz = 2 z + pz: all eight loads are issued
      // before the transform so that their latency hides behind it
      T gval[8];
      unsigned gvar = 0;
      if (LEAN) asm volatile("" : "+v"(gvar));  // reloaded per tile instead of held in registers
      const unsigned goff_lo = (unsigned)(2 * u + pz) * G.KXP + col + gvar;                 // rows kz
      const unsigned goff_hi = (unsigned)(2 * (L8 - 1 - u) + 1 - pz) * G.KXP + col + gvar;  // rows 2 nz - kz
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        // m < 4: row 2 m L/8 + (2u + pz).  m >= 4: row 2 nz - 2 m L/8 - (2u + pz), split as
        // (2 nz - 2 m L/8 - 2 L/8 + 1) + (2 (L/8 - 1 - u) + 1 - pz) so that both parts are >= 0.
        const T* gm = m < 4 ? gbase + (size_t)(2 * m * L8) * G.KXP
                            : gbase + (size_t)(2 * L - 2 * m * L8 - 2 * L8 + 1) * G.KXP;
        gval[m] = ok ? gm[m < 4 ? goff_lo : goff_hi] : T(0);
      }
      fft_run<-1, false, LG, Lay>(v, sel, t, u, tw);
      // the transform leaves element u + m L/8 in v[m], which is exactly what the first stage
      // of the inverse wants from this work item: scale in registers and go straight on
#pragma unroll
      for (int m = 0; m < 8; ++m) v[m] = scale(v[m], gval[m]);
      if (LEAN && pz == 1 && it + 1 < ntile) {
#pragma unroll
        for (int m = 0; m < 8; ++m) a0[m] = ok ? stream_load(&(next + m * mstep)[off]) : Cx<T>{0, 0};
      }
      fft_run<+1, false, LG, Lay>(v, sel, t, u, tw);
      if (pz == 0) {
#pragma unroll
        for (int m = 0; m < 8; ++m) res[m] = v[m];
      } else {
        if (ok) {
#pragma unroll
          for (int m = 0; m < 8; ++m) stream_store(&(base + m * mstep)[off], res[m] + conj(mul_c16(wu, m)) * v[m]);
        }
      }
    }
    base = next;
  }
}

// ------------------------------------------------------------------------- x forward
// Real rows of nx points, zero padded to 2 nx, at the half-sample-shifted frequencies k + 1/2 of the 2 nx-point
// grid: X[k + 1/2] = sum_n x[n] exp(-2 pi i n (k + 1/2) / 2nx).  A real row has X[(2nx - 1 - k) + 1/2] =
// conj X[k + 1/2], so the nx values at even k = 2 j are a complete set, all of them complex (no real-only columns as
// the unshifted transform has at kx = 0 and at the Nyquist frequency: nx columns instead of nx + 1, whole 16-column
// tiles for every power-of-two grid and rank count), and
//   X[2 j + 1/2] = sum_{n < nx} (x[n] exp(-i pi n / 2nx)) exp(-2 pi i n j / nx):
// ONE nx-point complex transform of the pre-twiddled row.  The shifted transform diagonalises the negacyclic
// convolution; the Green's table is the shifted transform of the negacyclic (odd-wrapped) extension of the even
// kernel, which is real again (see smi_greens_spectrum_shifted), and on the nx outputs that are kept the negacyclic
// convolution of a row padded with nx zeros IS the linear convolution.
// Lanes run along a row: a work item loads and stores its own eight elements directly (each a 256-byte or longer
// contiguous run per row); the tile is only used by the transform stages.
struct FieldGeo {
  int g, px;
  long long plane, vol;
};

// w * exp(-i pi m / 16): the pre-twiddle exp(-i pi n / 2L) of element n = u + m L/8 is the per-thread factor
// exp(-i pi u / 2L) times this constant
template <typename T>
__device__ __forceinline__ Cx<T> mul_c32(Cx<T> w, int m) {
  constexpr double c[8] = {1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708,
                           0.70710678118654752440, 0.55557023301960222474, 0.38268343236508977173,
                           0.19509032201612826785};
  constexpr double sn[8] = {0.0, 0.19509032201612826785, 0.38268343236508977173, 0.55557023301960222474,
                            0.70710678118654752440, 0.83146961230254523708, 0.92387953251128675613,
                            0.98078528040323044913};
  if (m == 0) return w;
  const T cc = (T)c[m], ss = (T)sn[m];
  return Cx<T>{w.x * cc + w.y * ss, w.y * cc - w.x * ss};
}

// element offset of spectral column j of row (c, zl, y) in the exchange layout
__device__ __forceinline__ size_t colA(const PGeo& G, size_t rowbase, size_t rank_stride, int j) {
  if (G.P == 1) return rowbase + j;
  const int rk = j / G.KXL;
  return rowbase + rk * rank_stride + (j - rk * G.KXL);
}

template <typename T, int LG>
__global__ void __launch_bounds__((TileCfg<T, LG, true>::NT))
    k_fft_x_fwd(const T* __restrict__ field, Cx<T>* __restrict__ out, PGeo G, FieldGeo F,
                const Cx<T>* __restrict__ tw4) {
  FFT_KERNEL_PROLOGUE(true)
  const int c = G.c0 + blockIdx.z, zl = G.zl0 + blockIdx.y;
  const size_t rank_stride = (size_t)G.nzl * G.ny * G.KXA;
  const Cx<T> wh = tw4[item ? u : 0];  // exp(-i pi u / 2L)
  Cx<T> pre[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) pre[m] = mul_c32(wh, m);
  for (int it = 0; it < x_tiles<LG>(true); ++it) {
    const int y = (blockIdx.x * x_tiles<LG>(true) + it) * TS + t;
    const bool ok = y < G.ny && item;
    const T* row = field + (size_t)c * F.vol + (size_t)(zl + F.g) * F.plane + (size_t)(y + F.g) * F.px + F.g;
    const size_t rowbase = ((((size_t)c * G.P) * G.nzl + zl) * G.ny + y) * (size_t)G.KXA;
    Cx<T> v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const T x = ok ? row[u + m * L8] : T(0);
      v[m] = Cx<T>{x * pre[m].x, x * pre[m].y};
    }
    fft_run<-1, false, LG, Lay>(v, sel, t, u, tw);
    if (ok) {
#pragma unroll
      for (int m = 0; m < 8; ++m) stream_store(&out[colA(G, rowbase, rank_stride, u + m * L8)], v[m]);
    }
  }
}

// ------------------------------------------------------------------------- x inverse
// x[n] = (1 / nx) Re[ exp(+i pi n / 2nx) sum_j X[2 j + 1/2] exp(+2 pi i n j / nx) ] for n < nx (the odd frequencies
// are the conjugates of the even ones and double the real part; 1 / nx sits in the Green's table): one nx-point
// inverse transform and the real part of the post-twiddled result.
template <typename T, int LG>
__global__ void __launch_bounds__((TileCfg<T, LG, true>::NT))
    k_fft_x_inv(const Cx<T>* __restrict__ in, T* __restrict__ field, PGeo G, FieldGeo F,
                const Cx<T>* __restrict__ tw4) {
  FFT_KERNEL_PROLOGUE(true)
  const int c = G.c0 + blockIdx.z, zl = G.zl0 + blockIdx.y;
  const size_t rank_stride = (size_t)G.nzl * G.ny * G.KXA;
  const Cx<T> wh = tw4[item ? u : 0];
  Cx<T> pre[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) pre[m] = mul_c32(wh, m);
  for (int it = 0; it < x_tiles<LG>(false); ++it) {
    const int y = (blockIdx.x * x_tiles<LG>(false) + it) * TS + t;
    const bool ok = y < G.ny && item;
    T* row = field + (size_t)c * F.vol + (size_t)(zl + F.g) * F.plane + (size_t)(y + F.g) * F.px + F.g;
    const size_t rowbase = ((((size_t)c * G.P) * G.nzl + zl) * G.ny + y) * (size_t)G.KXA;
    Cx<T> v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m)
      v[m] = ok ? stream_load(&in[colA(G, rowbase, rank_stride, u + m * L8)]) : Cx<T>{0, 0};
    fft_run<+1, false, LG, Lay>(v, sel, t, u, tw);
    if (ok) {
#pragma unroll
      for (int m = 0; m < 8; ++m) row[u + m * L8] = v[m].x * pre[m].x + v[m].y * pre[m].y;  // Re(conj(pre) v)
    }
  }
}

// Folded real Green's table from the spectrum of smi_greens_spectrum_shifted (complex (2nz, 2ny, nx), real part).
template <typename T>
__global__ void __launch_bounds__(kThreads) k_fold_green(T* __restrict__ green, const T* __restrict__ ghat, PGeo G,
                                                         int rank, T norm) {
  const long long total = (long long)(G.nz + 1) * (G.ny + 1) * G.KXP;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int jl = (int)(i % G.KXP);
  long long r = i / G.KXP;
  const int kzc = (int)(r % (G.nz + 1)), kyc = (int)(r / (G.nz + 1));
  const int j = rank * G.KXL + jl;
  T val = T(0);
  if (jl < G.KXL && j < G.nx) val = ghat[2 * (((long long)kzc * (2 * G.ny) + kyc) * G.nx + j)] * norm;
  green[i] = val;
}

bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }
int ilog2(int n);
// template key of the plan for an n-point transform (see Plan): log2(n), n itself for 192 / 320 / 384 / 640 / 768,
// 0 if none
int plan_key(int n) {
  if (is_pow2(n) && n >= 8 && n <= 2048) return ilog2(n);
  return (n == 192 || n == 320 || n == 384 || n == 640 || n == 768) ? n : 0;
}

int ilog2(int n) {
  int s = 0;
  while ((1 << s) < n) ++s;
  return s;
}

}  // namespace

int smi_greens_spectrum_shifted(smi_ctx* c, int gnz, int gny, int gnx, double dx, double x_range, void** ghat_out);
typedef int (*alltoall_fn)(smi_ctx*, const void*, void*, size_t);
int smi_alltoall_impl(smi_ctx* c, const void* send, void* recv, size_t bytes_per_peer, hipStream_t stream);
int smi_alltoall_slice_impl(smi_ctx* c, const void* send, void* recv, size_t peer_stride, size_t offset, size_t bytes,
                            void* stage_send, void* stage_recv, hipStream_t stream);

struct PrunedState {
  PGeo G;
  int lgx, lgy, lgz;  // plan keys of the transform lengths nx, ny, nz
  void* tw[3] = {nullptr, nullptr, nullptr};  // x, y, z twiddle tables (4L complex each)
  void* bufA = nullptr;
  void* bufT = nullptr;  // == bufA on a single rank
  void* bufB = nullptr;
  void* green = nullptr;
  size_t green_bytes = 0;
  // multi-rank: the transposes run on their own stream, one component at a time, so that they
  // overlap the transforms of the other components
  hipStream_t comm_stream = nullptr;
  // per component and z chunk: [0..3] forward chunk ready, [4..7] forward chunk arrived, [8..11] backward chunk
  // ready, [12..15] backward chunk arrived
  hipEvent_t ev[3][16] = {};
  // host-staged test transport only: contiguous copies of one z chunk of every rank block
  void* stage[2] = {nullptr, nullptr};
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
  for (auto& row : s->ev)
    for (auto& e : row)
      if (e) hipEventDestroy(e);
  if (s->comm_stream) hipStreamDestroy(s->comm_stream);
  for (void* b : s->stage) hipFree(b);
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

// full-spectrum Green's function (device, c64 / c128) -> the folded real table of the z convolution
static int fold_greens_table(smi_ctx* c, PrunedState* s, const void* ghat) {
  const PGeo& G = s->G;
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
  return 0;
}

// grids the pruned back end has plans for (what smi_poisson_supports reports)
bool smi_pruned_supports(int gnz, int gny, int gnx) {
  return plan_key(gnx) && plan_key(gny) && plan_key(gnz) && gnx >= 8 && gny >= 8 && gnz >= 8 && gnx <= 2048 &&
         gny <= 2048 && gnz <= 2048;
}

int smi_pruned_create(smi_poisson* p) {
  smi_ctx* c = p->ctx;
  SMI_REQUIRE(plan_key(p->gnx) && plan_key(p->gny) && plan_key(p->gnz) && p->gnx >= 8 && p->gny >= 8 && p->gnz >= 8,
              "SMI_POISSON_PRUNED needs transform lengths nx, ny, nz that are powers of two (8 .. 2048) or 192, 320, 384, "
              "640, 768 "
              "(got %d %d %d)",
              p->gnz, p->gny, p->gnx);
  SMI_REQUIRE(p->gnx <= 2048 && p->gny <= 2048 && p->gnz <= 2048, "SMI_POISSON_PRUNED: grid too large");
  PrunedState* s = new PrunedState();
  p->pruned = s;
  PGeo& G = s->G;
  G.nz = p->gnz;
  G.ny = p->gny;
  G.nx = p->gnx;
  G.P = c->nranks;
  G.nzl = c->nz;
  G.nc = 3;
  G.c0 = 0;
  G.colbase = 0;
  G.zl0 = 0;
  G.KXL = (G.nx + G.P - 1) / G.P;
  G.KXP = (G.KXL + 15) / 16 * 16;
  // nx columns of a power-of-two grid make rows of a power-of-two pitch, and the column kernels walk down hundreds of
  // rows at that stride: one more tile of (never touched) columns keeps their accesses spread over the memory
  // channels (measured at 512^3: y_fwd 2.10 -> 1.9 ms)
  if (G.KXP % 64 == 0) G.KXP += 16;
  // Across ranks the exchange pitch is KXL rounded up to 64 bytes, not to whole lines: every pad column travels through
  // both all-to-alls, but rows that start on arbitrary 8-byte offsets make the y kernels write scattered partial
  // lines (measured on one GPU at 512^3, pitch unit 1 / 4 / 8 / 16 columns: y_inv 3.70 / 2.96 / 2.00 / 1.91 ms,
  // y_fwd 2.52 / 2.45 / 2.26 / 1.96 ms, x_fwd 1.24 / 1.16 / 1.07 / 1.00 ms).  At 8 ranks that is 72 columns for 65:
  // 11 % more bytes on the links than the tight pitch, 10 % fewer than whole lines, and all but 0.06 ms of the 0.3 ms
  // per step the tight pitch costs the kernels of a rank.
  // SMI_FFT_TIGHT_PITCH=<columns>: that rounding unit, on a single rank as well (tests, measurements; 1: exactly KXL).
  {
    const char* tp = getenv("SMI_FFT_TIGHT_PITCH");
    const int dflt = (int)(64 / (2 * smi_elem_size(c)));
    const int unit = tp ? atoi(tp) : dflt;
    const int round_to = unit >= 1 && unit <= 16 ? unit : dflt;
    G.KXA = (G.P > 1 || tp) ? (G.KXL + round_to - 1) / round_to * round_to : G.KXP;
  }
  // z blocks of at most 16 planes and at most nz / 8 (one work item's elements share z % ZB)
  G.zb_shift = 0;  // largest power of two <= 16 that divides nz / 8
  while (G.zb_shift < 4 && (G.nz / 8) % (2 << G.zb_shift) == 0) ++G.zb_shift;
  s->lgx = plan_key(G.nx);
  s->lgy = plan_key(G.ny);
  s->lgz = plan_key(G.nz);
  const size_t cs = 2 * smi_elem_size(c);
  const size_t nA = (size_t)G.P * G.nc * G.nzl * G.ny * G.KXA + 16;  // + what the last row's tile reads past it
  const size_t nB = (size_t)G.nc * G.nz * 2 * G.ny * G.KXP;
  // the column kernels index one workgroup's tile with 32-bit element offsets
  SMI_REQUIRE((size_t)G.nz * 2 * G.ny * G.KXP / 8 < (1ull << 32) && (size_t)(2 * G.nz + 2) * G.KXP < (1ull << 32) &&
                  (size_t)G.ny * G.KXP < (1ull << 32),
              "SMI_POISSON_PRUNED: grid too large for 32-bit tile offsets");
  SMI_CHECK_HIP(hipMalloc(&s->bufA, nA * cs));
  SMI_CHECK_HIP(hipMemsetAsync(s->bufA, 0, nA * cs, c->stream));
  if (G.P > 1) {
    SMI_CHECK_HIP(hipMalloc(&s->bufT, nA * cs));
    SMI_CHECK_HIP(hipMemsetAsync(s->bufT, 0, nA * cs, c->stream));
    SMI_CHECK_HIP(hipStreamCreateWithFlags(&s->comm_stream, hipStreamNonBlocking));
    for (auto& row : s->ev)
      for (auto& e : row) SMI_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  } else {
    s->bufT = s->bufA;
  }
  SMI_CHECK_HIP(hipMalloc(&s->bufB, nB * cs));
  SMI_CHECK_HIP(hipMemsetAsync(s->bufB, 0, nB * cs, c->stream));
  SMI_DISPATCH(c, {
    SMI_CHECK(upload_twiddles<T>(c, G.nx, &s->tw[0]));
    SMI_CHECK(upload_twiddles<T>(c, G.ny, &s->tw[1]));
    SMI_CHECK(upload_twiddles<T>(c, G.nz, &s->tw[2]));
  });
  // Green's table: full spectrum once through rocFFT, then fold / permute / normalise
  void* ghat = nullptr;
  SMI_CHECK(smi_greens_spectrum_shifted(c, G.nz, G.ny, G.nx, p->dx, p->x_range, &ghat));
  const int rc = fold_greens_table(c, s, ghat);
  hipFree(ghat);  // also when the fold failed
  return rc;
}

size_t smi_pruned_greens_size(smi_poisson* p) { return p->pruned ? p->pruned->green_bytes : 0; }

int smi_pruned_greens_download(smi_poisson* p, void* host) {
  SMI_REQUIRE(p->pruned, "pruned state missing");
  return smi_memcpy_d2h(p->ctx, host, p->pruned->green, p->pruned->green_bytes);
}

// Column launches of fewer workgroups than this give every component its own workgroups: on a small grid the walk
// over the components inside one workgroup is the longest chain of the launch (64^3: 130 workgroups of six tiles each
// in the z convolution).  Measured per step: 32^3 71 -> 55 us, 64^3 79 -> 62 us, 128^3 267 -> 262 us with the bound at
// 2048; at 8192 the z convolution of 256^3 loses 10 % (its workgroups no longer share a Green's table row).
#ifndef SMI_FFT_SPLIT_BELOW
#define SMI_FFT_SPLIT_BELOW 2048
#endif
constexpr long long kSplitComponentsBelow = SMI_FFT_SPLIT_BELOW;

template <typename T, int LG, int WC>
static int launch_cols_tiles(smi_ctx* c, PrunedState* s, int which, PGeo G, int colbase, int ncols) {
  G.colbase = colbase;
  // one-dimensional grid over (column tile, z or folded ky, component), padded to whole XCD rounds
  auto grid_for = [&](int ts, int pf, int n2, int n3) {
    G.nct = (ncols + ts - 1) / ts;
    G.n2 = n2;
    G.n3 = n3;
    const long long tiles = (long long)G.nct * n2 * n3, round = 8 * pf;
    return dim3((unsigned)((tiles + round - 1) / round * round));
  };
  if (which == 1) {
    // one workgroup per (column tile, folded ky), looping over mirror rows and components
    if constexpr (WC == SMI_FFT_ZCOLS) {
      using Cfg = ZCfg<T, LG, WC>;
      G.csplit = G.nc > 1 && (long long)((ncols + Cfg::TS - 1) / Cfg::TS) * (G.ny + 1) < kSplitComponentsBelow;
      const dim3 grid = grid_for(Cfg::TS, Cfg::PF, G.ny + 1, G.csplit ? G.nc : 1);
      SMI_CHECK(set_lds((const void*)k_fft_z_conv<T, LG, WC>, Cfg::BYTES));
      k_fft_z_conv<T, LG, WC><<<grid, Cfg::NT, Cfg::BYTES, c->stream>>>((Cx<T>*)s->bufB, (const T*)s->green, G,
                                                                       (const Cx<T>*)s->tw[2]);
    } else {
      SMI_REQUIRE(false, "the z convolution has no narrow-tile variant");
    }
  } else {
    using Cfg = TileCfg<T, LG, false, WC>;
    const int planes = G.yc_n ? G.P * G.yc_n : G.nz;
    G.zg = G.nc >= 3 ? 1 : 3;  // three tiles per workgroup
    if (G.nc < 3) {
      // one component of a rank's share, or a chunk of its planes, is a short launch (one workgroup per CU, a few
      // rounds of the chip): take the two to four planes per workgroup that leave the last round fullest
      const long long cols = (ncols + Cfg::TS - 1) / Cfg::TS;
      double best = 0.0;
      for (int zg : {3, 2, 4}) {
        const long long wgs = cols * ((planes + zg - 1) / zg), rounds = (wgs + 255) / 256;
        const double fill = (double)wgs / (double)(rounds * 256);
        if (fill > best + 0.02) {
          best = fill;
          G.zg = zg;
        }
      }
    }
    G.csplit = G.nc > 1 && (long long)((ncols + Cfg::TS - 1) / Cfg::TS) * ((planes + G.zg - 1) / G.zg) <
                               kSplitComponentsBelow;
    const dim3 grid = grid_for(Cfg::TS, Cfg::PF, (planes + G.zg - 1) / G.zg, G.csplit ? G.nc : 1);
    const Cx<T>* tw = (const Cx<T>*)s->tw[1];
    if (which == 0) {
      SMI_CHECK(set_lds((const void*)k_fft_y_fwd<T, LG, WC>, Cfg::BYTES));
      k_fft_y_fwd<T, LG, WC><<<grid, Cfg::NT, Cfg::BYTES, c->stream>>>((const Cx<T>*)s->bufT, (Cx<T>*)s->bufB, G, tw);
    } else {
      SMI_CHECK(set_lds((const void*)k_fft_y_inv<T, LG, WC>, Cfg::BYTES));
      k_fft_y_inv<T, LG, WC><<<grid, Cfg::NT, Cfg::BYTES, c->stream>>>((const Cx<T>*)s->bufB, (Cx<T>*)s->bufT, G, tw);
    }
  }
  SMI_LAUNCH_CHECK();
  return 0;
}

// yc_n > 0 (y kernels only): the planes of one z chunk of a transpose, local planes [yc_lo, yc_lo + yc_n) of every slab
template <typename T, int LG>
static int launch_cols(smi_ctx* c, PrunedState* s, int which, int c0, int nc, int yc_lo = 0, int yc_n = 0) {
  // which: 0 = y_fwd, 1 = z_conv, 2 = y_inv
  PGeo G = s->G;
  G.nc = nc;
  G.c0 = c0;
  G.yc_lo = yc_lo;
  G.yc_n = which == 1 ? 0 : yc_n;
  const double share = G.yc_n ? (double)G.yc_n / (double)G.nzl : 1.0;
  const int slot = which == 0 ? PROF_FFT_Y_FWD : (which == 1 ? PROF_FFT_Z_CONV : PROF_FFT_Y_INV);
  SmiProfScope prof(c, slot, (which == 1 ? 9.0 : 6.0) * nc * smi_S(c) * share);
  if (which == 1) return launch_cols_tiles<T, LG, SMI_FFT_ZCOLS>(c, s, which, G, 0, G.KXL);
  return launch_cols_tiles<T, LG, 0>(c, s, which, G, 0, G.KXL);
}

// planes [zl0, zl0 + nzc) of the local slab (nzc = 0: all of them)
template <typename T, int LG>
static int launch_rows(smi_ctx* c, PrunedState* s, bool fwd, int c0, int nc, void* field, int zl0 = 0, int nzc = 0) {
  PGeo G = s->G;
  G.nc = nc;
  G.c0 = c0;
  G.zl0 = zl0;
  if (nzc == 0) nzc = G.nzl - zl0;
  const double share = (double)nzc / (double)G.nzl;
  using Cfg = TileCfg<T, LG, true>;
  constexpr int TS = Cfg::TS, NT = Cfg::NT;
  const size_t lds_bytes = Cfg::BYTES;
  FieldGeo F{c->g, c->px, (long long)c->plane, (long long)c->vol};
  const int xt = x_tiles<LG>(fwd);
  dim3 grid((G.ny + TS * xt - 1) / (TS * xt), nzc, nc);
  const Cx<T>* tw = (const Cx<T>*)s->tw[0];
  if (fwd) {
    SMI_CHECK(set_lds((const void*)k_fft_x_fwd<T, LG>, lds_bytes));
    SmiProfScope prof(c, PROF_FFT_X_FWD, 3.0 * nc * smi_S(c) * share);
    k_fft_x_fwd<T, LG><<<grid, NT, lds_bytes, c->stream>>>((const T*)field, (Cx<T>*)s->bufA, G, F, tw);
  } else {
    SMI_CHECK(set_lds((const void*)k_fft_x_inv<T, LG>, lds_bytes));
    SmiProfScope prof(c, PROF_FFT_X_INV, 3.0 * nc * smi_S(c) * share);
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
    LG_CASE(11, __VA_ARGS__) LG_CASE(192, __VA_ARGS__) LG_CASE(384, __VA_ARGS__) LG_CASE(768, __VA_ARGS__) \
    LG_CASE(320, __VA_ARGS__) LG_CASE(640, __VA_ARGS__)                                          \
    default: smi_set_error("unsupported transform length 2^%d", lg); return 1;                  \
  }

int smi_pruned_solve(smi_poisson* p, void* solution, const void* rhs, int ncomp) {
  smi_ctx* c = p->ctx;
  PrunedState* s = p->pruned;
  SMI_REQUIRE(s, "pruned state missing");
  SMI_REQUIRE(ncomp == 1 || ncomp == 3, "smi_pruned_solve: ncomp must be 1 or 3");
  const PGeo& G = s->G;
  const size_t cs = 2 * smi_elem_size(c);
  if (G.P == 1) {
    SMI_DISPATCH(c, {
      LG_SWITCH(s->lgx, SMI_CHECK((launch_rows<T, LG>(c, s, true, 0, ncomp, (void*)rhs)));)
      LG_SWITCH(s->lgy, SMI_CHECK((launch_cols<T, LG>(c, s, 0, 0, ncomp)));)
      LG_SWITCH(s->lgz, SMI_CHECK((launch_cols<T, LG>(c, s, 1, 0, ncomp)));)
      LG_SWITCH(s->lgy, SMI_CHECK((launch_cols<T, LG>(c, s, 2, 0, ncomp)));)
      LG_SWITCH(s->lgx, SMI_CHECK((launch_rows<T, LG>(c, s, false, 0, ncomp, solution)));)
    });
    return 0;
  }
  // Slabs.  Per component k:  x_fwd(k) -> transpose(k) -> y_fwd, z_conv, y_inv (k) -> transpose back(k)
  // -> x_inv(k).  With RCCL the transposes go to the communication stream and events carry the
  // dependencies, so that a transpose overlaps the transforms of the neighbouring components;
  // a host-staged transport (callback) is synchronous and everything stays on one stream.
  //
  // Each transpose may be cut into C = 1, 2 or 4 chunks of local planes (context option transpose_chunks): a plane
  // range is a contiguous slice of every rank block of the exchange layout, on the sending and on the receiving side.
  // The x kernels then run chunk by chunk: x_fwd of chunk j + 1 runs under the transfer of chunk j, and x_inv of
  // chunk j under the transfer of chunk j + 1, which shortens what is exposed at the two ends of the pipeline (the
  // first forward and the last backward transpose) to one chunk.
  const size_t peer_bytes = (size_t)G.nzl * G.ny * G.KXA * cs;   // one component, one peer
  const size_t comp_bytes = peer_bytes * G.P;
  const bool overlap = !c->alltoall_cb;
  hipStream_t cstream = overlap ? s->comm_stream : c->stream;
  int C = c->opt_transpose_chunks;
  while (C > 1 && G.nzl < C) C /= 2;
  if (C > 1 && c->alltoall_cb && !s->stage[0]) {
    const size_t stage_bytes = (size_t)((G.nzl + C - 1) / C) * G.ny * G.KXA * cs * G.P;
    for (void*& b : s->stage) SMI_CHECK_HIP(hipMalloc(&b, stage_bytes));
  }
  auto chunk_lo = [&](int j) { return (int)((long long)G.nzl * j / C); };
  // Invariant of the split: work on the communication stream is bracketed by events on BOTH sides.  A transfer
  // starts only after `ready` (recorded on the compute stream behind the kernel that produced its input), and the
  // compute stream waits for `done` before the kernel that consumes its output.  The last thing this function
  // queues on the compute stream waits for the last `done`, so when it returns the communication stream holds
  // nothing that the compute stream has not been ordered after: the caller may treat the solve as ordinary work on
  // c->stream, and the next solve may reuse bufA / bufT / the events without further synchronisation.
  auto transfer = [&](int k, int j, bool back, hipEvent_t ready, hipEvent_t done) -> int {
    char* a = (char*)s->bufA + (size_t)k * comp_bytes;
    char* t = (char*)s->bufT + (size_t)k * comp_bytes;
    const int z0 = chunk_lo(j), z1 = chunk_lo(j + 1);
    const size_t off = (size_t)z0 * G.ny * G.KXA * cs, bytes = (size_t)(z1 - z0) * G.ny * G.KXA * cs;
    auto run = [&]() -> int {
      if (C == 1) return smi_alltoall_impl(c, back ? t : a, back ? a : t, peer_bytes, cstream);
      return smi_alltoall_slice_impl(c, back ? t : a, back ? a : t, peer_bytes, off, bytes, s->stage[0], s->stage[1],
                                     cstream);
    };
    if (overlap) {
      if (ready) {
        SMI_CHECK_HIP(hipEventRecord(ready, c->stream));
        SMI_CHECK_HIP(hipStreamWaitEvent(cstream, ready, 0));
      }
      SMI_CHECK(run());
      if (done) SMI_CHECK_HIP(hipEventRecord(done, cstream));
    } else {
      SmiProfScope prof(c, PROF_TRANSPOSE, (double)bytes * G.P);
      SMI_CHECK(run());
    }
    return 0;
  };
  // The y passes work plane by plane, so they follow the transposes chunk by chunk as well: y_fwd of the planes that
  // chunk j of a forward transpose delivered starts when that chunk has arrived, and chunk j of a backward transpose
  // leaves as soon as y_inv has written its planes; only the z convolution needs a whole component.
  SMI_DISPATCH(c, {
    for (int k = 0; k < ncomp; ++k)
      for (int j = 0; j < C; ++j) {
        const int z0 = chunk_lo(j), nzc = chunk_lo(j + 1) - z0;
        LG_SWITCH(s->lgx, SMI_CHECK((launch_rows<T, LG>(c, s, true, k, 1, (void*)rhs, z0, nzc)));)
        SMI_CHECK(transfer(k, j, false, s->ev[k][j], s->ev[k][4 + j]));
      }
    for (int k = 0; k < ncomp; ++k) {
      for (int j = 0; j < C; ++j) {
        const int z0 = chunk_lo(j), nzc = chunk_lo(j + 1) - z0;
        if (overlap) SMI_CHECK_HIP(hipStreamWaitEvent(c->stream, s->ev[k][4 + j], 0));
        LG_SWITCH(s->lgy, SMI_CHECK((launch_cols<T, LG>(c, s, 0, k, 1, z0, C > 1 ? nzc : 0)));)
      }
      LG_SWITCH(s->lgz, SMI_CHECK((launch_cols<T, LG>(c, s, 1, k, 1)));)
      for (int j = 0; j < C; ++j) {
        const int z0 = chunk_lo(j), nzc = chunk_lo(j + 1) - z0;
        LG_SWITCH(s->lgy, SMI_CHECK((launch_cols<T, LG>(c, s, 2, k, 1, z0, C > 1 ? nzc : 0)));)
        SMI_CHECK(transfer(k, j, true, s->ev[k][8 + j], s->ev[k][12 + j]));
      }
    }
    for (int k = 0; k < ncomp; ++k)
      for (int j = 0; j < C; ++j) {
        const int z0 = chunk_lo(j), nzc = chunk_lo(j + 1) - z0;
        if (overlap) SMI_CHECK_HIP(hipStreamWaitEvent(c->stream, s->ev[k][12 + j], 0));
        LG_SWITCH(s->lgx, SMI_CHECK((launch_rows<T, LG>(c, s, false, k, 1, solution, z0, nzc)));)
      }
  });
  return 0;
}

// ------------------------------------------------------------------ solve without allocation
// smi_poisson_solve_noalloc: the scalar single-rank solve above with its two spectral work buffers (bufA after
// the x pass, bufB after the y pass) taken from this fixed array instead of the buffers smi_poisson_create
// allocated.  The array is one per GPU and is shared by every call on that GPU; it keeps the contents of the
// last call, so each call clears the part it uses first (the transforms leave the pad columns of a row alone and
// count on them being finite).  Two calls that overlap on one GPU race on it.
constexpr size_t kScratchBytes = 64ull << 20;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];
constexpr int kScratchMaxDevices = 64;
// device address of `scratch` on each GPU (the name itself is only a host-side handle for hipGetSymbolAddress)
static void* g_scratch_ptr[kScratchMaxDevices];

// Called by smi_ctx_create with the context's device current.
int smi_pruned_scratch_init(int device) {
  SMI_REQUIRE(device >= 0 && device < kScratchMaxDevices, "device index %d beyond %d", device, kScratchMaxDevices);
  if (g_scratch_ptr[device]) return 0;
  void* ptr = nullptr;
  SMI_CHECK_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
  SMI_REQUIRE(ptr, "hipGetSymbolAddress returned a null address for the solver scratch");
  g_scratch_ptr[device] = ptr;
  return 0;
}

namespace {
__global__ void __launch_bounds__(256) k_scratch_clear(uint4* __restrict__ p, size_t n16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) p[i] = uint4{0u, 0u, 0u, 0u};
}
}  // namespace

size_t smi_pruned_noalloc_capacity(void) { return kScratchBytes; }

int smi_pruned_solve_noalloc(smi_poisson* p, void* solution, const void* rhs) {
  smi_ctx* c = p->ctx;
  PrunedState* s = p->pruned;
  SMI_REQUIRE(s, "smi_poisson_solve_noalloc: the solver was not created with SMI_POISSON_PRUNED");
  const PGeo& G = s->G;
  SMI_REQUIRE(G.P == 1, "smi_poisson_solve_noalloc: single rank only (this context is rank %d of %d)", c->rank, G.P);
  void* base = g_scratch_ptr[c->device];
  SMI_REQUIRE(base, "smi_poisson_solve_noalloc: no scratch address for device %d", c->device);
  // one component of the work buffers, sized as smi_pruned_create sizes three
  const size_t cs = 2 * smi_elem_size(c);
  const size_t bytesA = (((size_t)G.nzl * G.ny * G.KXA + 16) * cs + 255) / 256 * 256;
  const size_t bytesB = ((size_t)G.nz * 2 * G.ny * G.KXP * cs + 255) / 256 * 256;
  SMI_REQUIRE(bytesA + bytesB <= kScratchBytes,
              "smi_poisson_solve_noalloc: a %d x %d x %d solve needs %zu bytes of work space, the fixed scratch "
              "holds %zu (64^3 in double and 128^3 in single precision fit); use smi_poisson_solve",
              G.nz, G.ny, G.nx, bytesA + bytesB, kScratchBytes);
  const size_t n16 = (bytesA + bytesB) / 16;
  const size_t want = (n16 + 255) / 256;
  k_scratch_clear<<<(int)(want < 2048 ? want : 2048), 256, 0, c->stream>>>((uint4*)base, n16);
  SMI_LAUNCH_CHECK();
  PrunedState t = *s;  // same plans, twiddles and Green's table; only the work buffers differ
  t.bufA = base;
  t.bufT = base;
  t.bufB = (char*)base + bytesA;
  SMI_DISPATCH(c, {
    LG_SWITCH(t.lgx, SMI_CHECK((launch_rows<T, LG>(c, &t, true, 0, 1, (void*)rhs)));)
    LG_SWITCH(t.lgy, SMI_CHECK((launch_cols<T, LG>(c, &t, 0, 0, 1)));)
    LG_SWITCH(t.lgz, SMI_CHECK((launch_cols<T, LG>(c, &t, 1, 0, 1)));)
    LG_SWITCH(t.lgy, SMI_CHECK((launch_cols<T, LG>(c, &t, 2, 0, 1)));)
    LG_SWITCH(t.lgx, SMI_CHECK((launch_rows<T, LG>(c, &t, false, 0, 1, solution)));)
  });
  return 0;
}
