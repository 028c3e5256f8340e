// Unbounded Poisson solver of libsophtmi (gfx950).
//
// Back end SMI_POISSON_ROCFFT restates the reference literally on one GPU: zero-pad the
// right-hand side into the doubled domain, forward real FFT (unnormalised), multiply by
// the Green's-function spectrum times dx^3, backward FFT (normalised), copy the first
// octant back (UnboundedPoissonSolverMPI3D.py:133-167).  rocFFT (through hipFFT) does
// the transforms; pad, multiply and extract are fused pointwise kernels here.
//
// Across z-slabs the same back end divides the transform instead (any grid size rocFFT plans, which is what the
// reference offers through mpi4py-fft, fft_mpi_3d.py:27-48): rows are transformed where they live, an all-to-all
// turns planes into spectral columns, and the y and z transforms run on the doubled (2nz, 2ny) planes of a rank's
// columns (rocfft_solve_divided below).
//
// Back end SMI_POISSON_PRUNED (smi_fft.hip) never materialises the padding.
#include <hipfft/hipfft.h>

#include <algorithm>

#include "smi_common.h"
#include "smi_poisson.h"
#include "smi_prof.h"

#define SMI_CHECK_FFT(expr)                                                       \
  do {                                                                            \
    hipfftResult _e = (expr);                                                     \
    if (_e != HIPFFT_SUCCESS) {                                                   \
      smi_set_error("%s:%d: %s -> hipfft error %d", __FILE__, __LINE__, #expr, (int)_e); \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

int smi_alltoall_impl(smi_ctx* c, const void* send, void* recv, size_t bytes_per_peer, hipStream_t stream);

namespace {
constexpr int kBlock = 256;
// capped grid: the kernels below stride over their range, so that ranges beyond 2^32 elements
// (the doubled grid of a 1024^3 solve) stay within the launch limits
inline int blocks_for(long long n) {
  long long b = (n + kBlock - 1) / kBlock;
  return (int)(b < (1 << 20) ? (b ? b : 1) : (1 << 20));
}
#define SMI_GRID_STRIDE(i, total)                                                   \
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (total); \
       i += (long long)gridDim.x * blockDim.x)

// doubled[zz][yy][xx] = rhs interior or 0 (set_fixed_val + copy_to_doubled_domain, fused)
template <typename T>
__global__ void __launch_bounds__(kBlock) k_pad_load(T* __restrict__ dbl, const T* __restrict__ rhs, int nz, int ny,
                                                     int nx, int g, int px, long long plane) {
  const long long n2x = 2LL * nx, n2y = 2LL * ny, total = 2LL * nz * n2y * n2x;
  SMI_GRID_STRIDE(i, total) {
    int xx = (int)(i % n2x);
    long long r = i / n2x;
    int yy = (int)(r % n2y), zz = (int)(r / n2y);
    T v = T(0);
    if (xx < nx && yy < ny && zz < nz) v = rhs[(long long)(zz + g) * plane + (long long)(yy + g) * px + (xx + g)];
    dbl[i] = v;
  }
}

// Green's function on the doubled node grid (UnboundedPoissonSolverMPI3D.py:82-114)
template <typename T>
__global__ void __launch_bounds__(kBlock) k_greens(T* __restrict__ dbl, int n2z, int n2y, int n2x, T dx, T two_xr,
                                                   T two_yr, T two_zr) {
  const long long total = (long long)n2z * n2y * n2x;
  SMI_GRID_STRIDE(i, total) {
    int xx = (int)(i % n2x);
    long long r = i / n2x;
    int yy = (int)(r % n2y), zz = (int)(r / n2y);
    T x = (T)((double)xx * (double)dx), y = (T)((double)yy * (double)dx), z = (T)((double)zz * (double)dx);
    T xr = fmin(x, two_xr - x), yr = fmin(y, two_yr - y), zr = fmin(z, two_zr - z);
    T dist = sqrt(xr * xr + yr * yr + zr * zr);
    T v = (T(1) / dist) / (T)(4 * M_PI);
    if (i == 0) v = T(1) / ((T)(4 * M_PI) * dx);
    dbl[i] = v;
  }
}

template <typename T>
__global__ void __launch_bounds__(kBlock) k_scale(T* __restrict__ a, long long n, T s) {
  SMI_GRID_STRIDE(i, n) a[i] *= s;
}

// complex multiply in place: a *= b   (elementwise_complex_product)
template <typename T>
__global__ void __launch_bounds__(kBlock) k_cmul(T* __restrict__ a, const T* __restrict__ b, long long ncplx) {
  SMI_GRID_STRIDE(i, ncplx) {
    T ar = a[2 * i], ai = a[2 * i + 1], br = b[2 * i], bi = b[2 * i + 1];
    a[2 * i] = ar * br - ai * bi;
    a[2 * i + 1] = ar * bi + ai * br;
  }
}

// solution interior = doubled[first octant] * norm  (backward normalisation + copy_from_doubled_domain)
template <typename T>
__global__ void __launch_bounds__(kBlock) k_extract(T* __restrict__ sol, const T* __restrict__ dbl, int nz, int ny,
                                                    int nx, int g, int px, long long plane, T norm) {
  const long long total = (long long)nz * ny * nx;
  SMI_GRID_STRIDE(i, total) {
    int x = (int)(i % nx);
    long long r = i / nx;
    int y = (int)(r % ny), z = (int)(r / ny);
    T v = dbl[((long long)z * 2 * ny + y) * 2 * nx + x] * norm;
    sol[(long long)(z + g) * plane + (long long)(y + g) * px + (x + g)] = v;
  }
}

// ---- divided transform (z-slabs on several ranks)
// Green's function on planes z0 .. z0 + nzp - 1 of the doubled node grid (k_greens, a few planes at a time)
template <typename T>
__global__ void __launch_bounds__(kBlock) k_greens_planes(T* __restrict__ dbl, int z0, int nzp, int n2y, int n2x, T dx,
                                                          T two_xr, T two_yr, T two_zr) {
  const long long total = (long long)nzp * n2y * n2x;
  SMI_GRID_STRIDE(i, total) {
    int xx = (int)(i % n2x);
    long long r = i / n2x;
    int yy = (int)(r % n2y), zz = z0 + (int)(r / n2y);
    T x = (T)((double)xx * (double)dx), y = (T)((double)yy * (double)dx), z = (T)((double)zz * (double)dx);
    T xr = fmin(x, two_xr - x), yr = fmin(y, two_yr - y), zr = fmin(z, two_zr - z);
    T dist = sqrt(xr * xr + yr * yr + zr * zr);
    T v = (T(1) / dist) / (T)(4 * M_PI);
    if (zz == 0 && i == 0) v = T(1) / ((T)(4 * M_PI) * dx);
    dbl[i] = v;
  }
}
// this rank's interior rows, zero padded to 2nx: (nzl ny) rows
template <typename T>
__global__ void __launch_bounds__(kBlock) k_pad_rows(T* __restrict__ rows, const T* __restrict__ rhs, int nzl, int ny,
                                                     int nx, int g, int px, long long plane) {
  const long long n2x = 2LL * nx, total = (long long)nzl * ny * n2x;
  SMI_GRID_STRIDE(i, total) {
    const int x = (int)(i % n2x);
    const long long r = i / n2x;
    const int y = (int)(r % ny), z = (int)(r / ny);
    rows[i] = x < nx ? rhs[(long long)(z + g) * plane + (long long)(y + g) * px + x + g] : T(0);
  }
}
// row spectra (rows, nx + 1) <-> blocks by peer (P, rows, KL): peer r holds columns r KL .. r KL + KL - 1 (zeros
// past column nx)
template <typename T, bool TO_BLOCKS>
__global__ void __launch_bounds__(kBlock) k_col_blocks(T* __restrict__ spec, T* __restrict__ blocks, long long rows,
                                                       int nx, int KL, int P) {
  const long long total = (long long)P * rows * KL;
  SMI_GRID_STRIDE(i, total) {
    const int k = (int)(i % KL);
    const long long t = i / KL;
    const long long row = t % rows;
    const int col = (int)(t / rows) * KL + k;
    if (TO_BLOCKS) {
      const bool in = col <= nx;
      const long long src = row * (nx + 1) + col;
      blocks[2 * i] = in ? spec[2 * src] : T(0);
      blocks[2 * i + 1] = in ? spec[2 * src + 1] : T(0);
    } else if (col <= nx) {
      const long long dst = row * (nx + 1) + col;
      spec[2 * dst] = blocks[2 * i];
      spec[2 * dst + 1] = blocks[2 * i + 1];
    }
  }
}
// dense columns (nz, ny, KL) <-> the first quadrant of the doubled planes (2nz, 2ny, KL); PAD fills the rest with zeros
template <typename T, bool PAD>
__global__ void __launch_bounds__(kBlock) k_col_quadrant(T* __restrict__ dbl, T* __restrict__ dense, int nz, int ny,
                                                         int KL) {
  const long long total = PAD ? 4LL * nz * ny * KL : (long long)nz * ny * KL;
  SMI_GRID_STRIDE(i, total) {
    const int k = (int)(i % KL);
    const long long r = i / KL;
    if (PAD) {
      const int yy = (int)(r % (2 * ny)), zz = (int)(r / (2 * ny));
      const bool in = yy < ny && zz < nz;
      const long long src = ((long long)zz * ny + yy) * KL + k;
      dbl[2 * i] = in ? dense[2 * src] : T(0);
      dbl[2 * i + 1] = in ? dense[2 * src + 1] : T(0);
    } else {
      const int y = (int)(r % ny), z = (int)(r / ny);
      const long long src = ((long long)z * 2 * ny + y) * KL + k;
      dense[2 * i] = dbl[2 * src];
      dense[2 * i + 1] = dbl[2 * src + 1];
    }
  }
}
// columns c0 .. c0 + KL - 1 of `rows` row spectra (rows, nx + 1) -> (rows, KL) at out, zeros past column nx
template <typename T>
__global__ void __launch_bounds__(kBlock) k_take_cols(T* __restrict__ out, const T* __restrict__ spec, long long rows,
                                                      int nx, int KL, int c0) {
  const long long total = rows * KL;
  SMI_GRID_STRIDE(i, total) {
    const int k = (int)(i % KL), col = c0 + k;
    const long long src = (i / KL) * (nx + 1) + col;
    out[2 * i] = col <= nx ? spec[2 * src] : T(0);
    out[2 * i + 1] = col <= nx ? spec[2 * src + 1] : T(0);
  }
}
// solution interior = first nx values of this rank's rows * norm
template <typename T>
__global__ void __launch_bounds__(kBlock) k_extract_rows(T* __restrict__ sol, const T* __restrict__ rows, int nzl,
                                                         int ny, int nx, int g, int px, long long plane, T norm) {
  const long long total = (long long)nzl * ny * nx;
  SMI_GRID_STRIDE(i, total) {
    const int x = (int)(i % nx);
    const long long r = i / nx;
    const int y = (int)(r % ny), z = (int)(r / ny);
    sol[(long long)(z + g) * plane + (long long)(y + g) * px + x + g] = rows[r * 2 * nx + x] * norm;
  }
}
}  // namespace

struct RocfftState {
  hipfftHandle fwd = 0, inv = 0;
  void* real_buf = nullptr;   // (2nz, 2ny, 2nx) real
  void* cplx_buf = nullptr;   // (2nz, 2ny, nx+1) complex
  void* ghat = nullptr;       // Green's spectrum * dx^3, complex, same shape
  long long nreal = 0, ncplx = 0;
  // several ranks: the divided transform
  int KL = 0;                              // spectral columns per rank: ceil((nx + 1) / P)
  hipfftHandle row_fwd = 0, row_inv = 0;   // nzl ny rows of 2nx points, real <-> complex
  hipfftHandle col = 0;                    // (2nz, 2ny) planes of KL interleaved columns, complex, in place
  void* row_real = nullptr;                // (nzl, ny, 2nx) real
  void* row_spec = nullptr;                // (nzl, ny, nx + 1) complex
  void* xchg[2] = {nullptr, nullptr};      // (P, nzl, ny, KL) complex: blocks by peer / (nz, ny, KL) dense columns
  void* col_spec = nullptr;                // (2nz, 2ny, KL) complex
  void* ghat_cols = nullptr;               // the same shape: this rank's columns of the Green's spectrum * dx^3
};

// Set-up of the divided transform: plans, buffers, and this rank's columns of the Green's spectrum, which come out of
// the same pipeline (rows of a few doubled planes at a time -> this rank's columns -> the (2nz, 2ny) transform), so no
// rank ever holds the doubled grid.
static int rocfft_create_divided(smi_poisson* p, RocfftState* s) {
  smi_ctx* c = p->ctx;
  const int P = c->nranks, nz = p->gnz, ny = p->gny, nx = p->gnx, nzl = c->nz;
  SMI_REQUIRE(c->nranks_y == 1 && c->nranks_x == 1, "SMI_POISSON_ROCFFT across ranks needs z-slabs");
  const size_t es = smi_elem_size(c);
  const bool f32 = c->dtype == SMI_F32;
  s->KL = (nx + 1 + P - 1) / P;
  const int KL = s->KL;
  s->nreal = 8LL * nz * ny * nx;
  s->ncplx = 4LL * nz * ny * KL;
  const size_t rows = (size_t)nzl * ny;
  // the set-up below pushes the doubled planes through the row buffers in groups of `gz` planes of 2ny rows
  const int gz = std::max(1, (int)(rows / (2 * (size_t)ny)));
  const size_t srows = std::max(rows, (size_t)gz * 2 * ny);
  SMI_CHECK_HIP(hipMalloc(&s->row_real, srows * 2 * nx * es));
  SMI_CHECK_HIP(hipMalloc(&s->row_spec, srows * (nx + 1) * 2 * es));
  for (void*& b : s->xchg) SMI_CHECK_HIP(hipMalloc(&b, (size_t)P * rows * KL * 2 * es));
  SMI_CHECK_HIP(hipMalloc(&s->col_spec, (size_t)s->ncplx * 2 * es));
  SMI_CHECK_HIP(hipMalloc(&s->ghat_cols, (size_t)s->ncplx * 2 * es));
  int nrow[1] = {2 * nx}, ncol[2] = {2 * nz, 2 * ny};
  SMI_CHECK_FFT(hipfftPlanMany(&s->col, 2, ncol, ncol, KL, 1, ncol, KL, 1, f32 ? HIPFFT_C2C : HIPFFT_Z2Z, KL));
  SMI_CHECK_FFT(hipfftSetStream(s->col, c->stream));
  // Green's spectrum: rows of gz doubled planes at a time
  hipfftHandle grow = 0;
  SMI_CHECK_FFT(hipfftPlanMany(&grow, 1, nrow, nullptr, 1, 0, nullptr, 1, 0, f32 ? HIPFFT_R2C : HIPFFT_D2Z, gz * 2 * ny));
  SMI_CHECK_FFT(hipfftSetStream(grow, c->stream));
  const double y_range = p->x_range * ((double)ny / nx), z_range = p->x_range * ((double)nz / nx);
  int rc = 0;
  SMI_DISPATCH(c, {
    const T dx = (T)p->dx;
    for (int z0 = 0; z0 < 2 * nz && rc == 0; z0 += gz) {
      const int nzp = std::min(gz, 2 * nz - z0);
      const long long prow = (long long)nzp * 2 * ny;
      k_greens_planes<T><<<blocks_for(prow * 2 * nx), kBlock, 0, c->stream>>>(
          (T*)s->row_real, z0, nzp, 2 * ny, 2 * nx, dx, (T)(2 * p->x_range), (T)(2 * y_range), (T)(2 * z_range));
      // (a last, shorter group transforms stale rows behind its own as well: they are not copied out)
      hipfftResult fr = f32 ? hipfftExecR2C(grow, (hipfftReal*)s->row_real, (hipfftComplex*)s->row_spec)
                            : hipfftExecD2Z(grow, (hipfftDoubleReal*)s->row_real, (hipfftDoubleComplex*)s->row_spec);
      if (fr != HIPFFT_SUCCESS) {
        smi_set_error("smi_poisson_create: row transform of the Green's function failed (hipfft error %d)", (int)fr);
        rc = 1;
        break;
      }
      k_take_cols<T><<<blocks_for(prow * KL), kBlock, 0, c->stream>>>(
          (T*)s->ghat_cols + 2 * (size_t)z0 * 2 * ny * KL, (const T*)s->row_spec, prow, nx, KL, c->rank * KL);
    }
  });
  hipfftDestroy(grow);
  if (rc) return rc;
  SMI_LAUNCH_CHECK();
  if (f32)
    SMI_CHECK_FFT(hipfftExecC2C(s->col, (hipfftComplex*)s->ghat_cols, (hipfftComplex*)s->ghat_cols, HIPFFT_FORWARD));
  else
    SMI_CHECK_FFT(hipfftExecZ2Z(s->col, (hipfftDoubleComplex*)s->ghat_cols, (hipfftDoubleComplex*)s->ghat_cols,
                                HIPFFT_FORWARD));
  SMI_DISPATCH(c, {
    const T dx = (T)p->dx;
    k_scale<T><<<blocks_for(2 * s->ncplx), kBlock, 0, c->stream>>>((T*)s->ghat_cols, 2 * s->ncplx, dx * dx * dx);
  });
  SMI_LAUNCH_CHECK();
  SMI_CHECK_FFT(hipfftPlanMany(&s->row_fwd, 1, nrow, nullptr, 1, 0, nullptr, 1, 0, f32 ? HIPFFT_R2C : HIPFFT_D2Z,
                               (int)rows));
  SMI_CHECK_FFT(hipfftPlanMany(&s->row_inv, 1, nrow, nullptr, 1, 0, nullptr, 1, 0, f32 ? HIPFFT_C2R : HIPFFT_Z2D,
                               (int)rows));
  SMI_CHECK_FFT(hipfftSetStream(s->row_fwd, c->stream));
  SMI_CHECK_FFT(hipfftSetStream(s->row_inv, c->stream));
  SMI_CHECK_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

static int rocfft_create(smi_poisson* p) {
  smi_ctx* c = p->ctx;
  RocfftState* s = new RocfftState();
  p->rocfft = s;
  if (c->nranks > 1) return rocfft_create_divided(p, s);
  const int n2z = 2 * p->gnz, n2y = 2 * p->gny, n2x = 2 * p->gnx;
  s->nreal = (long long)n2z * n2y * n2x;
  s->ncplx = (long long)n2z * n2y * (p->gnx + 1);
  size_t es = smi_elem_size(c);
  SMI_CHECK_HIP(hipMalloc(&s->real_buf, s->nreal * es));
  SMI_CHECK_HIP(hipMalloc(&s->cplx_buf, s->ncplx * 2 * es));
  SMI_CHECK_HIP(hipMalloc(&s->ghat, s->ncplx * 2 * es));
  const bool f32 = c->dtype == SMI_F32;
  SMI_CHECK_FFT(hipfftPlan3d(&s->fwd, n2z, n2y, n2x, f32 ? HIPFFT_R2C : HIPFFT_D2Z));
  SMI_CHECK_FFT(hipfftPlan3d(&s->inv, n2z, n2y, n2x, f32 ? HIPFFT_C2R : HIPFFT_Z2D));
  SMI_CHECK_FFT(hipfftSetStream(s->fwd, c->stream));
  SMI_CHECK_FFT(hipfftSetStream(s->inv, c->stream));
  // Green's function spectrum, once
  const double y_range = p->x_range * ((double)p->gny / p->gnx), z_range = p->x_range * ((double)p->gnz / p->gnx);
  SMI_DISPATCH(c, {
    T dx = (T)p->dx;
    k_greens<T><<<blocks_for(s->nreal), kBlock, 0, c->stream>>>((T*)s->real_buf, n2z, n2y, n2x, dx,
                                                              (T)(2 * p->x_range), (T)(2 * y_range),
                                                              (T)(2 * z_range));
    SMI_LAUNCH_CHECK();
    if (f32)
      SMI_CHECK_FFT(hipfftExecR2C(s->fwd, (hipfftReal*)s->real_buf, (hipfftComplex*)s->ghat));
    else
      SMI_CHECK_FFT(hipfftExecD2Z(s->fwd, (hipfftDoubleReal*)s->real_buf, (hipfftDoubleComplex*)s->ghat));
    T dx3 = dx * dx * dx;
    k_scale<T><<<blocks_for(2 * s->ncplx), kBlock, 0, c->stream>>>((T*)s->ghat, 2 * s->ncplx, dx3);
    SMI_LAUNCH_CHECK();
  });
  SMI_CHECK_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

// Green's function on the doubled node grid in y and z and on the first nx nodes in x, as the complex sequence
// c_l g(z, y, l) exp(-i pi l / (2 nx)) with c_0 = 1, c_l = 2: the nx-point transform of that sequence along x is the
// half-sample-shifted (odd-frequency) transform of the negacyclic extension of g at the frequencies 2 j + 1/2.
template <typename T>
__global__ void __launch_bounds__(kBlock) k_greens_shifted(T* __restrict__ cplx, int n2z, int n2y, int nx, T dx,
                                                           T two_yr, T two_zr) {
  const long long total = (long long)n2z * n2y * nx;
  SMI_GRID_STRIDE(i, total) {
    int xx = (int)(i % nx);
    long long r = i / nx;
    int yy = (int)(r % n2y), zz = (int)(r / n2y);
    T x = (T)((double)xx * (double)dx), y = (T)((double)yy * (double)dx), z = (T)((double)zz * (double)dx);
    T yr = fmin(y, two_yr - y), zr = fmin(z, two_zr - z);
    T dist = sqrt(x * x + yr * yr + zr * zr);
    T v = (T(1) / dist) / (T)(4 * M_PI);
    if (i == 0) v = T(1) / ((T)(4 * M_PI) * dx);
    if (xx > 0) v *= T(2);
    const double a = -M_PI * (double)xx / (2.0 * (double)nx);
    cplx[2 * i] = v * (T)cos(a);
    cplx[2 * i + 1] = v * (T)sin(a);
  }
}

// Spectrum of the Green's function times dx^3 for the pruned back end: regular frequencies of the doubled grid in z
// and y, half-sample-shifted frequencies 2 j + 1/2 (j = 0 .. nx - 1) of the 2 nx-point grid in x; complex
// (2nz, 2ny, nx) in a fresh device buffer the caller frees.  Its real part is the multiplier (the imaginary part
// belongs to the odd extension that a real, even kernel does not have).
int smi_greens_spectrum_shifted(smi_ctx* c, int gnz, int gny, int gnx, double dx_d, double x_range, void** ghat_out) {
  const int n2z = 2 * gnz, n2y = 2 * gny;
  const long long ncplx = (long long)n2z * n2y * gnx;
  const size_t es = smi_elem_size(c);
  const bool f32 = c->dtype == SMI_F32;
  void* ghat = nullptr;
  hipfftHandle plan = 0;
  SMI_CHECK_HIP(hipMalloc(&ghat, ncplx * 2 * es));
  SMI_CHECK_FFT(hipfftPlan3d(&plan, n2z, n2y, gnx, f32 ? HIPFFT_C2C : HIPFFT_Z2Z));
  SMI_CHECK_FFT(hipfftSetStream(plan, c->stream));
  const double y_range = x_range * ((double)gny / gnx), z_range = x_range * ((double)gnz / gnx);
  SMI_DISPATCH(c, {
    T dx = (T)dx_d;
    k_greens_shifted<T><<<blocks_for(ncplx), kBlock, 0, c->stream>>>((T*)ghat, n2z, n2y, gnx, dx, (T)(2 * y_range),
                                                                    (T)(2 * z_range));
    SMI_LAUNCH_CHECK();
    if (f32)
      SMI_CHECK_FFT(hipfftExecC2C(plan, (hipfftComplex*)ghat, (hipfftComplex*)ghat, HIPFFT_FORWARD));
    else
      SMI_CHECK_FFT(hipfftExecZ2Z(plan, (hipfftDoubleComplex*)ghat, (hipfftDoubleComplex*)ghat, HIPFFT_FORWARD));
    T dx3 = dx * dx * dx;
    k_scale<T><<<blocks_for(2 * ncplx), kBlock, 0, c->stream>>>((T*)ghat, 2 * ncplx, dx3);
    SMI_LAUNCH_CHECK();
  });
  SMI_CHECK_HIP(hipStreamSynchronize(c->stream));
  hipfftDestroy(plan);
  *ghat_out = ghat;
  return 0;
}

static int rocfft_destroy(smi_poisson* p) {
  RocfftState* s = p->rocfft;
  if (!s) return 0;
  if (s->fwd) hipfftDestroy(s->fwd);
  if (s->inv) hipfftDestroy(s->inv);
  hipFree(s->real_buf);
  hipFree(s->cplx_buf);
  hipFree(s->ghat);
  if (s->row_fwd) hipfftDestroy(s->row_fwd);
  if (s->row_inv) hipfftDestroy(s->row_inv);
  if (s->col) hipfftDestroy(s->col);
  hipFree(s->row_real);
  hipFree(s->row_spec);
  hipFree(s->xchg[0]);
  hipFree(s->xchg[1]);
  hipFree(s->col_spec);
  hipFree(s->ghat_cols);
  delete s;
  p->rocfft = nullptr;
  return 0;
}

// z-slabs on several ranks.  Every rank transforms its own rows (zero padded in x only), an all-to-all turns its
// planes into spectral columns (KL of the nx + 1 per rank, all nz planes of ny rows), the columns are padded into the
// doubled (2nz, 2ny) planes, transformed, multiplied by this rank's columns of the Green's spectrum and transformed
// back, and the mirror image brings the first octant home: the reference's MPIDomainDoublingCommunicator3D + PFFT pair
// (UnboundedPoissonSolverMPI3D.py:133-167, 190-382) with two all-to-alls of nzl ny (nx + 1) complex values.
template <typename T>
static int rocfft_solve_divided(smi_poisson* p, void* sol, const void* rhs) {
  smi_ctx* c = p->ctx;
  RocfftState* s = p->rocfft;
  constexpr bool f32 = sizeof(T) == 4;
  const int P = c->nranks, nz = p->gnz, ny = p->gny, nx = p->gnx, nzl = c->nz, KL = s->KL;
  const long long rows = (long long)nzl * ny, nblk = (long long)P * rows * KL;
  const size_t per_peer = (size_t)rows * KL * 2 * sizeof(T);
  k_pad_rows<T><<<blocks_for(rows * 2 * nx), kBlock, 0, c->stream>>>((T*)s->row_real, (const T*)rhs, nzl, ny, nx, c->g,
                                                                    c->px, (long long)c->plane);
  SMI_LAUNCH_CHECK();
  if (f32)
    SMI_CHECK_FFT(hipfftExecR2C(s->row_fwd, (hipfftReal*)s->row_real, (hipfftComplex*)s->row_spec));
  else
    SMI_CHECK_FFT(hipfftExecD2Z(s->row_fwd, (hipfftDoubleReal*)s->row_real, (hipfftDoubleComplex*)s->row_spec));
  k_col_blocks<T, true><<<blocks_for(nblk), kBlock, 0, c->stream>>>((T*)s->row_spec, (T*)s->xchg[0], rows, nx, KL, P);
  SMI_LAUNCH_CHECK();
  // block r of the result = planes r nzl .. of the grid: (nz, ny, KL) in order
  SMI_CHECK(smi_alltoall_impl(c, s->xchg[0], s->xchg[1], per_peer, c->stream));
  k_col_quadrant<T, true><<<blocks_for(s->ncplx), kBlock, 0, c->stream>>>((T*)s->col_spec, (T*)s->xchg[1], nz, ny, KL);
  SMI_LAUNCH_CHECK();
  if (f32)
    SMI_CHECK_FFT(hipfftExecC2C(s->col, (hipfftComplex*)s->col_spec, (hipfftComplex*)s->col_spec, HIPFFT_FORWARD));
  else
    SMI_CHECK_FFT(hipfftExecZ2Z(s->col, (hipfftDoubleComplex*)s->col_spec, (hipfftDoubleComplex*)s->col_spec,
                                HIPFFT_FORWARD));
  k_cmul<T><<<blocks_for(s->ncplx), kBlock, 0, c->stream>>>((T*)s->col_spec, (const T*)s->ghat_cols, s->ncplx);
  SMI_LAUNCH_CHECK();
  if (f32)
    SMI_CHECK_FFT(hipfftExecC2C(s->col, (hipfftComplex*)s->col_spec, (hipfftComplex*)s->col_spec, HIPFFT_BACKWARD));
  else
    SMI_CHECK_FFT(hipfftExecZ2Z(s->col, (hipfftDoubleComplex*)s->col_spec, (hipfftDoubleComplex*)s->col_spec,
                                HIPFFT_BACKWARD));
  k_col_quadrant<T, false><<<blocks_for((long long)nz * ny * KL), kBlock, 0, c->stream>>>((T*)s->col_spec,
                                                                                         (T*)s->xchg[1], nz, ny, KL);
  SMI_LAUNCH_CHECK();
  SMI_CHECK(smi_alltoall_impl(c, s->xchg[1], s->xchg[0], per_peer, c->stream));
  k_col_blocks<T, false><<<blocks_for(nblk), kBlock, 0, c->stream>>>((T*)s->row_spec, (T*)s->xchg[0], rows, nx, KL, P);
  SMI_LAUNCH_CHECK();
  if (f32)
    SMI_CHECK_FFT(hipfftExecC2R(s->row_inv, (hipfftComplex*)s->row_spec, (hipfftReal*)s->row_real));
  else
    SMI_CHECK_FFT(hipfftExecZ2D(s->row_inv, (hipfftDoubleComplex*)s->row_spec, (hipfftDoubleReal*)s->row_real));
  k_extract_rows<T><<<blocks_for(rows * nx), kBlock, 0, c->stream>>>((T*)sol, (const T*)s->row_real, nzl, ny, nx, c->g,
                                                                    c->px, (long long)c->plane,
                                                                    (T)(1.0 / (double)s->nreal));
  SMI_LAUNCH_CHECK();
  return 0;
}

static int rocfft_solve(smi_poisson* p, void* sol, const void* rhs) {
  smi_ctx* c = p->ctx;
  RocfftState* s = p->rocfft;
  const bool f32 = c->dtype == SMI_F32;
  if (c->nranks > 1) {
    SmiProfScope prof(c, PROF_POISSON_FFT_FWD, 0.0);
    SMI_DISPATCH(c, { SMI_CHECK(rocfft_solve_divided<T>(p, sol, rhs)); });
    return 0;
  }
  SMI_DISPATCH(c, {
    {
      SmiProfScope prof(c, PROF_POISSON_PAD, 9.0 * smi_S(c));
      k_pad_load<T><<<blocks_for(s->nreal), kBlock, 0, c->stream>>>((T*)s->real_buf, (const T*)rhs, c->nz, c->ny,
                                                                  c->nx, c->g, c->px, (long long)c->plane);
      SMI_LAUNCH_CHECK();
    }
    {
      SmiProfScope prof(c, PROF_POISSON_FFT_FWD, 48.0 * smi_S(c));
      if (f32)
        SMI_CHECK_FFT(hipfftExecR2C(s->fwd, (hipfftReal*)s->real_buf, (hipfftComplex*)s->cplx_buf));
      else
        SMI_CHECK_FFT(hipfftExecD2Z(s->fwd, (hipfftDoubleReal*)s->real_buf, (hipfftDoubleComplex*)s->cplx_buf));
    }
    {
      SmiProfScope prof(c, PROF_POISSON_MUL, 24.0 * smi_S(c));
      k_cmul<T><<<blocks_for(s->ncplx), kBlock, 0, c->stream>>>((T*)s->cplx_buf, (const T*)s->ghat, s->ncplx);
      SMI_LAUNCH_CHECK();
    }
    {
      SmiProfScope prof(c, PROF_POISSON_FFT_INV, 48.0 * smi_S(c));
      if (f32)
        SMI_CHECK_FFT(hipfftExecC2R(s->inv, (hipfftComplex*)s->cplx_buf, (hipfftReal*)s->real_buf));
      else
        SMI_CHECK_FFT(hipfftExecZ2D(s->inv, (hipfftDoubleComplex*)s->cplx_buf, (hipfftDoubleReal*)s->real_buf));
    }
    {
      SmiProfScope prof(c, PROF_POISSON_EXTRACT, 2.0 * smi_S(c));
      T norm = (T)(1.0 / (double)s->nreal);
      k_extract<T><<<blocks_for((long long)c->nz * c->ny * c->nx), kBlock, 0, c->stream>>>(
          (T*)sol, (const T*)s->real_buf, c->nz, c->ny, c->nx, c->g, c->px, (long long)c->plane, norm);
      SMI_LAUNCH_CHECK();
    }
  });
  return 0;
}

// ------------------------------------------------------------------ C ABI
extern "C" int smi_poisson_create(smi_poisson** out, smi_ctx* ctx, int global_nz, int global_ny, int global_nx,
                                  double dx, double x_range, int backend) {
  SMI_REQUIRE(out && ctx, "smi_poisson_create: null argument");
  SMI_REQUIRE(backend == SMI_POISSON_ROCFFT || backend == SMI_POISSON_PRUNED, "smi_poisson_create: bad backend %d",
              backend);
  SMI_REQUIRE(global_ny == ctx->ny && global_nx == ctx->nx && global_nz == ctx->nz * ctx->nranks,
              "smi_poisson_create: global grid (%d,%d,%d) inconsistent with the z-slab context", global_nz,
              global_ny, global_nx);
  SMI_CHECK_HIP(hipSetDevice(ctx->device));
  smi_poisson* p = new smi_poisson();
  p->ctx = ctx;
  p->backend = backend;
  p->gnz = global_nz;
  p->gny = global_ny;
  p->gnx = global_nx;
  p->dx = dx;
  p->x_range = x_range;
  p->rocfft = nullptr;
  p->pruned = nullptr;
  p->pencil = nullptr;
  p->route_rhs = p->route_sol = p->route_work[0] = p->route_work[1] = nullptr;
  int rc = backend == SMI_POISSON_ROCFFT ? rocfft_create(p) : smi_pruned_create(p);
  if (rc != 0) {
    if (backend == SMI_POISSON_ROCFFT) rocfft_destroy(p); else smi_pruned_destroy(p);
    delete p;
    return rc;
  }
  *out = p;
  return 0;
}

extern "C" int smi_poisson_supports(int global_nz, int global_ny, int global_nx, int nranks, int backend) {
  if (global_nz < 1 || global_ny < 1 || global_nx < 1 || nranks < 1 || global_nz % nranks != 0) return 0;
  if (backend == SMI_POISSON_ROCFFT) return 1;  // one rank: the doubled grid whole; z-slabs: the divided transform
  if (backend == SMI_POISSON_PRUNED) return smi_pruned_supports(global_nz, global_ny, global_nx) ? 1 : 0;
  return 0;
}

extern "C" int smi_poisson_destroy(smi_poisson* p) {
  if (!p) return 0;
  hipSetDevice(p->ctx->device);
  hipStreamSynchronize(p->ctx->stream);
  if (p->backend == SMI_POISSON_ROCFFT) rocfft_destroy(p); else smi_pruned_destroy(p);
  delete p;
  return 0;
}

extern "C" int smi_poisson_solve(smi_poisson* p, void* solution, const void* rhs) {
  SMI_REQUIRE(p && solution && rhs, "smi_poisson_solve: null argument");
  return p->backend == SMI_POISSON_ROCFFT ? rocfft_solve(p, solution, rhs) : smi_pruned_solve(p, solution, rhs, 1);
}

extern "C" int smi_poisson_solve_noalloc(smi_poisson* p, void* solution, const void* rhs) {
  SMI_REQUIRE(p && solution && rhs, "smi_poisson_solve_noalloc: null argument");
  SMI_REQUIRE(p->backend == SMI_POISSON_PRUNED,
              "smi_poisson_solve_noalloc: only solvers created with SMI_POISSON_PRUNED (hipFFT may allocate)");
  return smi_pruned_solve_noalloc(p, solution, rhs);
}

extern "C" size_t smi_poisson_noalloc_capacity(void) { return smi_pruned_noalloc_capacity(); }

extern "C" int smi_poisson_vector_solve(smi_poisson* p, void* solution, const void* rhs) {
  SMI_REQUIRE(p && solution && rhs, "smi_poisson_vector_solve: null argument");
  if (p->backend == SMI_POISSON_PRUNED) return smi_pruned_solve(p, solution, rhs, 3);
  size_t stride = p->ctx->vol * smi_elem_size(p->ctx);
  for (int k = 0; k < 3; ++k)
    SMI_CHECK(rocfft_solve(p, (char*)solution + k * stride, (const char*)rhs + k * stride));
  return 0;
}

// A pencil decomposition solves through the z-slab decomposition of the same grid (the reference's solver moves its
// fields between the grid's and the transform's decomposition likewise, UnboundedPoissonSolverMPI3D.py:190-382).
// The route names the pencil context, two three-component slab arrays (right-hand side, solution) and two work arrays
// of 3 * (pencil interior cells) elements; all stay the caller's.  Null pencil withdraws it.
extern "C" int smi_poisson_set_pencil_route(smi_poisson* p, smi_ctx* pencil, void* slab_rhs, void* slab_sol,
                                            void* work_a, void* work_b) {
  SMI_REQUIRE(p, "smi_poisson_set_pencil_route: null solver");
  if (!pencil) {
    p->pencil = nullptr;
    return 0;
  }
  SMI_REQUIRE(slab_rhs && slab_sol && work_a && work_b && slab_rhs != slab_sol && work_a != work_b,
              "smi_poisson_set_pencil_route: four distinct arrays are needed");
  SMI_REQUIRE(pencil != p->ctx && pencil->device == p->ctx->device && pencil->rank == p->ctx->rank &&
                  pencil->nranks == p->ctx->nranks && pencil->dtype == p->ctx->dtype,
              "smi_poisson_set_pencil_route: the pencil context does not describe the solver's rank");
  p->pencil = pencil;
  p->route_rhs = slab_rhs;
  p->route_sol = slab_sol;
  p->route_work[0] = work_a;
  p->route_work[1] = work_b;
  return 0;
}

// -del^2 solution = rhs for ncomp (1 or 3) components of fields of the routed pencil context: right-hand side to
// slabs, solve, solution back to pencils.  The two contexts have their own streams; events order them and the host
// does not wait.
extern "C" int smi_poisson_pencil_solve(smi_poisson* p, void* solution, const void* rhs, int ncomp) {
  SMI_REQUIRE(p && solution && rhs && (ncomp == 1 || ncomp == 3), "smi_poisson_pencil_solve: bad argument");
  SMI_REQUIRE(p->pencil, "smi_poisson_pencil_solve: no pencil route set (smi_poisson_set_pencil_route)");
  smi_ctx *pen = p->pencil, *slab = p->ctx;
  SMI_CHECK(smi_pencil_to_slab(pen, slab, p->route_rhs, rhs, ncomp, p->route_work[0], p->route_work[1]));
  SMI_CHECK(smi_ctx_wait_for(slab, pen));
  if (ncomp == 3)
    SMI_CHECK(smi_poisson_vector_solve(p, p->route_sol, p->route_rhs));
  else
    SMI_CHECK(smi_poisson_solve(p, p->route_sol, p->route_rhs));
  SMI_CHECK(smi_ctx_wait_for(pen, slab));
  SMI_CHECK(smi_slab_to_pencil(pen, slab, solution, p->route_sol, ncomp, p->route_work[0], p->route_work[1]));
  // the next redistribution reuses the work and slab arrays on the pencil stream: the slab stream must not run
  // ahead of it with the next solve's input
  return smi_ctx_wait_for(slab, pen);
}

int smi_poisson_vector_solve_for(smi_ctx* c, smi_poisson* p, void* solution, const void* rhs) {
  SMI_REQUIRE(c && p, "Poisson solve: null argument");
  if (c == p->ctx) return smi_poisson_vector_solve(p, solution, rhs);
  SMI_REQUIRE(c == p->pencil, "Poisson solve: the solver serves neither this context nor a pencil context routed to it");
  return smi_poisson_pencil_solve(p, solution, rhs, 3);
}

extern "C" size_t smi_poisson_greens_size(smi_poisson* p) {
  if (!p) return 0;
  // (several ranks: this rank's columns, (2nz, 2ny, KL) complex)
  if (p->backend == SMI_POISSON_ROCFFT) return (size_t)p->rocfft->ncplx * 2 * smi_elem_size(p->ctx);
  return smi_pruned_greens_size(p);
}

extern "C" int smi_poisson_greens_download(smi_poisson* p, void* host) {
  SMI_REQUIRE(p && host, "smi_poisson_greens_download: null argument");
  if (p->backend == SMI_POISSON_ROCFFT)
    return smi_memcpy_d2h(p->ctx, host, p->ctx->nranks > 1 ? p->rocfft->ghat_cols : p->rocfft->ghat,
                          smi_poisson_greens_size(p));
  return smi_pruned_greens_download(p, host);
}
