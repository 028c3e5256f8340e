// Unbounded Poisson solver of libsophtmi (gfx950).
//
// Back end SMI_POISSON_ROCFFT restates the reference literally on one GPU: zero-pad the
// right-hand side into the doubled domain, forward real FFT (unnormalised), multiply by
// the Green's-function spectrum times dx^3, backward FFT (normalised), copy the first
// octant back (UnboundedPoissonSolverMPI3D.py:133-167).  rocFFT (through hipFFT) does
// the transforms; pad, multiply and extract are fused pointwise kernels here.
//
// Back end SMI_POISSON_PRUNED (smi_fft.hip) never materialises the padding.
#include <hipfft/hipfft.h>

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
}  // namespace

struct RocfftState {
  hipfftHandle fwd = 0, inv = 0;
  void* real_buf = nullptr;   // (2nz, 2ny, 2nx) real
  void* cplx_buf = nullptr;   // (2nz, 2ny, nx+1) complex
  void* ghat = nullptr;       // Green's spectrum * dx^3, complex, same shape
  long long nreal = 0, ncplx = 0;
};

static int rocfft_create(smi_poisson* p) {
  smi_ctx* c = p->ctx;
  SMI_REQUIRE(c->nranks == 1, "SMI_POISSON_ROCFFT is a single-GPU back end; use SMI_POISSON_PRUNED for slabs");
  RocfftState* s = new RocfftState();
  p->rocfft = s;
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
  delete s;
  p->rocfft = nullptr;
  return 0;
}

static int rocfft_solve(smi_poisson* p, void* sol, const void* rhs) {
  smi_ctx* c = p->ctx;
  RocfftState* s = p->rocfft;
  const bool f32 = c->dtype == SMI_F32;
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
  if (backend == SMI_POISSON_ROCFFT) return nranks == 1;
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
  if (p->backend == SMI_POISSON_ROCFFT) return (size_t)p->rocfft->ncplx * 2 * smi_elem_size(p->ctx);
  return smi_pruned_greens_size(p);
}

extern "C" int smi_poisson_greens_download(smi_poisson* p, void* host) {
  SMI_REQUIRE(p && host, "smi_poisson_greens_download: null argument");
  if (p->backend == SMI_POISSON_ROCFFT)
    return smi_memcpy_d2h(p->ctx, host, p->rocfft->ghat, smi_poisson_greens_size(p));
  return smi_pruned_greens_download(p, host);
}
