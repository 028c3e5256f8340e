// Stand-alone real FFT plans and the actual <-> doubled domain copies of libsophtmi (gfx950):
// the two helper types the reference's Poisson solver is assembled from,
//   FFTMPI3D / FFTMPI2D                     (poisson_solver_3d/fft_mpi_3d.py:5-48, poisson_solver_2d/fft_mpi_2d.py:5-45)
//   MPIDomainDoublingCommunicator3D / 2D    (UnboundedPoissonSolverMPI3D.py:190-382, UnboundedPoissonSolverMPI2D.py:156-352)
// exposed on their own so that code written against them keeps working.  The flow step does not
// go through here: its Poisson solve never builds the doubled arrays (smi_fft.hip).
//
// Single rank.  Transforms run through hipFFT on the context's stream: forward real-to-complex
// unnormalised, backward complex-to-real normalised by 1 / (nz ny nx), the convention the
// reference pins against scipy.fft.rfftn / irfftn.
#include <hipfft/hipfft.h>

#include "smi_common.h"

#define SMI_CHECK_FFT3(expr)                                                              \
  do {                                                                                    \
    hipfftResult _e = (expr);                                                             \
    if (_e != HIPFFT_SUCCESS) {                                                           \
      smi_set_error("%s:%d: %s -> hipfft error %d", __FILE__, __LINE__, #expr, (int)_e);  \
      return 1;                                                                           \
    }                                                                                     \
  } while (0)

struct smi_fft {
  smi_ctx* ctx;
  int nz, ny, nx;  // nz = 1: a 2D plan over (ny, nx)
  hipfftHandle fwd, inv;
  void* scratch;   // complex (nz, ny, nx / 2 + 1): hipFFT's C2R may overwrite its input
  long long nreal, ncplx;
};

namespace {
constexpr int kBlk = 256;
inline int grid_for(long long n) {
  long long b = (n + kBlk - 1) / kBlk;
  return (int)(b < (1 << 20) ? (b ? b : 1) : (1 << 20));
}

template <typename T>
__global__ void __launch_bounds__(kBlk) k_scale_real(T* __restrict__ a, long long n, T s) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    a[i] *= s;
}

// TO_DOUBLED: doubled[z][y][x] = actual[z + gz][y + g][x + g] on the first octant (quadrant), the rest of
// the doubled array is not touched; otherwise the copy back into the interior of the ghosted array.
template <typename T, bool TO_DOUBLED>
__global__ void __launch_bounds__(kBlk) k_doubled_copy(T* __restrict__ dbl, T* __restrict__ act, int nz, int ny, int nx,
                                                      int gz, int g, int px, long long plane) {
  const long long total = (long long)nz * ny * nx;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % nx);
    const long long r = i / nx;
    const int y = (int)(r % ny), z = (int)(r / ny);
    const long long a = (long long)(z + gz) * plane + (long long)(y + g) * px + (x + g);
    const long long d = ((long long)z * 2 * ny + y) * 2 * nx + x;
    if (TO_DOUBLED)
      dbl[d] = act[a];
    else
      act[a] = dbl[d];
  }
}
}  // namespace

extern "C" int smi_fft_create(smi_fft** out, smi_ctx* c, int nz, int ny, int nx) {
  SMI_REQUIRE(out && c, "smi_fft_create: null argument");
  SMI_REQUIRE(nz >= 1 && ny >= 2 && nx >= 2, "smi_fft_create: bad grid (%d, %d, %d)", nz, ny, nx);
  SMI_REQUIRE(c->nranks == 1, "smi_fft_create: the stand-alone FFT plans are single rank; the slab-distributed "
                              "transform is part of the pruned Poisson solver");
  SMI_CHECK_HIP(hipSetDevice(c->device));
  smi_fft* f = new smi_fft();
  f->ctx = c;
  f->nz = nz;
  f->ny = ny;
  f->nx = nx;
  f->fwd = f->inv = 0;
  f->scratch = nullptr;
  f->nreal = (long long)nz * ny * nx;
  f->ncplx = (long long)nz * ny * (nx / 2 + 1);
  const bool f32 = c->dtype == SMI_F32;
  hipfftResult r1, r2;
  if (nz == 1) {
    r1 = hipfftPlan2d(&f->fwd, ny, nx, f32 ? HIPFFT_R2C : HIPFFT_D2Z);
    r2 = hipfftPlan2d(&f->inv, ny, nx, f32 ? HIPFFT_C2R : HIPFFT_Z2D);
  } else {
    r1 = hipfftPlan3d(&f->fwd, nz, ny, nx, f32 ? HIPFFT_R2C : HIPFFT_D2Z);
    r2 = hipfftPlan3d(&f->inv, nz, ny, nx, f32 ? HIPFFT_C2R : HIPFFT_Z2D);
  }
  hipError_t e = hipMalloc(&f->scratch, (size_t)f->ncplx * 2 * smi_elem_size(c));
  if (r1 != HIPFFT_SUCCESS || r2 != HIPFFT_SUCCESS || e != hipSuccess ||
      hipfftSetStream(f->fwd, c->stream) != HIPFFT_SUCCESS || hipfftSetStream(f->inv, c->stream) != HIPFFT_SUCCESS) {
    if (f->fwd) hipfftDestroy(f->fwd);
    if (f->inv) hipfftDestroy(f->inv);
    if (f->scratch) hipFree(f->scratch);
    delete f;
    smi_set_error("smi_fft_create: plan or scratch allocation failed for (%d, %d, %d)", nz, ny, nx);
    return 1;
  }
  *out = f;
  return 0;
}

extern "C" int smi_fft_destroy(smi_fft* f) {
  if (!f) return 0;
  hipSetDevice(f->ctx->device);
  hipStreamSynchronize(f->ctx->stream);
  if (f->fwd) hipfftDestroy(f->fwd);
  if (f->inv) hipfftDestroy(f->inv);
  hipFree(f->scratch);
  delete f;
  return 0;
}

// field: contiguous real (nz, ny, nx); fourier_field: contiguous complex (nz, ny, nx / 2 + 1)
extern "C" int smi_fft_forward(smi_fft* f, const void* field, void* fourier_field) {
  SMI_REQUIRE(f && field && fourier_field, "smi_fft_forward: null argument");
  if (f->ctx->dtype == SMI_F32)
    SMI_CHECK_FFT3(hipfftExecR2C(f->fwd, (hipfftReal*)field, (hipfftComplex*)fourier_field));
  else
    SMI_CHECK_FFT3(hipfftExecD2Z(f->fwd, (hipfftDoubleReal*)field, (hipfftDoubleComplex*)fourier_field));
  return 0;
}

// inv_fourier_field = irfftn(fourier_field) including the 1 / N normalisation; fourier_field is preserved
extern "C" int smi_fft_backward(smi_fft* f, const void* fourier_field, void* inv_fourier_field) {
  SMI_REQUIRE(f && fourier_field && inv_fourier_field, "smi_fft_backward: null argument");
  smi_ctx* c = f->ctx;
  SMI_CHECK_HIP(hipMemcpyAsync(f->scratch, fourier_field, (size_t)f->ncplx * 2 * smi_elem_size(c),
                               hipMemcpyDeviceToDevice, c->stream));
  if (c->dtype == SMI_F32)
    SMI_CHECK_FFT3(hipfftExecC2R(f->inv, (hipfftComplex*)f->scratch, (hipfftReal*)inv_fourier_field));
  else
    SMI_CHECK_FFT3(hipfftExecZ2D(f->inv, (hipfftDoubleComplex*)f->scratch, (hipfftDoubleReal*)inv_fourier_field));
  SMI_DISPATCH(c, {
    k_scale_real<T><<<grid_for(f->nreal), kBlk, 0, c->stream>>>((T*)inv_fourier_field, f->nreal,
                                                               (T)(1.0 / (double)f->nreal));
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

// copy_to_doubled_domain (UnboundedPoissonSolverMPI3D.py:314-347): interior of the ghosted local field
// -> first octant of the (2nz, 2ny, 2nx) array (first quadrant of (2ny, 2nx) on a 2D context).
extern "C" int smi_copy_to_doubled_domain(smi_ctx* c, const void* local_field, void* local_doubled_field) {
  SMI_REQUIRE(c && local_field && local_doubled_field, "smi_copy_to_doubled_domain: null argument");
  SMI_REQUIRE(c->nranks == 1, "smi_copy_to_doubled_domain: single rank only");
  const bool two_d = c->nz == 1;
  const long long n = (long long)c->nz * c->ny * c->nx;
  SMI_DISPATCH(c, {
    k_doubled_copy<T, true><<<grid_for(n), kBlk, 0, c->stream>>>(
        (T*)local_doubled_field, (T*)local_field, c->nz, c->ny, c->nx, two_d ? 0 : c->g, c->g, c->px,
        (long long)c->plane);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

// copy_from_doubled_domain (:349-382): first octant of the doubled array -> interior of the ghosted field
extern "C" int smi_copy_from_doubled_domain(smi_ctx* c, const void* local_doubled_field, void* local_field) {
  SMI_REQUIRE(c && local_field && local_doubled_field, "smi_copy_from_doubled_domain: null argument");
  SMI_REQUIRE(c->nranks == 1, "smi_copy_from_doubled_domain: single rank only");
  const bool two_d = c->nz == 1;
  const long long n = (long long)c->nz * c->ny * c->nx;
  SMI_DISPATCH(c, {
    k_doubled_copy<T, false><<<grid_for(n), kBlk, 0, c->stream>>>(
        (T*)local_doubled_field, (T*)local_field, c->nz, c->ny, c->nx, two_d ? 0 : c->g, c->g, c->px,
        (long long)c->plane);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}
