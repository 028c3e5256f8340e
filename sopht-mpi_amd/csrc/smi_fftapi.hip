// Stand-alone real FFT plans and the actual <-> doubled domain copies of libsophtmi (gfx950):
// the two helper types the reference's Poisson solver is assembled from,
//   FFTMPI3D / FFTMPI2D                     (poisson_solver_3d/fft_mpi_3d.py:5-48, poisson_solver_2d/fft_mpi_2d.py:5-45)
//   MPIDomainDoublingCommunicator3D / 2D    (UnboundedPoissonSolverMPI3D.py:190-382, UnboundedPoissonSolverMPI2D.py:156-352)
// exposed on their own so that code written against them keeps working.  The flow step does not
// go through here: its Poisson solve never builds the doubled arrays (smi_fft.hip).
//
// Transforms run through hipFFT on the context's stream: forward real-to-complex unnormalised, backward
// complex-to-real normalised by 1 / (nz ny nx), the convention the reference pins against scipy.fft.rfftn / irfftn.
//
// Across z-slabs (3D, P ranks) the layouts are mpi4py-fft's for a slab decomposition: the real array is divided
// along z, (nz / P, ny, nx) per rank, its spectrum along y, (nz, ny / P, nx / 2 + 1) per rank.  Forward: batched 2D
// real-to-complex transforms of the local planes, one all-to-all (the y ranges of every plane go to their owners;
// what arrives is already the (nz, ny / P, nx / 2 + 1) array), transforms along z.  Backward is the mirror image.
// The doubled-domain copies send every rank's interior block to rank r / 2 of the same slab decomposition of the
// doubled grid, which holds two such blocks in the first quadrant of its planes; the upper half of the ranks holds
// padding only.
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

int smi_alltoall_impl(smi_ctx* c, const void* send, void* recv, size_t bytes_per_peer, hipStream_t stream);

struct smi_fft {
  smi_ctx* ctx;
  int nz, ny, nx;  // GLOBAL sizes; nz = 1: a 2D plan over (ny, nx)
  hipfftHandle fwd, inv;
  void* scratch;   // complex (nz, ny, nx / 2 + 1): hipFFT's C2R may overwrite its input
  long long nreal, ncplx;
  // slabs: plans of the local planes and along z, a second scratch array; nzl = nz / P, nyl = ny / P
  int P, nzl, nyl;
  hipfftHandle zfwd;   // complex-to-complex along z (both directions)
  void* scratch2;
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

// TO_DOUBLED: doubled[z + zoff][y][x] = actual[z + gz][y + g][x + g] on an (nz, ny, nx) block of an array of
// (.., dny, dnx) cells per plane and row (the first octant / quadrant of the doubled array: dny = 2 ny, dnx = 2 nx;
// a contiguous staging block: dny = ny, dnx = nx), the rest of that array is not touched; otherwise the copy back
// into the interior of the ghosted array (g = 0, plane = ny nx, px = nx: a contiguous block instead).
template <typename T, bool TO_DOUBLED>
__global__ void __launch_bounds__(kBlk) k_doubled_copy(T* __restrict__ dbl, T* __restrict__ act, int nz, int ny, int nx,
                                                      int gz, int g, int px, long long plane, int zoff, int dny,
                                                      int dnx) {
  const long long total = (long long)nz * ny * nx;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % nx);
    const long long r = i / nx;
    const int y = (int)(r % ny), z = (int)(r / ny);
    const long long a = (long long)(z + gz) * plane + (long long)(y + g) * px + (x + g);
    const long long d = ((long long)(z + zoff) * dny + y) * dnx + x;
    if (TO_DOUBLED)
      dbl[d] = act[a];
    else
      act[a] = dbl[d];
  }
}

// planes[zl][y][k] <-> blocks[s][zl][yl][k], y = s nyl + yl: the y ranges of the local planes, one block per owner
template <typename C2, bool TO_BLOCKS>
__global__ void __launch_bounds__(kBlk) k_y_blocks(C2* __restrict__ blocks, C2* __restrict__ planes, int nzl, int ny,
                                                  int nyl, int nk) {
  const long long total = (long long)nzl * ny * nk;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % nk);
    const long long r = i / nk;
    const int y = (int)(r % ny), zl = (int)(r / ny);
    const int s = y / nyl, yl = y - s * nyl;
    const long long b = (((long long)s * nzl + zl) * nyl + yl) * nk + k;
    if (TO_BLOCKS)
      blocks[b] = planes[i];
    else
      planes[i] = blocks[b];
  }
}
}  // namespace

static void fft_release(smi_fft* f) {
  if (f->fwd) hipfftDestroy(f->fwd);
  if (f->inv) hipfftDestroy(f->inv);
  if (f->zfwd) hipfftDestroy(f->zfwd);
  if (f->scratch) hipFree(f->scratch);
  if (f->scratch2) hipFree(f->scratch2);
  delete f;
}

// nz, ny, nx: GLOBAL grid of the transform; the context says how many z-slabs divide it
extern "C" int smi_fft_create(smi_fft** out, smi_ctx* c, int nz, int ny, int nx) {
  SMI_REQUIRE(out && c, "smi_fft_create: null argument");
  SMI_REQUIRE(nz >= 1 && ny >= 2 && nx >= 2, "smi_fft_create: bad grid (%d, %d, %d)", nz, ny, nx);
  const int P = c->nranks;
  SMI_REQUIRE(P == 1 || (nz > 1 && c->nranks_y == 1 && c->nranks_x == 1),
              "smi_fft_create: several ranks divide a 3D transform into z-slabs");
  SMI_REQUIRE(nz % P == 0 && ny % P == 0,
              "smi_fft_create: %d ranks must divide nz = %d (real array) and ny = %d (spectrum)", P, nz, ny);
  SMI_CHECK_HIP(hipSetDevice(c->device));
  smi_fft* f = new smi_fft();
  f->ctx = c;
  f->nz = nz;
  f->ny = ny;
  f->nx = nx;
  f->fwd = f->inv = f->zfwd = 0;
  f->scratch = f->scratch2 = nullptr;
  f->P = P;
  f->nzl = nz / P;
  f->nyl = ny / P;
  const int nk = nx / 2 + 1;
  f->nreal = (long long)f->nzl * ny * nx;       // local real cells
  f->ncplx = (long long)nz * f->nyl * nk;       // local spectrum (= nzl ny nk)
  const bool f32 = c->dtype == SMI_F32;
  hipfftResult r1, r2, r3 = HIPFFT_SUCCESS;
  if (P > 1) {
    int n2[2] = {ny, nx};
    r1 = hipfftPlanMany(&f->fwd, 2, n2, nullptr, 1, 0, nullptr, 1, 0, f32 ? HIPFFT_R2C : HIPFFT_D2Z, f->nzl);
    r2 = hipfftPlanMany(&f->inv, 2, n2, nullptr, 1, 0, nullptr, 1, 0, f32 ? HIPFFT_C2R : HIPFFT_Z2D, f->nzl);
    int n1[1] = {nz};
    const int stride = f->nyl * nk;
    r3 = hipfftPlanMany(&f->zfwd, 1, n1, n1, stride, 1, n1, stride, 1, f32 ? HIPFFT_C2C : HIPFFT_Z2Z, stride);
  } else if (nz == 1) {
    r1 = hipfftPlan2d(&f->fwd, ny, nx, f32 ? HIPFFT_R2C : HIPFFT_D2Z);
    r2 = hipfftPlan2d(&f->inv, ny, nx, f32 ? HIPFFT_C2R : HIPFFT_Z2D);
  } else {
    r1 = hipfftPlan3d(&f->fwd, nz, ny, nx, f32 ? HIPFFT_R2C : HIPFFT_D2Z);
    r2 = hipfftPlan3d(&f->inv, nz, ny, nx, f32 ? HIPFFT_C2R : HIPFFT_Z2D);
  }
  const size_t cbytes = (size_t)f->ncplx * 2 * smi_elem_size(c);
  hipError_t e = hipMalloc(&f->scratch, cbytes);
  if (e == hipSuccess && P > 1) e = hipMalloc(&f->scratch2, cbytes);
  bool ok = r1 == HIPFFT_SUCCESS && r2 == HIPFFT_SUCCESS && r3 == HIPFFT_SUCCESS && e == hipSuccess &&
            hipfftSetStream(f->fwd, c->stream) == HIPFFT_SUCCESS && hipfftSetStream(f->inv, c->stream) == HIPFFT_SUCCESS;
  if (ok && P > 1) ok = hipfftSetStream(f->zfwd, c->stream) == HIPFFT_SUCCESS;
  if (!ok) {
    fft_release(f);
    smi_set_error("smi_fft_create: plan or scratch allocation failed for (%d, %d, %d) on %d ranks", nz, ny, nx, P);
    return 1;
  }
  *out = f;
  return 0;
}

extern "C" int smi_fft_destroy(smi_fft* f) {
  if (!f) return 0;
  hipSetDevice(f->ctx->device);
  hipStreamSynchronize(f->ctx->stream);
  fft_release(f);
  return 0;
}

template <typename C2, bool TO_BLOCKS>
static int y_blocks(smi_fft* f, void* blocks, void* planes) {
  const int nk = f->nx / 2 + 1;
  k_y_blocks<C2, TO_BLOCKS><<<grid_for(f->ncplx), kBlk, 0, f->ctx->stream>>>((C2*)blocks, (C2*)planes, f->nzl, f->ny,
                                                                            f->nyl, nk);
  SMI_LAUNCH_CHECK();
  return 0;
}

// field: contiguous real (nz / P, ny, nx); fourier_field: contiguous complex (nz, ny / P, nx / 2 + 1)
extern "C" int smi_fft_forward(smi_fft* f, const void* field, void* fourier_field) {
  SMI_REQUIRE(f && field && fourier_field, "smi_fft_forward: null argument");
  smi_ctx* c = f->ctx;
  const bool f32 = c->dtype == SMI_F32;
  void* planes = f->P > 1 ? f->scratch : fourier_field;
  if (f32)
    SMI_CHECK_FFT3(hipfftExecR2C(f->fwd, (hipfftReal*)field, (hipfftComplex*)planes));
  else
    SMI_CHECK_FFT3(hipfftExecD2Z(f->fwd, (hipfftDoubleReal*)field, (hipfftDoubleComplex*)planes));
  if (f->P == 1) return 0;
  if (f32)
    SMI_CHECK((y_blocks<float2, true>(f, f->scratch2, f->scratch)));
  else
    SMI_CHECK((y_blocks<double2, true>(f, f->scratch2, f->scratch)));
  const size_t per_peer = (size_t)f->nzl * f->nyl * (f->nx / 2 + 1) * 2 * smi_elem_size(c);
  SMI_CHECK(smi_alltoall_impl(c, f->scratch2, fourier_field, per_peer, c->stream));
  if (f32)
    SMI_CHECK_FFT3(hipfftExecC2C(f->zfwd, (hipfftComplex*)fourier_field, (hipfftComplex*)fourier_field, HIPFFT_FORWARD));
  else
    SMI_CHECK_FFT3(hipfftExecZ2Z(f->zfwd, (hipfftDoubleComplex*)fourier_field, (hipfftDoubleComplex*)fourier_field,
                                 HIPFFT_FORWARD));
  return 0;
}

// inv_fourier_field = irfftn(fourier_field) including the 1 / N normalisation; fourier_field is preserved
extern "C" int smi_fft_backward(smi_fft* f, const void* fourier_field, void* inv_fourier_field) {
  SMI_REQUIRE(f && fourier_field && inv_fourier_field, "smi_fft_backward: null argument");
  smi_ctx* c = f->ctx;
  const bool f32 = c->dtype == SMI_F32;
  const size_t cbytes = (size_t)f->ncplx * 2 * smi_elem_size(c);
  if (f->P == 1) {
    SMI_CHECK_HIP(hipMemcpyAsync(f->scratch, fourier_field, cbytes, hipMemcpyDeviceToDevice, c->stream));
  } else {
    // along z (out of place, so that the input survives), back to the owners of the planes, y ranges reassembled
    if (f32)
      SMI_CHECK_FFT3(hipfftExecC2C(f->zfwd, (hipfftComplex*)fourier_field, (hipfftComplex*)f->scratch2, HIPFFT_BACKWARD));
    else
      SMI_CHECK_FFT3(hipfftExecZ2Z(f->zfwd, (hipfftDoubleComplex*)fourier_field, (hipfftDoubleComplex*)f->scratch2,
                                   HIPFFT_BACKWARD));
    const size_t per_peer = (size_t)f->nzl * f->nyl * (f->nx / 2 + 1) * 2 * smi_elem_size(c);
    SMI_CHECK(smi_alltoall_impl(c, f->scratch2, f->scratch, per_peer, c->stream));
    // scratch holds blocks [s][zl][yl][k]; the planes go to scratch2, whose contents have been sent
    if (f32)
      SMI_CHECK((y_blocks<float2, false>(f, f->scratch, f->scratch2)));
    else
      SMI_CHECK((y_blocks<double2, false>(f, f->scratch, f->scratch2)));
  }
  void* planes = f->P > 1 ? f->scratch2 : f->scratch;
  if (f32)
    SMI_CHECK_FFT3(hipfftExecC2R(f->inv, (hipfftComplex*)planes, (hipfftReal*)inv_fourier_field));
  else
    SMI_CHECK_FFT3(hipfftExecZ2D(f->inv, (hipfftDoubleComplex*)planes, (hipfftDoubleReal*)inv_fourier_field));
  const double total = (double)f->nz * (double)f->ny * (double)f->nx;
  SMI_DISPATCH(c, {
    k_scale_real<T><<<grid_for(f->nreal), kBlk, 0, c->stream>>>((T*)inv_fourier_field, f->nreal, (T)(1.0 / total));
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ actual <-> doubled domain
extern "C" int smi_set_p2p_callback(smi_ctx* c, smi_p2p_cb cb) {
  SMI_REQUIRE(c, "smi_set_p2p_callback: null context");
  c->p2p_cb = cb;
  return 0;
}

// A few messages of `bytes` bytes each between device buffers: sends[i] goes to rank send_peers[i], recvs[j] comes
// from rank recv_peers[j] (a message to this rank itself is a device copy).  One RCCL group, or the host-staged
// callback of the tests.
int smi_p2p_exchange(smi_ctx* c, int nsend, const int* send_peers, const void* const* sends, int nrecv,
                        const int* recv_peers, void* const* recvs, size_t bytes) {
  if (c->p2p_cb) {
    SMI_CHECK_HIP(hipStreamSynchronize(c->stream));
    const int rc = c->p2p_cb(c->exchange_user, nsend, send_peers, sends, nrecv, recv_peers, recvs, bytes);
    SMI_REQUIRE(rc == 0, "point-to-point callback failed (%d)", rc);
    return 0;
  }
  SMI_REQUIRE(c->comm_ready, "point-to-point messages across ranks need smi_comm_init or a point-to-point callback");
  for (int i = 0; i < nsend; ++i)
    if (send_peers[i] == c->rank)
      for (int j = 0; j < nrecv; ++j)
        if (recv_peers[j] == c->rank)
          SMI_CHECK_HIP(hipMemcpyAsync(recvs[j], sends[i], bytes, hipMemcpyDeviceToDevice, c->stream));
  SMI_CHECK_NCCL(ncclGroupStart());
  for (int i = 0; i < nsend; ++i)
    if (send_peers[i] != c->rank) SMI_CHECK_NCCL(ncclSend(sends[i], bytes, ncclChar, send_peers[i], c->comm, c->stream));
  for (int j = 0; j < nrecv; ++j)
    if (recv_peers[j] != c->rank) SMI_CHECK_NCCL(ncclRecv(recvs[j], bytes, ncclChar, recv_peers[j], c->comm, c->stream));
  SMI_CHECK_NCCL(ncclGroupEnd());
  return 0;
}

// Across P z-slabs (P even) the doubled grid is divided over the same ranks: rank r of the doubled decomposition
// holds 2 nzl planes, the blocks of ranks 2 r and 2 r + 1 of the actual decomposition one above the other; ranks
// P / 2 and up hold padding only (UnboundedPoissonSolverMPI3D.py:314-382: one send to coords / 2, two receives
// on the lower half of the ranks).  Blocks travel through three contiguous staging buffers kept by the context.
static int doubled_stage(smi_ctx* c, size_t bytes) {
  if (c->dbl_stage_cap >= bytes) return 0;
  for (void*& b : c->dbl_stage) {
    if (b) SMI_CHECK_HIP(hipFree(b));
    b = nullptr;
  }
  c->dbl_stage_cap = 0;
  for (void*& b : c->dbl_stage) SMI_CHECK_HIP(hipMalloc(&b, bytes));
  c->dbl_stage_cap = bytes;
  return 0;
}

template <bool TO_DOUBLED>
static int doubled_copy_slabs(smi_ctx* c, void* local_field, void* local_doubled_field) {
  const int P = c->nranks, r = c->rank;
  SMI_REQUIRE(c->nranks_y == 1 && c->nranks_x == 1 && c->nz > 1 && P % 2 == 0,
              "the doubled-domain copies across ranks serve an even number of z-slabs (got %d ranks)", P);
  const long long n = (long long)c->nz * c->ny * c->nx;
  const size_t bytes = (size_t)n * smi_elem_size(c);
  SMI_CHECK(doubled_stage(c, bytes));
  const long long dplane = 4LL * c->ny * c->nx;
  const bool lower = r < P / 2;
  const int parent = r / 2, kids[2] = {2 * r, 2 * r + 1};
  // block <-> interior of the ghosted field (stage 0), blocks <-> the two halves of the doubled planes (1, 2)
  auto field_block = [&](bool pack) {
    SMI_DISPATCH(c, {
      if (pack)
        k_doubled_copy<T, true><<<grid_for(n), kBlk, 0, c->stream>>>((T*)c->dbl_stage[0], (T*)local_field, c->nz, c->ny,
                                                                    c->nx, c->g, c->g, c->px, (long long)c->plane, 0,
                                                                    c->ny, c->nx);
      else
        k_doubled_copy<T, false><<<grid_for(n), kBlk, 0, c->stream>>>((T*)c->dbl_stage[0], (T*)local_field, c->nz, c->ny,
                                                                     c->nx, c->g, c->g, c->px, (long long)c->plane, 0,
                                                                     c->ny, c->nx);
    });
  };
  auto doubled_block = [&](int h, bool pack) {
    SMI_DISPATCH(c, {
      if (pack)
        k_doubled_copy<T, true><<<grid_for(n), kBlk, 0, c->stream>>>((T*)c->dbl_stage[1 + h], (T*)local_doubled_field,
                                                                    c->nz, c->ny, c->nx, h * c->nz, 0, 2 * c->nx, dplane,
                                                                    0, c->ny, c->nx);
      else
        k_doubled_copy<T, false><<<grid_for(n), kBlk, 0, c->stream>>>((T*)c->dbl_stage[1 + h], (T*)local_doubled_field,
                                                                     c->nz, c->ny, c->nx, h * c->nz, 0, 2 * c->nx,
                                                                     dplane, 0, c->ny, c->nx);
    });
  };
  if (TO_DOUBLED) {
    field_block(true);
    SMI_LAUNCH_CHECK();
    const void* sends[1] = {c->dbl_stage[0]};
    void* recvs[2] = {c->dbl_stage[1], c->dbl_stage[2]};
    SMI_CHECK(smi_p2p_exchange(c, 1, &parent, sends, lower ? 2 : 0, kids, recvs, bytes));
    if (lower) {
      doubled_block(0, false);
      doubled_block(1, false);
      SMI_LAUNCH_CHECK();
    }
  } else {
    if (lower) {
      doubled_block(0, true);
      doubled_block(1, true);
      SMI_LAUNCH_CHECK();
    }
    const void* sends[2] = {c->dbl_stage[1], c->dbl_stage[2]};
    void* recvs[1] = {c->dbl_stage[0]};
    SMI_CHECK(smi_p2p_exchange(c, lower ? 2 : 0, kids, sends, 1, &parent, recvs, bytes));
    field_block(false);
    SMI_LAUNCH_CHECK();
  }
  return 0;
}

// copy_to_doubled_domain (UnboundedPoissonSolverMPI3D.py:314-347): interior of the ghosted local field
// -> first octant of the (2nz, 2ny, 2nx) array (first quadrant of (2ny, 2nx) on a 2D context).
extern "C" int smi_copy_to_doubled_domain(smi_ctx* c, const void* local_field, void* local_doubled_field) {
  SMI_REQUIRE(c && local_field && local_doubled_field, "smi_copy_to_doubled_domain: null argument");
  if (c->nranks > 1) return doubled_copy_slabs<true>(c, (void*)local_field, local_doubled_field);
  const bool two_d = c->nz == 1;
  const long long n = (long long)c->nz * c->ny * c->nx;
  SMI_DISPATCH(c, {
    k_doubled_copy<T, true><<<grid_for(n), kBlk, 0, c->stream>>>(
        (T*)local_doubled_field, (T*)local_field, c->nz, c->ny, c->nx, two_d ? 0 : c->g, c->g, c->px,
        (long long)c->plane, 0, 2 * c->ny, 2 * c->nx);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

// copy_from_doubled_domain (:349-382): first octant of the doubled array -> interior of the ghosted field
extern "C" int smi_copy_from_doubled_domain(smi_ctx* c, const void* local_doubled_field, void* local_field) {
  SMI_REQUIRE(c && local_field && local_doubled_field, "smi_copy_from_doubled_domain: null argument");
  if (c->nranks > 1) return doubled_copy_slabs<false>(c, local_field, (void*)local_doubled_field);
  const bool two_d = c->nz == 1;
  const long long n = (long long)c->nz * c->ny * c->nx;
  SMI_DISPATCH(c, {
    k_doubled_copy<T, false><<<grid_for(n), kBlk, 0, c->stream>>>(
        (T*)local_doubled_field, (T*)local_field, c->nz, c->ny, c->nx, two_d ? 0 : c->g, c->g, c->px,
        (long long)c->plane, 0, 2 * c->ny, 2 * c->nx);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}
