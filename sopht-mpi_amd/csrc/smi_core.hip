// Context, memory, elementwise operators and reductions of libsophtmi (gfx950).
#include <cstdlib>

#include "smi_common.h"
#include "smi_prof.h"

static thread_local char g_err[1024] = "";

void smi_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* smi_last_error(void) { return g_err; }
extern "C" int smi_version(void) { return 100; }

extern "C" int smi_device_count(int* count) {
  SMI_REQUIRE(count, "smi_device_count: null output");
  SMI_CHECK_HIP(hipGetDeviceCount(count));
  return 0;
}

// streams, events and the small device / pinned buffers of a context whose sizes are set
static int ctx_allocate(smi_ctx* c) {
  SMI_CHECK_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  SMI_CHECK_HIP(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  SMI_CHECK_HIP(hipEventCreateWithFlags(&c->ev_a, hipEventDisableTiming));
  SMI_CHECK_HIP(hipEventCreateWithFlags(&c->ev_b, hipEventDisableTiming));
  SMI_CHECK_HIP(hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming));
  SMI_CHECK_HIP(hipMalloc(&c->red_partials, sizeof(double) * (kRedBlocks + 8)));
  SMI_CHECK_HIP(hipHostMalloc((void**)&c->red_host, kPinnedBytes, hipHostMallocDefault));
  SMI_CHECK_HIP(hipMalloc(&c->fac_dev, sizeof(double) * 6 * 64));
  SMI_CHECK_HIP(hipMalloc(&c->umax_slots, sizeof(unsigned long long) * kUmaxSlots));
  for (int i = 0; i < 2; ++i) SMI_CHECK_HIP(hipMalloc(&c->halo_recv[i], c->halo_cap));
  return 0;
}

// layers_needed: interior layers per ghost layer a block must have along every axis - 2 for the operators (their
// bands next to a face and the planes they send must not overlap), 1 for a block that only exchanges ghost cells
static int ctx_create(smi_ctx** out, int device, int dtype, int nz, int ny, int nx, int ghost, int rank, int nranks,
                      int layers_needed);

extern "C" int smi_ctx_create(smi_ctx** out, int device, int dtype, int nz, int ny, int nx, int ghost,
                              int rank, int nranks) {
  return ctx_create(out, device, dtype, nz, ny, nx, ghost, rank, nranks, 2);
}

extern "C" int smi_ctx_create_block(smi_ctx** out, int device, int dtype, int nz, int ny, int nx, int ghost, int rank,
                                    int nranks, const int* coords_zyx, const int* topo_zyx, const int* periodic_zyx) {
  SMI_CHECK(ctx_create(out, device, dtype, nz, ny, nx, ghost, rank, nranks, 1));
  const int rc = smi_ctx_set_block(*out, coords_zyx, topo_zyx, periodic_zyx);
  if (rc != 0) {
    smi_ctx_destroy(*out);
    *out = nullptr;
  }
  return rc;
}

static int ctx_create(smi_ctx** out, int device, int dtype, int nz, int ny, int nx, int ghost, int rank, int nranks,
                      int layers_needed) {
  SMI_REQUIRE(out, "smi_ctx_create: null output");
  SMI_REQUIRE(dtype == SMI_F32 || dtype == SMI_F64, "smi_ctx_create: invalid dtype %d", dtype);
  SMI_REQUIRE(nz > 0 && ny > 0 && nx > 0, "smi_ctx_create: invalid grid %d %d %d", nz, ny, nx);
  SMI_REQUIRE(ghost >= 1 && ghost <= 8, "smi_ctx_create: ghost size %d must be in [1, 8]", ghost);
  SMI_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "smi_ctx_create: invalid rank %d of %d", rank,
              nranks);
  // nz == 1 is the plane of the 2D operators, which ignore the z extent (several ranks: slabs of rows)
  const int least = layers_needed * ghost;
  SMI_REQUIRE((nz == 1 || nz >= least) && ny >= least && nx >= least,
              "smi_ctx_create: local grid %d %d %d smaller than %d*ghost", nz, ny, nx, layers_needed);
  SMI_CHECK_HIP(hipSetDevice(device));
  SMI_CHECK(smi_pruned_scratch_init(device));
  smi_ctx* c = new smi_ctx();
  memset(c, 0, sizeof(*c));
  c->device = device;
  c->dtype = dtype;
  c->nz = nz;
  c->ny = ny;
  c->nx = nx;
  c->g = ghost;
  c->pz = nz + 2 * ghost;
  c->py = ny + 2 * ghost;
  c->px = nx + 2 * ghost;
  c->plane = (size_t)c->py * c->px;
  c->vol = c->plane * c->pz;
  c->rank = rank;
  c->nranks = nranks;
  c->phys_zlo = rank == 0;
  c->phys_zhi = rank == nranks - 1;
  c->rank_z = rank;
  c->nranks_z = nranks;
  c->rank_y = 0;
  c->nranks_y = 1;
  c->phys_ylo = c->phys_yhi = 1;
  c->rank_x = 0;
  c->nranks_x = 1;
  c->phys_xlo = c->phys_xhi = 1;
  if (nz == 1 && nranks > 1) {
    // 2D: the ranks divide the rows (mpi_utils_2d.py:9-84 with the default rank_distribution)
    c->rank_z = 0;
    c->nranks_z = 1;
    c->rank_y = rank;
    c->nranks_y = nranks;
    c->phys_zlo = c->phys_zhi = 1;
    c->phys_ylo = rank == 0;
    c->phys_yhi = rank == nranks - 1;
  }
  // environment switches are read here, once, and never on the step path
  c->opt_no_fused = getenv("SMI_NO_FUSED_ADVECT_DIFFUSE") != nullptr;
  c->opt_split_launch = getenv("SMI_FUSED_SPLIT_LAUNCH") != nullptr;
  c->opt_check_forcing_support = getenv("SMI_CHECK_FORCING_SUPPORT") != nullptr;
  if (const char* e = getenv("SMI_FUSED_ZCHUNK")) c->opt_fused_zchunk = atoi(e) > 0 ? atoi(e) : 32;
  // One chunk per transpose unless asked otherwise.  At eight ranks two chunks halve what is exposed of the first
  // forward and the last backward transpose (about 0.15 ms of a 2 ms step at 512^3) and cost 0.09 ms of shorter
  // launches, a margin that rests on RCCL groups being cheap, which no multi-GPU node was at hand to measure; with
  // fewer ranks the links are the bound by a wide margin and chunks only add groups.
  c->opt_transpose_chunks = 1;
  if (const char* e = getenv("SMI_FFT_TRANSPOSE_CHUNKS")) {
    const int n = atoi(e);
    SMI_REQUIRE(n == 1 || n == 2 || n == 4, "SMI_FFT_TRANSPOSE_CHUNKS=%s: must be 1, 2 or 4", e);
    c->opt_transpose_chunks = n;
  }
  // staging for the ghost sum (and the host-staged plane upload): 3 components of g planes
  c->halo_cap = (size_t)3 * ghost * c->plane * smi_elem_size(c);
  const int rc = ctx_allocate(c);
  if (rc) {
    smi_ctx_destroy(c);  // releases whatever was created before the failure
    return rc;
  }
  *out = c;
  return 0;
}

// Changes one of the switches smi_ctx_create read from the environment (tests, measurements).
extern "C" int smi_ctx_set_option(smi_ctx* c, const char* name, int value) {
  SMI_REQUIRE(c && name, "smi_ctx_set_option: null argument");
  if (!strcmp(name, "no_fused_advect_diffuse")) {
    c->opt_no_fused = value != 0;
  } else if (!strcmp(name, "fused_split_launch")) {
    c->opt_split_launch = value != 0;
  } else if (!strcmp(name, "check_forcing_support")) {
    c->opt_check_forcing_support = value != 0;
  } else if (!strcmp(name, "fused_zchunk")) {
    SMI_REQUIRE(value >= 0, "smi_ctx_set_option: fused_zchunk %d must not be negative", value);
    c->opt_fused_zchunk = value;
  } else if (!strcmp(name, "transpose_chunks")) {
    SMI_REQUIRE(value == 1 || value == 2 || value == 4, "smi_ctx_set_option: transpose_chunks %d must be 1, 2 or 4",
                value);
    c->opt_transpose_chunks = value;
  } else {
    SMI_REQUIRE(false, "smi_ctx_set_option: unknown option '%s'", name);
  }
  return 0;
}

extern "C" int smi_ctx_get_option(smi_ctx* c, const char* name, int* value) {
  SMI_REQUIRE(c && name && value, "smi_ctx_get_option: null argument");
  if (!strcmp(name, "no_fused_advect_diffuse")) {
    *value = c->opt_no_fused;
  } else if (!strcmp(name, "fused_split_launch")) {
    *value = c->opt_split_launch;
  } else if (!strcmp(name, "check_forcing_support")) {
    *value = c->opt_check_forcing_support;
  } else if (!strcmp(name, "fused_zchunk")) {
    *value = c->opt_fused_zchunk;
  } else if (!strcmp(name, "transpose_chunks")) {
    *value = c->opt_transpose_chunks;
  } else {
    SMI_REQUIRE(false, "smi_ctx_get_option: unknown option '%s'", name);
  }
  return 0;
}

extern "C" int smi_ctx_set_grid3(smi_ctx* c, int rank_z, int nranks_z, int rank_y, int nranks_y, int rank_x,
                                 int nranks_x) {
  SMI_REQUIRE(c, "smi_ctx_set_grid3: null context");
  SMI_REQUIRE(nranks_z >= 1 && nranks_y >= 1 && nranks_x >= 1 && nranks_z * nranks_y * nranks_x == c->nranks,
              "smi_ctx_set_grid3: a %d x %d x %d rank grid does not hold the %d ranks of this context", nranks_z,
              nranks_y, nranks_x, c->nranks);
  SMI_REQUIRE(rank_z >= 0 && rank_z < nranks_z && rank_y >= 0 && rank_y < nranks_y && rank_x >= 0 && rank_x < nranks_x &&
                  (rank_z * nranks_y + rank_y) * nranks_x + rank_x == c->rank,
              "smi_ctx_set_grid3: position (%d, %d, %d) is not rank %d of a %d x %d x %d grid", rank_z, rank_y, rank_x,
              c->rank, nranks_z, nranks_y, nranks_x);
  SMI_REQUIRE(c->nz > 1 || nranks_x == 1, "smi_ctx_set_grid3: the 2D operators divide rows only");
  c->rank_z = rank_z;
  c->nranks_z = nranks_z;
  c->rank_y = rank_y;
  c->nranks_y = nranks_y;
  c->rank_x = rank_x;
  c->nranks_x = nranks_x;
  c->phys_zlo = rank_z == 0;
  c->phys_zhi = rank_z == nranks_z - 1;
  c->phys_ylo = rank_y == 0;
  c->phys_yhi = rank_y == nranks_y - 1;
  c->phys_xlo = rank_x == 0;
  c->phys_xhi = rank_x == nranks_x - 1;
  if ((nranks_y > 1 || nranks_x > 1) && !c->ybuf[0]) {
    // packed y or x faces: 3 components of g rows (columns) of every plane
    SMI_CHECK_HIP(hipSetDevice(c->device));
    c->ybuf_cap = (size_t)3 * c->pz * c->g * (c->px > c->py ? c->px : c->py) * smi_elem_size(c);
    for (int i = 0; i < 4; ++i) SMI_CHECK_HIP(hipMalloc(&c->ybuf[i], c->ybuf_cap));
  }
  return 0;
}

extern "C" int smi_ctx_set_pencil(smi_ctx* c, int rank_z, int nranks_z, int rank_y, int nranks_y) {
  SMI_REQUIRE(c, "smi_ctx_set_pencil: null context");
  return smi_ctx_set_grid3(c, rank_z, nranks_z, rank_y, nranks_y, 0, 1);
}

extern "C" int smi_ctx_destroy(smi_ctx* c) {
  if (!c) return 0;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->comm_stream) hipStreamSynchronize(c->comm_stream);
  if (c->comm_ready) ncclCommDestroy(c->comm);
  for (int i = 0; i < 2; ++i) {
    if (c->halo_recv[i]) hipFree(c->halo_recv[i]);
  }
  for (int i = 0; i < 4; ++i) {
    if (c->ybuf[i]) hipFree(c->ybuf[i]);
  }
  for (void* b : c->dbl_stage)
    if (b) hipFree(b);
  for (void* b : c->blk_buf)
    if (b) hipFree(b);
  delete c->prof;
  if (c->red_partials) hipFree(c->red_partials);
  if (c->fac_dev) hipFree(c->fac_dev);
  if (c->umax_slots) hipFree(c->umax_slots);
  if (c->red_host) hipHostFree(c->red_host);
  if (c->ev_a) hipEventDestroy(c->ev_a);
  if (c->ev_b) hipEventDestroy(c->ev_b);
  if (c->ev_order) hipEventDestroy(c->ev_order);
  if (c->stream) hipStreamDestroy(c->stream);
  if (c->comm_stream) hipStreamDestroy(c->comm_stream);
  delete c;
  return 0;
}

// Orders the streams of two contexts of one process without a host synchronisation: everything queued on `other` so
// far completes before anything queued on `waiter` from now on starts.
extern "C" int smi_ctx_wait_for(smi_ctx* waiter, smi_ctx* other) {
  SMI_REQUIRE(waiter && other, "smi_ctx_wait_for: null context");
  if (waiter == other) return 0;
  SMI_REQUIRE(waiter->device == other->device, "smi_ctx_wait_for: the two contexts live on different devices");
  SMI_CHECK_HIP(hipEventRecord(other->ev_order, other->stream));
  SMI_CHECK_HIP(hipStreamWaitEvent(waiter->stream, other->ev_order, 0));
  return 0;
}

extern "C" int smi_sync(smi_ctx* c) {
  SMI_REQUIRE(c, "smi_sync: null context");
  SMI_CHECK_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

// ------------------------------------------------------------------ memory
extern "C" int smi_malloc(smi_ctx* c, size_t bytes, void** dptr) {
  SMI_REQUIRE(c && dptr, "smi_malloc: null argument");
  SMI_CHECK_HIP(hipSetDevice(c->device));
  SMI_CHECK_HIP(hipMalloc(dptr, bytes ? bytes : 16));
  return 0;
}
extern "C" int smi_free(smi_ctx* c, void* dptr) {
  SMI_REQUIRE(c, "smi_free: null context");
  if (!dptr) return 0;
  SMI_CHECK_HIP(hipStreamSynchronize(c->stream));
  SMI_CHECK_HIP(hipFree(dptr));
  return 0;
}
// Transfers of a few kilobytes (per-frame loads of a body, node counts, scalars) go through the context's pinned
// host words: a copy from or into pageable memory is staged by the runtime and costs some ten microseconds more, which
// a loop that reads a handful of small results per step notices.
extern "C" int smi_memcpy_h2d(smi_ctx* c, void* dst, const void* src, size_t bytes) {
  SMI_REQUIRE(c && dst && src, "smi_memcpy_h2d: null argument");
  if (bytes <= kPinnedStageBytes) {
    char* stage = reinterpret_cast<char*>(c->red_host) + kPinnedStageOffset;
    // (the last transfer staged here has completed: every user of the words synchronises before it returns)
    memcpy(stage, src, bytes);
    SMI_CHECK_HIP(hipMemcpyAsync(dst, stage, bytes, hipMemcpyHostToDevice, c->stream));
  } else {
    SMI_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  }
  SMI_CHECK_HIP(hipStreamSynchronize(c->stream));
  return 0;
}
extern "C" int smi_memcpy_d2h(smi_ctx* c, void* dst, const void* src, size_t bytes) {
  SMI_REQUIRE(c && dst && src, "smi_memcpy_d2h: null argument");
  if (bytes <= kPinnedStageBytes) {
    char* stage = reinterpret_cast<char*>(c->red_host) + kPinnedStageOffset;
    SMI_CHECK_HIP(hipMemcpyAsync(stage, src, bytes, hipMemcpyDeviceToHost, c->stream));
    SMI_CHECK_HIP(hipStreamSynchronize(c->stream));
    memcpy(dst, stage, bytes);
    return 0;
  }
  SMI_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  SMI_CHECK_HIP(hipStreamSynchronize(c->stream));
  return 0;
}
extern "C" int smi_memcpy_d2d(smi_ctx* c, void* dst, const void* src, size_t bytes) {
  SMI_REQUIRE(c && dst && src, "smi_memcpy_d2d: null argument");
  SMI_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

__global__ void k_fill_zero(uint32_t* p, size_t n_words) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n_words; i += stride) p[i] = 0u;
}

extern "C" int smi_memset_zero(smi_ctx* c, void* dptr, size_t bytes) {
  SMI_REQUIRE(c && dptr, "smi_memset_zero: null argument");
  SMI_REQUIRE(bytes % 4 == 0, "smi_memset_zero: byte count must be a multiple of 4");
  // a fill kernel, not hipMemsetAsync, so that callers may capture this into a graph
  size_t n = bytes / 4;
  if (n == 0) return 0;
  int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  k_fill_zero<<<blocks, 256, 0, c->stream>>>((uint32_t*)dptr, n);
  SMI_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ elementwise
// One generic grid-stride kernel: F maps (inputs) -> output, four reals per lane when
// every pointer is aligned to four elements.
constexpr int kEwBlock = 256;
static inline int ew_blocks(size_t work) {
  size_t b = (work + kEwBlock - 1) / kEwBlock;
  return (int)(b < 8192 ? (b ? b : 1) : 8192);
}
template <typename T>
static inline bool aligned4(const void* p) {
  return ((uintptr_t)p % (4 * sizeof(T))) == 0;
}

template <typename T, typename F>
__global__ void __launch_bounds__(kEwBlock) k_ew2(T* out, const T* a, const T* b, size_t n, F f) {
  size_t n4 = n / 4;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = i; k < n4; k += stride) {
    Vec4<T> va = load4(a + 4 * k), vb = load4(b + 4 * k), vo;
#pragma unroll
    for (int e = 0; e < 4; ++e) vo.v[e] = f(va.v[e], vb.v[e]);
    store4(out + 4 * k, vo);
  }
  for (size_t k = 4 * n4 + i; k < n; k += stride) out[k] = f(a[k], b[k]);
}
template <typename T, typename F>
__global__ void __launch_bounds__(kEwBlock) k_ew2_scalar(T* out, const T* a, const T* b, size_t n, F f) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = i; k < n; k += stride) out[k] = f(a[k], b[k]);
}

template <typename T, typename F>
static int launch_ew2(smi_ctx* c, T* out, const T* a, const T* b, size_t n, F f) {
  if (aligned4<T>(out) && aligned4<T>(a) && aligned4<T>(b)) {
    k_ew2<T, F><<<ew_blocks(n / 4 + 1), kEwBlock, 0, c->stream>>>(out, a, b, n, f);
  } else {
    k_ew2_scalar<T, F><<<ew_blocks(n), kEwBlock, 0, c->stream>>>(out, a, b, n, f);
  }
  SMI_LAUNCH_CHECK();
  return 0;
}

template <typename T>
struct OpSum {
  __device__ T operator()(T a, T b) const { return a + b; }
};
template <typename T>
struct OpCopy {
  __device__ T operator()(T a, T) const { return a; }
};
template <typename T>
struct OpSaxpby {
  T pa, pb;
  __device__ T operator()(T a, T b) const { return pa * a + pb * b; }
};
template <typename T>
struct OpAddConst {
  T v;
  __device__ T operator()(T a, T) const { return a + v; }
};
template <typename T>
struct OpSetConst {
  T v;
  __device__ T operator()(T, T) const { return v; }
};

extern "C" int smi_set_fixed_val(smi_ctx* c, void* field, int ncomp, const double* vals) {
  SMI_REQUIRE(c && field && vals && ncomp >= 1, "smi_set_fixed_val: bad argument");
  SMI_DISPATCH(c, {
    for (int k = 0; k < ncomp; ++k) {
      T* f = (T*)field + (size_t)k * c->vol;
      SMI_CHECK(launch_ew2<T>(c, f, f, f, c->vol, OpSetConst<T>{(T)vals[k]}));
    }
  });
  return 0;
}

extern "C" int smi_add_fixed_val(smi_ctx* c, void* sum_field, const void* field, int ncomp,
                                 const double* vals) {
  SMI_REQUIRE(c && sum_field && field && vals && ncomp >= 1, "smi_add_fixed_val: bad argument");
  SMI_DISPATCH(c, {
    for (int k = 0; k < ncomp; ++k) {
      T* o = (T*)sum_field + (size_t)k * c->vol;
      const T* f = (const T*)field + (size_t)k * c->vol;
      SMI_CHECK(launch_ew2<T>(c, o, f, f, c->vol, OpAddConst<T>{(T)vals[k]}));
    }
  });
  return 0;
}

extern "C" int smi_elementwise_sum(smi_ctx* c, void* sum_field, const void* f1, const void* f2, int ncomp) {
  SMI_REQUIRE(c && sum_field && f1 && f2 && ncomp >= 1, "smi_elementwise_sum: bad argument");
  SmiProfScope prof(c, PROF_EW_SUM, 3.0 * ncomp * smi_S(c));
  SMI_DISPATCH(c, {
    SMI_CHECK(launch_ew2<T>(c, (T*)sum_field, (const T*)f1, (const T*)f2, c->vol * ncomp, OpSum<T>{}));
  });
  return 0;
}

extern "C" int smi_elementwise_copy(smi_ctx* c, void* dst, const void* src, int ncomp) {
  SMI_REQUIRE(c && dst && src && ncomp >= 1, "smi_elementwise_copy: bad argument");
  SMI_DISPATCH(c, {
    SMI_CHECK(launch_ew2<T>(c, (T*)dst, (const T*)src, (const T*)src, c->vol * ncomp, OpCopy<T>{}));
  });
  return 0;
}

extern "C" int smi_elementwise_saxpby(smi_ctx* c, void* sum_field, const void* f1, double a,
                                      const void* f2, double b, int ncomp) {
  SMI_REQUIRE(c && sum_field && f1 && f2 && ncomp >= 1, "smi_elementwise_saxpby: bad argument");
  SMI_DISPATCH(c, {
    SMI_CHECK(launch_ew2<T>(c, (T*)sum_field, (const T*)f1, (const T*)f2, c->vol * ncomp,
                            OpSaxpby<T>{(T)a, (T)b}));
  });
  return 0;
}

// cross product: six inputs, three outputs, one pass.
template <typename T, int V>
__global__ void __launch_bounds__(kEwBlock) k_cross(T* __restrict__ r, const T* __restrict__ a,
                                                    const T* __restrict__ b, size_t vol) {
  size_t nv = vol / V;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = i; k < nv; k += stride) {
    T a0[V], a1[V], a2[V], b0[V], b1[V], b2[V], r0[V], r1[V], r2[V];
    if constexpr (V == 4) {
      *(Vec4<T>*)a0 = load4(a + 4 * k);
      *(Vec4<T>*)a1 = load4(a + vol + 4 * k);
      *(Vec4<T>*)a2 = load4(a + 2 * vol + 4 * k);
      *(Vec4<T>*)b0 = load4(b + 4 * k);
      *(Vec4<T>*)b1 = load4(b + vol + 4 * k);
      *(Vec4<T>*)b2 = load4(b + 2 * vol + 4 * k);
    } else {
      a0[0] = a[k]; a1[0] = a[vol + k]; a2[0] = a[2 * vol + k];
      b0[0] = b[k]; b1[0] = b[vol + k]; b2[0] = b[2 * vol + k];
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
      r0[e] = a1[e] * b2[e] - a2[e] * b1[e];
      r1[e] = a2[e] * b0[e] - a0[e] * b2[e];
      r2[e] = a0[e] * b1[e] - a1[e] * b0[e];
    }
    if constexpr (V == 4) {
      store4(r + 4 * k, *(Vec4<T>*)r0);
      store4(r + vol + 4 * k, *(Vec4<T>*)r1);
      store4(r + 2 * vol + 4 * k, *(Vec4<T>*)r2);
    } else {
      r[k] = r0[0]; r[vol + k] = r1[0]; r[2 * vol + k] = r2[0];
    }
  }
}

extern "C" int smi_cross_product(smi_ctx* c, void* result, const void* f1, const void* f2) {
  SMI_REQUIRE(c && result && f1 && f2, "smi_cross_product: bad argument");
  SmiProfScope prof(c, PROF_CROSS, 9.0 * smi_S(c));
  SMI_DISPATCH(c, {
    bool v4 = c->vol % 4 == 0 && aligned4<T>(result) && aligned4<T>(f1) && aligned4<T>(f2);
    if (v4)
      k_cross<T, 4><<<ew_blocks(c->vol / 4), kEwBlock, 0, c->stream>>>((T*)result, (const T*)f1,
                                                                       (const T*)f2, c->vol);
    else
      k_cross<T, 1><<<ew_blocks(c->vol), kEwBlock, 0, c->stream>>>((T*)result, (const T*)f1,
                                                                   (const T*)f2, c->vol);
    SMI_LAUNCH_CHECK();
  });
  return 0;
}

// ------------------------------------------------------------------ reductions
// Interior-only reductions.  Stage 1: kRedBlocks blocks each reduce a strided set of
// interior rows; stage 2: one block reduces the partials.  Accumulation in double.
enum { RED_MAX_ABS_SUM3 = 0, RED_MAX = 1, RED_SUM_SQ = 2 };

__device__ __forceinline__ double wave_reduce(double v, bool is_max) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    double o = __shfl_down(v, off, 64);
    v = is_max ? fmax(v, o) : v + o;
  }
  return v;
}

__device__ __forceinline__ double block_reduce(double v, bool is_max, double* sh) {
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  v = wave_reduce(v, is_max);
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double r = is_max ? -INFINITY : 0.0;
  if (threadIdx.x < (blockDim.x >> 6)) r = sh[threadIdx.x];
  if (w == 0) r = wave_reduce(r, is_max);
  return r;  // valid in thread 0
}

// One pass over the interior of a field (or of a vector field's three components at once for
// the CFL quantity).  A workgroup takes kRedRows rows per trip, so that each lane has up to a
// dozen independent loads in flight; maxima are taken in the field's own type (exact), squares
// are summed in double.
constexpr int kRedRows = 4;

template <typename T, int MODE>
__global__ void __launch_bounds__(256) k_reduce_rows(const T* __restrict__ f, int ncomp, int nz, int ny,
                                                     int nx, int g, int px, size_t plane, size_t vol,
                                                     double* __restrict__ partials) {
  __shared__ double sh[4];
  constexpr bool is_max = MODE != RED_SUM_SQ;
  T amax = -INFINITY;
  double asum = 0.0;
  const size_t rows = (size_t)nz * ny * (MODE == RED_MAX_ABS_SUM3 ? 1 : ncomp);
  for (size_t row0 = (size_t)blockIdx.x * kRedRows; row0 < rows; row0 += (size_t)gridDim.x * kRedRows) {
    const T* base[kRedRows];
#pragma unroll
    for (int j = 0; j < kRedRows; ++j) {
      // rows past the end repeat the last one: harmless for a maximum, skipped for the sum
      size_t r = row0 + j < rows ? row0 + j : rows - 1;
      int comp = 0;
      if (MODE != RED_MAX_ABS_SUM3) {
        comp = (int)(r / ((size_t)nz * ny));
        r -= (size_t)comp * nz * ny;
      }
      const int z = (int)(r / ny) + g, y = (int)(r % ny) + g;
      base[j] = f + (size_t)comp * vol + (size_t)z * plane + (size_t)y * px + g;
    }
    for (int x = threadIdx.x; x < nx; x += blockDim.x) {
      if (MODE == RED_MAX_ABS_SUM3) {
        T a[kRedRows], b[kRedRows], c[kRedRows];
#pragma unroll
        for (int j = 0; j < kRedRows; ++j) {
          a[j] = base[j][x];
          b[j] = base[j][vol + x];
          c[j] = base[j][2 * vol + x];
        }
#pragma unroll
        for (int j = 0; j < kRedRows; ++j) amax = fmax(amax, fabs(a[j]) + fabs(b[j]) + fabs(c[j]));
      } else {
        T a[kRedRows];
#pragma unroll
        for (int j = 0; j < kRedRows; ++j) a[j] = base[j][x];
#pragma unroll
        for (int j = 0; j < kRedRows; ++j) {
          if (MODE == RED_MAX) {
            amax = fmax(amax, a[j]);
          } else if (row0 + j < rows) {
            asum += (double)a[j] * (double)a[j];
          }
        }
      }
    }
  }
  double r = block_reduce(is_max ? (double)amax : asum, is_max, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

template <bool IS_MAX>
__global__ void __launch_bounds__(256) k_reduce_final(const double* __restrict__ partials, int n,
                                                      double* __restrict__ out) {
  __shared__ double sh[4];
  double acc = IS_MAX ? -INFINITY : 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc = IS_MAX ? fmax(acc, partials[i]) : acc + partials[i];
  double r = block_reduce(acc, IS_MAX, sh);
  if (threadIdx.x == 0) out[0] = r;
}

template <int MODE>
static int run_reduce(smi_ctx* c, const void* field, int ncomp, double* out) {
  SMI_REQUIRE(c && field && out, "smi_reduce: bad argument");
  double* partials = (double*)c->red_partials;
  const size_t rows = (size_t)c->nz * c->ny * (MODE == RED_MAX_ABS_SUM3 ? 1 : ncomp);
  const size_t trips = (rows + kRedRows - 1) / kRedRows;
  int blocks = (int)(trips < (size_t)kRedBlocks ? trips : kRedBlocks);
  SMI_DISPATCH(c, {
    k_reduce_rows<T, MODE><<<blocks, 256, 0, c->stream>>>((const T*)field, ncomp, c->nz, c->ny, c->nx, c->g,
                                                          c->px, c->plane, c->vol, partials);
  });
  SMI_LAUNCH_CHECK();
  k_reduce_final<MODE != RED_SUM_SQ><<<1, 256, 0, c->stream>>>(partials, blocks, partials + kRedBlocks);
  SMI_LAUNCH_CHECK();
  SMI_CHECK_HIP(hipMemcpyAsync(c->red_host, partials + kRedBlocks, sizeof(double), hipMemcpyDeviceToHost,
                               c->stream));
  SMI_CHECK_HIP(hipStreamSynchronize(c->stream));
  *out = c->red_host[0];
  return 0;
}

extern "C" int smi_reduce_max_abs_sum(smi_ctx* c, const void* vector_field, double* out) {
  return run_reduce<RED_MAX_ABS_SUM3>(c, vector_field, 3, out);
}
// The same quantity for the velocity the last smi_flow_step_ns3d wrote, which its curl kernel gathered on
// the way: no pass over the field.  Fails when no such value is at hand (no fused step yet, or the step ran
// its scalar-path curl); the caller then reduces the field.  It does not know whether the velocity field
// has been written since: that is the caller's book-keeping.
extern "C" int smi_step_velocity_max_abs_sum(smi_ctx* c, const void* velocity, double* out) {
  SMI_REQUIRE(c && velocity && out, "smi_step_velocity_max_abs_sum: null argument");
  SMI_REQUIRE(c->umax_valid, "smi_step_velocity_max_abs_sum: no fused step has recorded the maximum");
  SMI_REQUIRE(c->umax_field == velocity,
              "smi_step_velocity_max_abs_sum: the recorded maximum belongs to another velocity array of this context");
  // into the context's pinned host words (behind those of the reductions): a copy into pageable memory is staged
  // and costs the loop that asks for dt every step some ten microseconds more
  const unsigned long long* bits = reinterpret_cast<const unsigned long long*>(c->red_host + 8);
  const size_t es = smi_elem_size(c);
  SMI_CHECK_HIP(hipMemcpyAsync(c->red_host + 8, c->umax_slots, es * kUmaxSlots, hipMemcpyDeviceToHost, c->stream));
  SMI_CHECK_HIP(hipStreamSynchronize(c->stream));
  double m = c->umax_band;
  for (int i = 0; i < kUmaxSlots; ++i) {
    double v;
    if (c->dtype == SMI_F32) {
      float f;
      memcpy(&f, (const char*)bits + 4 * i, 4);
      v = f;
    } else {
      memcpy(&v, (const char*)bits + 8 * i, 8);
    }
    if (v > m) m = v;
  }
  *out = m;
  return 0;
}
extern "C" int smi_reduce_max(smi_ctx* c, const void* field, int ncomp, double* out) {
  SMI_REQUIRE(ncomp >= 1, "smi_reduce_max: ncomp must be >= 1");
  return run_reduce<RED_MAX>(c, field, ncomp, out);
}
extern "C" int smi_reduce_sum_sq(smi_ctx* c, const void* field, int ncomp, double* out) {
  SMI_REQUIRE(ncomp >= 1, "smi_reduce_sum_sq: ncomp must be >= 1");
  return run_reduce<RED_SUM_SQ>(c, field, ncomp, out);
}

// ------------------------------------------------------------------ profiling

static const char* kSlotNames[PROF_NUM_SLOTS] = {
    "cross_product", "curl_update", "diffusion_flux", "elementwise_sum", "laplacian_filter", "penalise",
    "curl", "add_set_fixed", "poisson_pad", "poisson_rocfft_fwd", "poisson_multiply", "poisson_rocfft_inv",
    "poisson_extract", "fft_x_fwd", "fft_y_fwd", "fft_z_conv", "fft_y_inv", "fft_x_inv", "transpose", "halo",
    "ib_forcing", "ib_spread", "fused_advect", "fused_diffuse", "advection_flux", "fused_advect_diffuse", "flow_step"};

const char* smi_prof_slot_name(int slot) { return slot >= 0 && slot < PROF_NUM_SLOTS ? kSlotNames[slot] : "?"; }

static hipEvent_t prof_event(SmiProf* p) {
  if (!p->pool.empty()) {
    hipEvent_t e = p->pool.back();
    p->pool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

SmiProfScope::SmiProfScope(smi_ctx* ctx, int slot, double alg_bytes) : c(ctx), active(false) {
  if (!ctx->prof || !ctx->prof->on) return;
  rec.slot = slot;
  rec.bytes = alg_bytes;
  rec.a = prof_event(ctx->prof);
  rec.b = prof_event(ctx->prof);
  if (!rec.a || !rec.b) return;
  active = hipEventRecord(rec.a, ctx->stream) == hipSuccess;
}

SmiProfScope::~SmiProfScope() {
  if (!active) return;
  if (hipEventRecord(rec.b, c->stream) == hipSuccess) c->prof->live.push_back(rec);
}

static int prof_resolve(smi_ctx* c) {
  SmiProf* p = c->prof;
  if (!p || p->live.empty()) return 0;
  SMI_CHECK_HIP(hipStreamSynchronize(c->stream));
  for (auto& r : p->live) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      p->launches[r.slot] += 1;
      p->total_ms[r.slot] += ms;
      p->bytes[r.slot] += r.bytes;
      if (p->samples[r.slot].size() < kProfMaxSamples) p->samples[r.slot].push_back(ms);
    }
    p->pool.push_back(r.a);
    p->pool.push_back(r.b);
  }
  p->live.clear();
  return 0;
}

extern "C" int smi_prof_enable(smi_ctx* c, int on) {
  SMI_REQUIRE(c, "smi_prof_enable: null context");
  if (!c->prof) c->prof = new SmiProf();
  SMI_CHECK(prof_resolve(c));
  c->prof->on = on != 0;
  return 0;
}

extern "C" int smi_prof_reset(smi_ctx* c) {
  SMI_REQUIRE(c, "smi_prof_reset: null context");
  if (!c->prof) return 0;
  SMI_CHECK(prof_resolve(c));
  for (int i = 0; i < PROF_NUM_SLOTS; ++i) {
    c->prof->launches[i] = 0;
    c->prof->total_ms[i] = 0;
    c->prof->bytes[i] = 0;
    c->prof->samples[i].clear();
  }
  return 0;
}

extern "C" int smi_prof_num_slots(void) { return PROF_NUM_SLOTS; }

extern "C" int smi_prof_get_samples(smi_ctx* c, int slot, double* out_ms, int capacity, int* count) {
  SMI_REQUIRE(c && slot >= 0 && slot < PROF_NUM_SLOTS && count && capacity >= 0 && (out_ms || capacity == 0),
              "smi_prof_get_samples: bad argument");
  *count = 0;
  if (!c->prof) return 0;
  SMI_CHECK(prof_resolve(c));
  const std::vector<float>& s = c->prof->samples[slot];
  const int n = (int)s.size() < capacity ? (int)s.size() : capacity;
  for (int i = 0; i < n; ++i) out_ms[i] = s[i];
  *count = n;
  return 0;
}

extern "C" int smi_prof_get(smi_ctx* c, int slot, char* name64, int64_t* launches, double* total_ms,
                            double* alg_bytes) {
  SMI_REQUIRE(c && slot >= 0 && slot < PROF_NUM_SLOTS, "smi_prof_get: bad argument");
  if (name64) {
    strncpy(name64, kSlotNames[slot], 63);
    name64[63] = 0;
  }
  if (c->prof) SMI_CHECK(prof_resolve(c));
  if (launches) *launches = c->prof ? c->prof->launches[slot] : 0;
  if (total_ms) *total_ms = c->prof ? c->prof->total_ms[slot] : 0.0;
  if (alg_bytes) *alg_bytes = c->prof ? c->prof->bytes[slot] : 0.0;
  return 0;
}
