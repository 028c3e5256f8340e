// Inter-GPU plumbing of libsophtmi: RCCL communicator over the z-slab ranks, halo
// exchange, ghost sum.  A z-slab's ghost planes are whole padded planes, contiguous in
// memory, so they travel as they lie with ncclSend / ncclRecv: no pack or unpack pass.
#include "smi_common.h"

extern "C" int smi_comm_unique_id(void* out128) {
  SMI_REQUIRE(out128, "smi_comm_unique_id: null output");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
  ncclUniqueId id;
  SMI_CHECK_NCCL(ncclGetUniqueId(&id));
  memcpy(out128, &id, sizeof(id));
  return 0;
}

extern "C" int smi_comm_init(smi_ctx* c, const void* unique_id128) {
  SMI_REQUIRE(c && unique_id128, "smi_comm_init: null argument");
  SMI_REQUIRE(!c->comm_ready, "smi_comm_init: communicator already initialised");
  SMI_CHECK_HIP(hipSetDevice(c->device));
  ncclUniqueId id;
  memcpy(&id, unique_id128, sizeof(id));
  SMI_CHECK_NCCL(ncclCommInitRank(&c->comm, c->nranks, id, c->rank));
  c->comm_ready = true;
  return 0;
}

extern "C" int smi_set_exchange_callback(smi_ctx* c, smi_exchange_cb cb, void* user) {
  SMI_REQUIRE(c, "smi_set_exchange_callback: null context");
  c->exchange_cb = cb;
  c->exchange_user = user;
  return 0;
}

extern "C" int smi_allreduce_f64(smi_ctx* c, double* host_inout, int n, int op) {
  SMI_REQUIRE(c && host_inout && n >= 1 && n <= 8, "smi_allreduce_f64: bad argument (n must be in [1, 8])");
  SMI_REQUIRE(op >= 0 && op <= 2, "smi_allreduce_f64: bad op %d", op);
  if (c->nranks == 1) return 0;
  SMI_REQUIRE(c->comm_ready, "smi_allreduce_f64 needs smi_comm_init");
  ncclRedOp_t rop = op == 0 ? ncclSum : (op == 1 ? ncclMin : ncclMax);
  SMI_CHECK_HIP(hipMemcpyAsync(c->fac_dev, host_inout, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  SMI_CHECK_NCCL(ncclAllReduce(c->fac_dev, c->fac_dev, n, ncclDouble, rop, c->comm, c->stream));
  SMI_CHECK_HIP(hipMemcpyAsync(c->red_host, c->fac_dev, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  SMI_CHECK_HIP(hipStreamSynchronize(c->stream));
  for (int i = 0; i < n; ++i) host_inout[i] = c->red_host[i];
  return 0;
}

static inline ncclDataType_t nccl_type(const smi_ctx* c) { return c->dtype == SMI_F32 ? ncclFloat : ncclDouble; }

// ---------------------------------------------------------------- y faces of a pencil
// d rows of every plane (z ghost planes included) of every component <-> a contiguous buffer [comp][z][row][x]
template <typename T, bool UNPACK, bool ADD>
__global__ void __launch_bounds__(256) k_rows(T* __restrict__ field, T* __restrict__ buf, int pz, int px, int d, int y0,
                                              long long plane, long long vol, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % px);
  long long r = i / px;
  const int row = (int)(r % d);
  r /= d;
  const int z = (int)(r % pz), comp = (int)(r / pz);
  T* cell = field + comp * vol + z * plane + (long long)(y0 + row) * px + x;
  if (!UNPACK)
    buf[i] = *cell;
  else if (ADD)
    *cell += buf[i];
  else
    *cell = buf[i];
}

template <bool UNPACK, bool ADD>
static int rows_copy(smi_ctx* c, void* field, void* buf, int ncomp, int d, int y0, hipStream_t stream) {
  const long long n = (long long)ncomp * c->pz * d * c->px;
  SMI_DISPATCH(c, {
    k_rows<T, UNPACK, ADD><<<(int)((n + 255) / 256), 256, 0, stream>>>((T*)field, (T*)buf, c->pz, c->px, d, y0,
                                                                      (long long)c->plane, (long long)c->vol, n);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

// d columns of every row of every plane (ghost rows and planes included) of every component <-> a contiguous buffer
// [comp][z][row][column]
template <typename T, bool UNPACK, bool ADD>
__global__ void __launch_bounds__(256) k_cols(T* __restrict__ field, T* __restrict__ buf, int pz, int py, int px, int d,
                                              int x0, long long plane, long long vol, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int col = (int)(i % d);
  long long r = i / d;
  const int row = (int)(r % py);
  r /= py;
  const int z = (int)(r % pz), comp = (int)(r / pz);
  T* cell = field + comp * vol + z * plane + (long long)row * px + x0 + col;
  if (!UNPACK)
    buf[i] = *cell;
  else if (ADD)
    *cell += buf[i];
  else
    *cell = buf[i];
}

template <bool UNPACK, bool ADD>
static int cols_copy(smi_ctx* c, void* field, void* buf, int ncomp, int d, int x0, hipStream_t stream) {
  const long long n = (long long)ncomp * c->pz * c->py * d;
  SMI_DISPATCH(c, {
    k_cols<T, UNPACK, ADD><<<(int)((n + 255) / 256), 256, 0, stream>>>((T*)field, (T*)buf, c->pz, c->py, c->px, d, x0,
                                                                      (long long)c->plane, (long long)c->vol, n);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

// Exchange with the two neighbours along y (AXIS 1: rank -/+ nranks_x) or x (AXIS 2: rank -/+ 1).
// Halo (sum = false): my first / last d interior rows (columns) -> the neighbour's ghost rows (columns) next to its
// interior (MPIGhostCommunicator3D, y and x faces).  Ghost sum (sum = true, d = g): my ghost rows (columns) are added
// to the neighbour's edge interior rows (columns) (MPIGhostSumCommunicator3D.ghost_sum, the y and x steps).
template <int AXIS>
static int face_exchange(smi_ctx* c, void* field, int ncomp, int d, bool sum, hipStream_t stream) {
  const int P = AXIS == 1 ? c->nranks_y : c->nranks_x, me = AXIS == 1 ? c->rank_y : c->rank_x;
  if (P == 1) return 0;
  SMI_REQUIRE(ncomp >= 1 && ncomp <= 3 && d >= 1 && d <= c->g, "face exchange: bad argument");
  const int g = c->g, p = AXIS == 1 ? c->py : c->px, other = AXIS == 1 ? c->px : c->py;
  const bool lo = me > 0, hi = me < P - 1;
  const size_t bytes = (size_t)ncomp * c->pz * d * other * smi_elem_size(c);
  SMI_REQUIRE(bytes <= c->ybuf_cap, "face exchange: staging buffer too small");
  const int src_lo = sum ? 0 : g, src_hi = sum ? p - g : p - g - d;
  const int dst_lo = sum ? g : g - d, dst_hi = sum ? p - 2 * g : p - g;
  auto pack = [&](void* buf, int at) {
    return AXIS == 1 ? rows_copy<false, false>(c, field, buf, ncomp, d, at, stream)
                     : cols_copy<false, false>(c, field, buf, ncomp, d, at, stream);
  };
  auto unpack = [&](void* buf, int at) {
    if (sum)
      return AXIS == 1 ? rows_copy<true, true>(c, field, buf, ncomp, d, at, stream)
                       : cols_copy<true, true>(c, field, buf, ncomp, d, at, stream);
    return AXIS == 1 ? rows_copy<true, false>(c, field, buf, ncomp, d, at, stream)
                     : cols_copy<true, false>(c, field, buf, ncomp, d, at, stream);
  };
  if (lo) SMI_CHECK(pack(c->ybuf[0], src_lo));
  if (hi) SMI_CHECK(pack(c->ybuf[1], src_hi));
  const int step = AXIS == 1 ? c->nranks_x : 1;
  const int peer_lo = lo ? c->rank - step : -1, peer_hi = hi ? c->rank + step : -1;
  if (c->sendrecv_cb) {
    SMI_CHECK_HIP(hipStreamSynchronize(stream));
    int rc = c->sendrecv_cb(c->exchange_user, peer_lo, peer_hi, c->ybuf[0], c->ybuf[1], c->ybuf[2], c->ybuf[3], bytes);
    SMI_REQUIRE(rc == 0, "face exchange callback failed (%d)", rc);
  } else {
    SMI_REQUIRE(c->comm_ready, "face exchange across ranks needs smi_comm_init or a send-receive callback");
    SMI_CHECK_NCCL(ncclGroupStart());
    if (lo) {
      SMI_CHECK_NCCL(ncclSend(c->ybuf[0], bytes, ncclChar, peer_lo, c->comm, stream));
      SMI_CHECK_NCCL(ncclRecv(c->ybuf[2], bytes, ncclChar, peer_lo, c->comm, stream));
    }
    if (hi) {
      SMI_CHECK_NCCL(ncclSend(c->ybuf[1], bytes, ncclChar, peer_hi, c->comm, stream));
      SMI_CHECK_NCCL(ncclRecv(c->ybuf[3], bytes, ncclChar, peer_hi, c->comm, stream));
    }
    SMI_CHECK_NCCL(ncclGroupEnd());
  }
  if (lo) SMI_CHECK(unpack(c->ybuf[2], dst_lo));
  if (hi) SMI_CHECK(unpack(c->ybuf[3], dst_hi));
  return 0;
}

// y faces, then x faces with the fresh ghost rows in them (the z faces went first): edges and vertices between
// diagonal neighbours are filled without messages of their own, and the ghost sum cascades z -> y -> x as the
// reference's does.
int smi_y_face_exchange(smi_ctx* c, void* field, int ncomp, int d, bool sum, hipStream_t stream) {
  SMI_CHECK(face_exchange<1>(c, field, ncomp, d, sum, stream));
  return face_exchange<2>(c, field, ncomp, d, sum, stream);
}

extern "C" int smi_set_sendrecv_callback(smi_ctx* c, smi_sendrecv_cb cb) {
  SMI_REQUIRE(c, "smi_set_sendrecv_callback: null context");
  c->sendrecv_cb = cb;
  return 0;
}

extern "C" int smi_set_group_alltoall_callback(smi_ctx* c, smi_group_alltoall_cb cb) {
  SMI_REQUIRE(c, "smi_set_group_alltoall_callback: null context");
  c->group_alltoall_cb = cb;
  return 0;
}

// MPIGhostCommunicator3D z faces: my first / last d interior planes -> the neighbour's
// high / low ghost planes next to its interior.  d = g (what the reference exchanges) unless the
// fused step has lowered halo_depth to what its stencils read.  Two fields may travel in one
// group (one launch): the velocity and the vorticity ahead of the fused advection.
int smi_halo_exchange2_impl(smi_ctx* c, void* field_a, int ncomp_a, void* field_b, int ncomp_b) {
  return smi_halo_exchange2_on(c, field_a, ncomp_a, field_b, ncomp_b, c->stream);
}

int smi_halo_exchange2_on(smi_ctx* c, void* field_a, int ncomp_a, void* field_b, int ncomp_b, hipStream_t stream) {
  SMI_REQUIRE(!c->blk_on,
              "this operator needs slabs or pencils over z and y of an unbounded domain; a context with a general "
              "block topology (smi_ctx_set_block) serves smi_halo_exchange_general only");
  if (c->nranks == 1) return 0;
  const int d = (c->halo_depth > 0 && c->halo_depth < c->g) ? c->halo_depth : c->g;
  void* fields[2] = {field_a, field_b};
  const int ncomps[2] = {ncomp_a, field_b ? ncomp_b : 0};
  if (c->exchange_cb) {
    for (int f = 0; f < 2; ++f) {
      if (!ncomps[f]) continue;
      int rc = c->exchange_cb(c->exchange_user, fields[f], ncomps[f], d == c->g ? 0 : 16 * d);
      SMI_REQUIRE(rc == 0, "halo exchange callback failed (%d)", rc);
      SMI_CHECK(smi_y_face_exchange(c, fields[f], ncomps[f], d, false, stream));
    }
    return 0;
  }
  SMI_REQUIRE(c->comm_ready, "halo exchange across ranks needs smi_comm_init or an exchange callback");
  const size_t es = smi_elem_size(c);
  const size_t n = (size_t)d * c->plane;  // elements per face per component
  const int zstep = c->nranks_y * c->nranks_x;  // the z neighbours of a pencil or block
  const int zprev = c->rank - zstep, znext = c->rank + zstep;
  if (c->nranks_z > 1) {
    SMI_CHECK_NCCL(ncclGroupStart());
    for (int f = 0; f < 2; ++f)
      for (int k = 0; k < ncomps[f]; ++k) {
        char* p = (char*)fields[f] + (size_t)k * c->vol * es;
        if (c->rank_z > 0) {
          SMI_CHECK_NCCL(ncclSend(p + (size_t)c->g * c->plane * es, n, nccl_type(c), zprev, c->comm, stream));
          SMI_CHECK_NCCL(ncclRecv(p + (size_t)(c->g - d) * c->plane * es, n, nccl_type(c), zprev, c->comm, stream));
        }
        if (c->rank_z < c->nranks_z - 1) {
          SMI_CHECK_NCCL(ncclSend(p + (size_t)(c->pz - c->g - d) * c->plane * es, n, nccl_type(c), znext, c->comm,
                                  stream));
          SMI_CHECK_NCCL(ncclRecv(p + (size_t)(c->pz - c->g) * c->plane * es, n, nccl_type(c), znext, c->comm, stream));
        }
      }
    SMI_CHECK_NCCL(ncclGroupEnd());
  }
  // y faces after the z faces: their rows include the z ghost planes just received, which fills the edges
  for (int f = 0; f < 2; ++f)
    if (ncomps[f]) SMI_CHECK(smi_y_face_exchange(c, fields[f], ncomps[f], d, false, stream));
  return 0;
}

int smi_halo_exchange_impl(smi_ctx* c, void* field, int ncomp) {
  return smi_halo_exchange2_impl(c, field, ncomp, nullptr, 0);
}

// All-to-all of equal contiguous blocks: block r of `send` goes to rank r, block s of `recv`
// comes from rank s.  Replaces the Alltoallw transposes inside mpi4py-fft's PFFT
// (fft_mpi_3d.py:27-48).  One RCCL group of point-to-point pairs: on xGMI every pair has its
// own link, so all seven peers move at once.
int smi_alltoall_impl(smi_ctx* c, const void* send, void* recv, size_t bytes_per_peer, hipStream_t stream) {
  if (c->nranks == 1) {
    if (send != recv) SMI_CHECK_HIP(hipMemcpyAsync(recv, send, bytes_per_peer, hipMemcpyDeviceToDevice, stream));
    return 0;
  }
  if (c->alltoall_cb) {
    int rc = c->alltoall_cb(c->exchange_user, send, recv, bytes_per_peer);
    SMI_REQUIRE(rc == 0, "all-to-all callback failed (%d)", rc);
    return 0;
  }
  SMI_REQUIRE(c->comm_ready, "all-to-all across ranks needs smi_comm_init or a transport callback");
  // own block: a plain device copy, stream-ordered with the group
  SMI_CHECK_HIP(hipMemcpyAsync((char*)recv + (size_t)c->rank * bytes_per_peer,
                               (const char*)send + (size_t)c->rank * bytes_per_peer, bytes_per_peer,
                               hipMemcpyDeviceToDevice, stream));
  SMI_CHECK_NCCL(ncclGroupStart());
  for (int r = 0; r < c->nranks; ++r) {
    if (r == c->rank) continue;
    SMI_CHECK_NCCL(ncclSend((const char*)send + (size_t)r * bytes_per_peer, bytes_per_peer, ncclChar, r, c->comm,
                            stream));
    SMI_CHECK_NCCL(ncclRecv((char*)recv + (size_t)r * bytes_per_peer, bytes_per_peer, ncclChar, r, c->comm, stream));
  }
  SMI_CHECK_NCCL(ncclGroupEnd());
  return 0;
}

// The same exchange for one slice of every block: block r starts at r * peer_stride and only its `bytes` bytes at
// `offset` travel (one z chunk of every rank block of the spectral exchange layout).  `stage_send` / `stage_recv`
// (nranks * bytes each) are only used by the host-staged callback transport, which wants contiguous blocks.
int smi_alltoall_slice_impl(smi_ctx* c, const void* send, void* recv, size_t peer_stride, size_t offset, size_t bytes,
                            void* stage_send, void* stage_recv, hipStream_t stream) {
  const char* s0 = (const char*)send + offset;
  char* r0 = (char*)recv + offset;
  if (c->nranks == 1) {
    if (send != recv) SMI_CHECK_HIP(hipMemcpyAsync(r0, s0, bytes, hipMemcpyDeviceToDevice, stream));
    return 0;
  }
  if (c->alltoall_cb) {
    SMI_REQUIRE(stage_send && stage_recv, "sliced all-to-all over a callback transport needs staging buffers");
    SMI_CHECK_HIP(hipMemcpy2DAsync(stage_send, bytes, s0, peer_stride, bytes, c->nranks, hipMemcpyDeviceToDevice, stream));
    int rc = c->alltoall_cb(c->exchange_user, stage_send, stage_recv, bytes);
    SMI_REQUIRE(rc == 0, "all-to-all callback failed (%d)", rc);
    SMI_CHECK_HIP(hipMemcpy2DAsync(r0, peer_stride, stage_recv, bytes, bytes, c->nranks, hipMemcpyDeviceToDevice, stream));
    return 0;
  }
  SMI_REQUIRE(c->comm_ready, "all-to-all across ranks needs smi_comm_init or a transport callback");
  SMI_CHECK_HIP(hipMemcpyAsync(r0 + (size_t)c->rank * peer_stride, s0 + (size_t)c->rank * peer_stride, bytes,
                               hipMemcpyDeviceToDevice, stream));
  SMI_CHECK_NCCL(ncclGroupStart());
  for (int r = 0; r < c->nranks; ++r) {
    if (r == c->rank) continue;
    SMI_CHECK_NCCL(ncclSend(s0 + (size_t)r * peer_stride, bytes, ncclChar, r, c->comm, stream));
    SMI_CHECK_NCCL(ncclRecv(r0 + (size_t)r * peer_stride, bytes, ncclChar, r, c->comm, stream));
  }
  SMI_CHECK_NCCL(ncclGroupEnd());
  return 0;
}

extern "C" int smi_set_alltoall_callback(smi_ctx* c, smi_alltoall_cb cb) {
  SMI_REQUIRE(c, "smi_set_alltoall_callback: null context");
  c->alltoall_cb = cb;
  return 0;
}

// ---------------------------------------------------------------- general block topology
// MPIConstruct3D / 2D with any rank_distribution and periodic_domain (utils/mpi_utils_3d.py:7-83, mpi_utils_2d.py:9-84)
// and the ghost exchange of MPIGhostCommunicator3D / 2D on it (utils/mpi_utils_3d.py:86-1142: faces, and with
// full_exchange edges and vertices).  The operators of this library keep x whole and the domain unbounded; a
// context turned into a general block serves the exchange only.
extern "C" int smi_ctx_set_block(smi_ctx* c, const int* coords_zyx, const int* topo_zyx, const int* periodic_zyx) {
  SMI_REQUIRE(c && coords_zyx && topo_zyx && periodic_zyx, "smi_ctx_set_block: null argument");
  long long n = 1, r = 0;
  for (int a = 0; a < 3; ++a) {
    SMI_REQUIRE(topo_zyx[a] >= 1 && coords_zyx[a] >= 0 && coords_zyx[a] < topo_zyx[a],
                "smi_ctx_set_block: position %d of %d along axis %d", coords_zyx[a], topo_zyx[a], a);
    n *= topo_zyx[a];
    r = r * topo_zyx[a] + coords_zyx[a];
  }
  SMI_REQUIRE(n == c->nranks && r == c->rank, "smi_ctx_set_block: block (%d, %d, %d) of a %d x %d x %d grid is not rank %d of %d",
              coords_zyx[0], coords_zyx[1], coords_zyx[2], topo_zyx[0], topo_zyx[1], topo_zyx[2], c->rank, c->nranks);
  SMI_REQUIRE(c->nz > 1 || topo_zyx[0] == 1, "smi_ctx_set_block: a 2D context (nz = 1) has no ranks along z");
  for (int a = 0; a < 3; ++a) {
    c->blk_coords[a] = coords_zyx[a];
    c->blk_topo[a] = topo_zyx[a];
    c->blk_periodic[a] = periodic_zyx[a] != 0;
  }
  c->blk_on = 1;
  return 0;
}

namespace {
// a box of `field` (or of a packed buffer) to a box of the same extents elsewhere: strides in elements
struct BoxSide {
  long long comp, z, y, off;
};
template <typename T>
__global__ void __launch_bounds__(256) k_box_copy(T* __restrict__ dst, const T* __restrict__ src, int bz, int by, int bx,
                                                 BoxSide D, BoxSide S, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % bx);
  long long r = i / bx;
  const int y = (int)(r % by);
  r /= by;
  const int z = (int)(r % bz), k = (int)(r / bz);
  dst[D.off + k * D.comp + z * D.z + y * D.y + x] = src[S.off + k * S.comp + z * S.z + y * S.y + x];
}
}  // namespace

// Ghost cells of `field` from the neighbouring blocks, axis by axis (z, y, x).  full_exchange: the faces of an axis
// span the whole padded extent of the axes already done, fresh ghost cells included, so edges and vertices arrive
// without diagonal messages; otherwise faces span interiors only.  A periodic axis with a single rank wraps onto
// the block itself.
extern "C" int smi_halo_exchange_general(smi_ctx* c, void* field, int ncomp, int full_exchange) {
  SMI_REQUIRE(c && field && ncomp >= 1 && ncomp <= 3, "smi_halo_exchange_general: bad argument");
  SMI_REQUIRE(c->blk_on, "smi_halo_exchange_general: the context has no block topology (smi_ctx_set_block)");
  const bool two_d = c->nz == 1;
  const int g = c->g;
  const int p[3] = {two_d ? 1 : c->pz, c->py, c->px};
  const int gh[3] = {two_d ? 0 : g, g, g};
  const long long comp_stride = two_d ? (long long)c->plane : (long long)c->vol;
  const BoxSide F{comp_stride, (long long)c->plane, (long long)c->px, 0};
  const size_t es = smi_elem_size(c);
  auto rank_of = [&](int a, int shift) {
    int co[3] = {c->blk_coords[0], c->blk_coords[1], c->blk_coords[2]};
    co[a] = (co[a] + shift + c->blk_topo[a]) % c->blk_topo[a];
    return (co[0] * c->blk_topo[1] + co[1]) * c->blk_topo[2] + co[2];
  };
  for (int a = two_d ? 1 : 0; a < 3; ++a) {
    const int P = c->blk_topo[a], co = c->blk_coords[a];
    const bool per = c->blk_periodic[a];
    const bool lo = co > 0 || per, hi = co < P - 1 || per;
    if (!lo && !hi) continue;
    // extents of the box on every axis: g layers along a, interiors or whole padded extents on the others
    int b0[3], bn[3];
    for (int b = 0; b < 3; ++b) {
      const bool whole = full_exchange && b < a;
      b0[b] = whole ? 0 : gh[b];
      bn[b] = whole ? p[b] : p[b] - 2 * gh[b];
    }
    bn[a] = g;
    const long long n = (long long)ncomp * bn[0] * bn[1] * bn[2];
    const size_t bytes = (size_t)n * es;
    const BoxSide B{(long long)bn[0] * bn[1] * bn[2], (long long)bn[1] * bn[2], (long long)bn[2], 0};
    auto at = [&](int start_a) {  // the box whose first layer along a is start_a
      int o[3] = {b0[0], b0[1], b0[2]};
      o[a] = start_a;
      BoxSide s = F;
      s.off = (long long)o[0] * (long long)c->plane + (long long)o[1] * c->px + o[2];
      return s;
    };
    const BoxSide edge_lo = at(g), edge_hi = at(p[a] - 2 * g), ghost_lo = at(0), ghost_hi = at(p[a] - g);
    const int blocks = (int)((n + 255) / 256);
    auto copy = [&](void* dst, BoxSide D, const void* src, BoxSide S) {
      SMI_DISPATCH(c, {
        k_box_copy<T><<<blocks, 256, 0, c->stream>>>((T*)dst, (const T*)src, bn[0], bn[1], bn[2], D, S, n);
      });
      return 0;
    };
    if (P == 1) {  // periodic onto itself
      copy(field, ghost_lo, field, edge_hi);
      copy(field, ghost_hi, field, edge_lo);
      SMI_LAUNCH_CHECK();
      continue;
    }
    if (c->blk_cap < bytes) {
      SMI_CHECK_HIP(hipSetDevice(c->device));
      SMI_CHECK_HIP(hipStreamSynchronize(c->stream));
      for (void*& b : c->blk_buf) {
        if (b) SMI_CHECK_HIP(hipFree(b));
        b = nullptr;
      }
      c->blk_cap = 0;
      for (void*& b : c->blk_buf) SMI_CHECK_HIP(hipMalloc(&b, bytes));
      c->blk_cap = bytes;
    }
    if (lo) copy(c->blk_buf[0], B, field, edge_lo);
    if (hi) copy(c->blk_buf[1], B, field, edge_hi);
    SMI_LAUNCH_CHECK();
    const int peer_lo = rank_of(a, -1), peer_hi = rank_of(a, +1);
    if (lo && hi && peer_lo == peer_hi) {
      // two ranks on a periodic axis: both faces go to the same peer, one message per call keeps them apart
      const void* s1[1] = {c->blk_buf[0]};
      void* r1[1] = {c->blk_buf[3]};  // the peer's low edge is this block's high ghost
      SMI_CHECK(smi_p2p_exchange(c, 1, &peer_lo, s1, 1, &peer_hi, r1, bytes));
      const void* s2[1] = {c->blk_buf[1]};
      void* r2[1] = {c->blk_buf[2]};
      SMI_CHECK(smi_p2p_exchange(c, 1, &peer_hi, s2, 1, &peer_lo, r2, bytes));
    } else {
      int sp[2], rp[2], ns = 0, nr = 0;
      const void* sb[2];
      void* rb[2];
      if (lo) {
        sp[ns] = peer_lo, sb[ns++] = c->blk_buf[0];
        rp[nr] = peer_lo, rb[nr++] = c->blk_buf[2];
      }
      if (hi) {
        sp[ns] = peer_hi, sb[ns++] = c->blk_buf[1];
        rp[nr] = peer_hi, rb[nr++] = c->blk_buf[3];
      }
      SMI_CHECK(smi_p2p_exchange(c, ns, sp, sb, nr, rp, rb, bytes));
    }
    if (lo) copy(field, ghost_lo, c->blk_buf[2], B);
    if (hi) copy(field, ghost_hi, c->blk_buf[3], B);
    SMI_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int smi_halo_exchange(smi_ctx* c, void* field, int ncomp) {
  SMI_REQUIRE(c && field && ncomp >= 1, "smi_halo_exchange: bad argument");
  return smi_halo_exchange_impl(c, field, ncomp);
}

extern "C" int smi_planes_d2h(smi_ctx* c, const void* field, int ncomp, int z0, int nplanes, void* host) {
  SMI_REQUIRE(c && field && host && ncomp >= 1, "smi_planes_d2h: bad argument");
  SMI_REQUIRE(z0 >= 0 && nplanes >= 1 && z0 + nplanes <= c->pz, "smi_planes_d2h: plane range out of bounds");
  const size_t es = smi_elem_size(c), bytes = (size_t)nplanes * c->plane * es;
  for (int k = 0; k < ncomp; ++k)
    SMI_CHECK_HIP(hipMemcpyAsync((char*)host + k * bytes,
                                 (const char*)field + ((size_t)k * c->vol + (size_t)z0 * c->plane) * es, bytes,
                                 hipMemcpyDeviceToHost, c->stream));
  SMI_CHECK_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

template <typename T>
__global__ void __launch_bounds__(256) k_add(T* __restrict__ dst, const T* __restrict__ src, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] += src[i];
}

extern "C" int smi_planes_h2d(smi_ctx* c, void* field, int ncomp, int z0, int nplanes, const void* host,
                              int accumulate) {
  SMI_REQUIRE(c && field && host && ncomp >= 1, "smi_planes_h2d: bad argument");
  SMI_REQUIRE(z0 >= 0 && nplanes >= 1 && z0 + nplanes <= c->pz, "smi_planes_h2d: plane range out of bounds");
  const size_t es = smi_elem_size(c), n = (size_t)nplanes * c->plane, bytes = n * es;
  for (int k = 0; k < ncomp; ++k) {
    char* dst = (char*)field + ((size_t)k * c->vol + (size_t)z0 * c->plane) * es;
    if (!accumulate) {
      SMI_CHECK_HIP(hipMemcpyAsync(dst, (const char*)host + k * bytes, bytes, hipMemcpyHostToDevice, c->stream));
    } else {
      SMI_REQUIRE(c->halo_recv[0] && bytes <= c->halo_cap, "smi_planes_h2d: staging buffer too small");
      SMI_CHECK_HIP(hipMemcpyAsync(c->halo_recv[0], (const char*)host + k * bytes, bytes, hipMemcpyHostToDevice,
                                   c->stream));
      int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
      SMI_DISPATCH(c, { k_add<T><<<blocks, 256, 0, c->stream>>>((T*)dst, (const T*)c->halo_recv[0], n); });
      SMI_LAUNCH_CHECK();
    }
  }
  SMI_CHECK_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

// zero the g ghost layers on all six sides (MPIGhostSumCommunicator3D.clear_ghost_cells).
// Only the shell is visited: the z ghost planes are contiguous runs, and on an interior plane
// the ghosts are the first and last g rows plus 2g cells of every other row.
template <typename T>
__global__ void __launch_bounds__(256) k_clear_ghost_planes(T* __restrict__ f, int pz, int g, long long plane,
                                                            long long vol) {
  // grid.y = component * 2 + (0: low planes, 1: high planes)
  T* base = f + (long long)(blockIdx.y >> 1) * vol + ((blockIdx.y & 1) ? (long long)(pz - g) * plane : 0);
  const long long n = (long long)g * plane;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    base[i] = T(0);
}
template <typename T>
__global__ void __launch_bounds__(256) k_clear_ghost_rims(T* __restrict__ f, int pz, int py, int px, int g,
                                                          long long plane, long long vol) {
  // grid.y = interior plane, grid.z = component
  T* p = f + (long long)blockIdx.z * vol + (long long)(g + blockIdx.y) * plane;
  const int rows = g * px, rim = 2 * rows + (py - 2 * g) * 2 * g;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rim; i += gridDim.x * blockDim.x) {
    long long off;
    if (i < rows) {
      off = i;
    } else if (i < 2 * rows) {
      off = (long long)(py - g) * px + (i - rows);
    } else {
      const int r = i - 2 * rows, y = g + r / (2 * g), s = r % (2 * g);
      off = (long long)y * px + (s < g ? s : px - 2 * g + s);
    }
    p[off] = T(0);
  }
}

extern "C" int smi_clear_ghost_cells(smi_ctx* c, void* field, int ncomp) {
  SMI_REQUIRE(c && field && ncomp >= 1, "smi_clear_ghost_cells: bad argument");
  if (c->g == 0) return 0;
  const long long n = (long long)c->g * c->plane;
  const int bp = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  const int rim = 2 * c->g * c->px + (c->py - 2 * c->g) * 2 * c->g;
  const int nz_int = c->pz - 2 * c->g;
  SMI_DISPATCH(c, {
    k_clear_ghost_planes<T><<<dim3(bp, 2 * ncomp), 256, 0, c->stream>>>((T*)field, c->pz, c->g, (long long)c->plane,
                                                                       (long long)c->vol);
    if (nz_int > 0)
      k_clear_ghost_rims<T><<<dim3((rim + 255) / 256, nz_int, ncomp), 256, 0, c->stream>>>(
          (T*)field, c->pz, c->py, c->px, c->g, (long long)c->plane, (long long)c->vol);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

// MPIGhostSumCommunicator3D.ghost_sum: my high ghost planes are added to the next z rank's first interior planes,
// my low ghost planes to the previous z rank's last interior planes; then the same with the ghost rows and the y
// neighbours of a pencil (whole faces travel, so what arrived in a ghost row from a z neighbour moves on to the
// diagonal rank: the cascade of the reference's z -> y -> x sequence); then all ghosts are zeroed.
extern "C" int smi_ghost_sum(smi_ctx* c, void* field, int ncomp) {
  SMI_REQUIRE(c && field && ncomp >= 1 && ncomp <= 3, "smi_ghost_sum: bad argument");
  if (c->nranks > 1 && c->exchange_cb) {
    int rc = c->exchange_cb(c->exchange_user, field, ncomp, 1);
    SMI_REQUIRE(rc == 0, "ghost sum callback failed (%d)", rc);
  } else if (c->nranks_z > 1) {
    SMI_REQUIRE(c->comm_ready, "smi_ghost_sum across ranks needs smi_comm_init or an exchange callback");
    const size_t es = smi_elem_size(c);
    const size_t n = (size_t)c->g * c->plane;
    SMI_REQUIRE((size_t)ncomp * n * es <= c->halo_cap, "smi_ghost_sum: staging buffer too small");
    char* base = (char*)field;
    SMI_CHECK_NCCL(ncclGroupStart());
    const int zstep = c->nranks_y * c->nranks_x;
    const int zprev = c->rank - zstep, znext = c->rank + zstep;
    for (int k = 0; k < ncomp; ++k) {
      char* f = base + (size_t)k * c->vol * es;
      if (c->rank_z > 0) {
        SMI_CHECK_NCCL(ncclSend(f, n, nccl_type(c), zprev, c->comm, c->stream));
        SMI_CHECK_NCCL(ncclRecv((char*)c->halo_recv[0] + k * n * es, n, nccl_type(c), zprev, c->comm, c->stream));
      }
      if (c->rank_z < c->nranks_z - 1) {
        SMI_CHECK_NCCL(ncclSend(f + (size_t)(c->pz - c->g) * c->plane * es, n, nccl_type(c), znext, c->comm,
                                c->stream));
        SMI_CHECK_NCCL(ncclRecv((char*)c->halo_recv[1] + k * n * es, n, nccl_type(c), znext, c->comm, c->stream));
      }
    }
    SMI_CHECK_NCCL(ncclGroupEnd());
    int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    for (int k = 0; k < ncomp; ++k) {
      char* f = base + (size_t)k * c->vol * es;
      SMI_DISPATCH(c, {
        if (c->rank_z > 0)
          k_add<T><<<blocks, 256, 0, c->stream>>>((T*)(f + (size_t)c->g * c->plane * es),
                                                 (const T*)((char*)c->halo_recv[0] + k * n * es), n);
        if (c->rank_z < c->nranks_z - 1)
          k_add<T><<<blocks, 256, 0, c->stream>>>((T*)(f + (size_t)(c->pz - 2 * c->g) * c->plane * es),
                                                 (const T*)((char*)c->halo_recv[1] + k * n * es), n);
      });
      SMI_LAUNCH_CHECK();
    }
  }
  SMI_CHECK(smi_y_face_exchange(c, field, ncomp, c->g, true, c->stream));
  return smi_clear_ghost_cells(c, field, ncomp);
}

// ---------------------------------------------------------------- pencil / block <-> slab redistribution
// Rank (rz, ry, rx) of a rank grid holds planes [rz nzp, (rz + 1) nzp), rows [ry nyp, ..) and columns [rx nxp, ..); as
// rank s = (rz Py + ry) Px + rx of the slab decomposition of the same grid it holds planes [s nzs, (s + 1) nzs),
// nzs = nzp / (Py Px), and every row and column.  The planes of a block are therefore the slabs of the Py Px ranks
// that share its rz: one all-to-all of equal blocks inside that group moves a field between the two.
// Block for peer j (block -> slab): [comp][nzs planes j nzs ..][nyp][nxp], interior cells only.
template <typename T, bool TO_WORK>
__global__ void __launch_bounds__(256) k_pencil_blocks(T* __restrict__ work, T* __restrict__ pencil, int ncomp, int nzs,
                                                       int nyp, int nx, int g, int px, long long plane, long long vol,
                                                       long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % nx);
  long long r = i / nx;
  const int y = (int)(r % nyp);
  r /= nyp;
  const int z = (int)(r % nzs);
  r /= nzs;
  const int comp = (int)(r % ncomp), j = (int)(r / ncomp);
  T* cell = pencil + comp * vol + (long long)(g + j * nzs + z) * plane + (long long)(g + y) * px + g + x;
  if (TO_WORK)
    work[i] = *cell;
  else
    *cell = work[i];
}
// Block from peer j = jy Px + jx (in the slab): [comp][nzs][nyp rows jy nyp ..][nxp columns jx nxp ..]
template <typename T, bool TO_WORK>
__global__ void __launch_bounds__(256) k_slab_blocks(T* __restrict__ work, T* __restrict__ slab, int ncomp, int nzs,
                                                     int nyp, int nxp, int Px, int g, int px, long long plane,
                                                     long long vol, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % nxp);
  long long r = i / nxp;
  const int y = (int)(r % nyp);
  r /= nyp;
  const int z = (int)(r % nzs);
  r /= nzs;
  const int comp = (int)(r % ncomp), j = (int)(r / ncomp);
  const int jy = j / Px, jx = j - jy * Px;
  T* cell = slab + comp * vol + (long long)(g + z) * plane + (long long)(g + jy * nyp + y) * px + g + jx * nxp + x;
  if (TO_WORK)
    work[i] = *cell;
  else
    *cell = work[i];
}

static int group_alltoall(smi_ctx* c, const void* send, void* recv, size_t bytes_per_peer, int first, int count) {
  if (count == 1) {
    SMI_CHECK_HIP(hipMemcpyAsync(recv, send, bytes_per_peer, hipMemcpyDeviceToDevice, c->stream));
    return 0;
  }
  if (c->group_alltoall_cb) {
    SMI_CHECK_HIP(hipStreamSynchronize(c->stream));
    int rc = c->group_alltoall_cb(c->exchange_user, send, recv, bytes_per_peer, first, count);
    SMI_REQUIRE(rc == 0, "group all-to-all callback failed (%d)", rc);
    return 0;
  }
  SMI_REQUIRE(c->comm_ready, "redistribution across ranks needs smi_comm_init or a transport callback");
  const int me = c->rank - first;
  SMI_CHECK_HIP(hipMemcpyAsync((char*)recv + (size_t)me * bytes_per_peer, (const char*)send + (size_t)me * bytes_per_peer,
                               bytes_per_peer, hipMemcpyDeviceToDevice, c->stream));
  SMI_CHECK_NCCL(ncclGroupStart());
  for (int j = 0; j < count; ++j) {
    if (j == me) continue;
    SMI_CHECK_NCCL(ncclSend((const char*)send + (size_t)j * bytes_per_peer, bytes_per_peer, ncclChar, first + j, c->comm,
                            c->stream));
    SMI_CHECK_NCCL(ncclRecv((char*)recv + (size_t)j * bytes_per_peer, bytes_per_peer, ncclChar, first + j, c->comm,
                            c->stream));
  }
  SMI_CHECK_NCCL(ncclGroupEnd());
  return 0;
}

static int redistribution_shapes(const smi_ctx* p, const smi_ctx* s, int ncomp, const char* who) {
  SMI_REQUIRE(p && s && ncomp >= 1 && ncomp <= 3, "%s: bad argument", who);
  SMI_REQUIRE(p->dtype == s->dtype && p->g == s->g && p->rank == s->rank && p->nranks == s->nranks &&
                  s->nranks_y == 1 && s->nranks_x == 1,
              "%s: the two contexts do not describe the same rank, precision and ghost size", who);
  const int group = p->nranks_y * p->nranks_x;
  SMI_REQUIRE(p->nz % group == 0 && s->nz == p->nz / group && s->ny == p->ny * p->nranks_y &&
                  s->nx == p->nx * p->nranks_x,
              "%s: a block of %d x %d x %d does not split into slabs of %d x %d x %d over %d ranks", who, p->nz, p->ny,
              p->nx, s->nz, s->ny, s->nx, group);
  return 0;
}

template <bool TO_SLAB>
static int redistribute(smi_ctx* p, smi_ctx* sl, void* slab_field, void* pencil_field, int ncomp, void* work_a,
                        void* work_b) {
  smi_ctx* c = p;
  const int group = p->nranks_y * p->nranks_x, nzs = sl->nz, nyp = p->ny, nxp = p->nx;
  const long long n = (long long)group * ncomp * nzs * nyp * nxp;
  const size_t per_peer = (size_t)ncomp * nzs * nyp * nxp * smi_elem_size(p);
  const int blocks = (int)((n + 255) / 256);
  // both copies run on the pencil context's stream: the caller orders the slab context's work around them
  SMI_DISPATCH(c, {
    if (TO_SLAB)
      k_pencil_blocks<T, true><<<blocks, 256, 0, p->stream>>>((T*)work_a, (T*)pencil_field, ncomp, nzs, nyp, nxp, p->g,
                                                             p->px, (long long)p->plane, (long long)p->vol, n);
    else
      k_slab_blocks<T, true><<<blocks, 256, 0, p->stream>>>((T*)work_a, (T*)slab_field, ncomp, nzs, nyp, nxp,
                                                           p->nranks_x, sl->g, sl->px, (long long)sl->plane,
                                                           (long long)sl->vol, n);
  });
  SMI_LAUNCH_CHECK();
  SMI_CHECK(group_alltoall(p, work_a, work_b, per_peer, p->rank_z * group, group));
  SMI_DISPATCH(c, {
    if (TO_SLAB)
      k_slab_blocks<T, false><<<blocks, 256, 0, p->stream>>>((T*)work_b, (T*)slab_field, ncomp, nzs, nyp, nxp,
                                                            p->nranks_x, sl->g, sl->px, (long long)sl->plane,
                                                            (long long)sl->vol, n);
    else
      k_pencil_blocks<T, false><<<blocks, 256, 0, p->stream>>>((T*)work_b, (T*)pencil_field, ncomp, nzs, nyp, nxp, p->g,
                                                              p->px, (long long)p->plane, (long long)p->vol, n);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int smi_pencil_to_slab(smi_ctx* p, smi_ctx* sl, void* slab_field, const void* pencil_field, int ncomp,
                                  void* work_a, void* work_b) {
  SMI_CHECK(redistribution_shapes(p, sl, ncomp, "smi_pencil_to_slab"));
  SMI_REQUIRE(slab_field && pencil_field && work_a && work_b, "smi_pencil_to_slab: null array");
  return redistribute<true>(p, sl, slab_field, (void*)pencil_field, ncomp, work_a, work_b);
}

extern "C" int smi_slab_to_pencil(smi_ctx* p, smi_ctx* sl, void* pencil_field, const void* slab_field, int ncomp,
                                  void* work_a, void* work_b) {
  SMI_CHECK(redistribution_shapes(p, sl, ncomp, "smi_slab_to_pencil"));
  SMI_REQUIRE(slab_field && pencil_field && work_a && work_b, "smi_slab_to_pencil: null array");
  return redistribute<false>(p, sl, (void*)slab_field, pencil_field, ncomp, work_a, work_b);
}
