// One Navier-Stokes flow step sequenced inside the library (flow_simulators_mpi_3d.py:395-424),
// so that the host pays one call per step instead of one per operator.
#include "smi_common.h"
#include "smi_poisson.h"
#include "smi_prof.h"

static int flow_step_body(smi_ctx* c, smi_poisson* p, void* vorticity, void* velocity, void* stream_func,
                          void* buffer_vector, void* buffer_scalar, void* forcing, double dt, double nu, double dx,
                          int penalty_width, const double* fac_lo, const double* fac_hi, const double* free_stream,
                          int filter_order, int filter_type, int* result_in_buffer);

// Every stencil of this step reads one ghost plane (kernel support 1) and every operator
// refreshes the ghost planes of its own input, so only the plane next to the interior needs to
// travel: half the halo bytes of the reference's full-depth exchange.  (The second ghost plane
// then holds stale values; nothing inside the step reads them into interior cells.)
static int flow_step(smi_ctx* c, smi_poisson* p, void* vorticity, void* velocity, void* stream_func,
                     void* buffer_vector, void* buffer_scalar, void* forcing, double dt, double nu, double dx,
                     int penalty_width, const double* fac_lo, const double* fac_hi, const double* free_stream,
                     int filter_order, int filter_type, int* result_in_buffer) {
  SMI_REQUIRE(c, "smi_flow_step_ns3d: null context");
  SMI_REQUIRE(p, "smi_flow_step_ns3d: null solver");
  SMI_REQUIRE(p->ctx == c || p->pencil == c,
              "smi_flow_step_ns3d: the Poisson solver belongs to another context (a pencil decomposition routes "
              "its solver with smi_poisson_set_pencil_route)");
  const int saved = c->halo_depth;
  const bool shallow = filter_order == 0 || (filter_order == 1 && filter_type == SMI_FILTER_MULTIPLICATIVE);
  if (shallow) c->halo_depth = 1;
  SmiProfScope prof(c, PROF_STEP, 0.0);
  const int rc = flow_step_body(c, p, vorticity, velocity, stream_func, buffer_vector, buffer_scalar, forcing, dt, nu,
                                dx, penalty_width, fac_lo, fac_hi, free_stream, filter_order, filter_type,
                                result_in_buffer);
  c->halo_depth = saved;
  c->fbox_valid = 0;  // the promise covered this step only
  return rc;
}

extern "C" int smi_flow_step_ns3d(smi_ctx* c, smi_poisson* p, void* vorticity, void* velocity, void* stream_func,
                                  void* buffer_vector, void* buffer_scalar, void* forcing, double dt, double nu,
                                  double dx, int penalty_width, const double* fac_lo, const double* fac_hi,
                                  const double* free_stream, int filter_order, int filter_type) {
  return flow_step(c, p, vorticity, velocity, stream_func, buffer_vector, buffer_scalar, forcing, dt, nu, dx,
                   penalty_width, fac_lo, fac_hi, free_stream, filter_order, filter_type, nullptr);
}

// The same step for a caller that can exchange the roles of `vorticity` and `buffer_vector` afterwards: advection and
// diffusion then run as ONE pass from `vorticity` into `buffer_vector`, and the rest of the step (penalisation,
// Poisson solves) works on `buffer_vector`.  *result_in_buffer = 1 says that this happened: the new vorticity is in
// `buffer_vector` and `vorticity` holds scratch; 0: the result is in `vorticity` as with smi_flow_step_ns3d.
extern "C" int smi_flow_step_ns3d_swap(smi_ctx* c, smi_poisson* p, void* vorticity, void* velocity, void* stream_func,
                                       void* buffer_vector, void* buffer_scalar, void* forcing, double dt, double nu,
                                       double dx, int penalty_width, const double* fac_lo, const double* fac_hi,
                                       const double* free_stream, int filter_order, int filter_type,
                                       int* result_in_buffer) {
  SMI_REQUIRE(result_in_buffer, "smi_flow_step_ns3d_swap: null result_in_buffer");
  *result_in_buffer = 0;
  return flow_step(c, p, vorticity, velocity, stream_func, buffer_vector, buffer_scalar, forcing, dt, nu, dx,
                   penalty_width, fac_lo, fac_hi, free_stream, filter_order, filter_type, result_in_buffer);
}

// Promise for the NEXT smi_flow_step_ns3d on this context: its forcing field is exactly zero outside the
// padded-index box [lo, hi] (z, y, x order, inclusive); lo[a] > hi[a] for any axis means zero everywhere;
// null pointers withdraw the promise.  The step then updates the vorticity and resets the forcing only on
// the planes and rows that box can reach (a body's forcing occupies a few percent of the grid).  Single
// rank only: across slabs the ghost sum brings in forcing from nodes this rank does not see, and the
// promise is ignored.
extern "C" int smi_set_forcing_support(smi_ctx* c, const int* lo_zyx, const int* hi_zyx) {
  SMI_REQUIRE(c, "smi_set_forcing_support: null context");
  SMI_REQUIRE((lo_zyx == nullptr) == (hi_zyx == nullptr), "smi_set_forcing_support: give both corners or neither");
  c->fbox_valid = 0;
  if (!lo_zyx || c->nranks > 1) return 0;
  for (int a = 0; a < 3; ++a) {
    c->fbox_lo[a] = lo_zyx[a];
    c->fbox_hi[a] = hi_zyx[a];
  }
  c->fbox_valid = 1;
  return 0;
}

namespace {
// counts the non-zero cells of a three-component field outside the padded-index box [lo, hi]
template <typename T>
__global__ void __launch_bounds__(256) k_count_outside_box(const T* __restrict__ f, long long vol, int py, int px,
                                                          int zlo, int zhi, int ylo, int yhi, int xlo, int xhi,
                                                          unsigned long long* __restrict__ count) {
  unsigned long long mine = 0;
  const long long plane = (long long)py * px;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < 3 * vol; i += (long long)gridDim.x * blockDim.x) {
    const long long cell = i % vol;
    const int z = (int)(cell / plane), y = (int)((cell % plane) / px), x = (int)(cell % px);
    const bool inside = z >= zlo && z <= zhi && y >= ylo && y <= yhi && x >= xlo && x <= xhi;
    if (!inside && f[i] != T(0)) ++mine;
  }
  if (mine) atomicAdd(count, mine);
}
}  // namespace

// The promise of smi_set_forcing_support, verified: *outside_nonzero = number of non-zero cells of `forcing` outside
// the box currently promised on this context (0 when no promise is set).  Synchronises.
extern "C" int smi_check_forcing_support(smi_ctx* c, const void* forcing, long long* outside_nonzero) {
  SMI_REQUIRE(c && forcing && outside_nonzero, "smi_check_forcing_support: null argument");
  *outside_nonzero = 0;
  if (!c->fbox_valid) return 0;
  unsigned long long* count = (unsigned long long*)c->fac_dev;
  SMI_CHECK(smi_memset_zero(c, count, sizeof(unsigned long long)));
  const long long n = 3 * (long long)c->vol;
  const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  SMI_DISPATCH(c, {
    k_count_outside_box<T><<<blocks, 256, 0, c->stream>>>((const T*)forcing, (long long)c->vol, c->py, c->px, c->fbox_lo[0],
                                                         c->fbox_hi[0], c->fbox_lo[1], c->fbox_hi[1], c->fbox_lo[2],
                                                         c->fbox_hi[2], count);
  });
  SMI_LAUNCH_CHECK();
  unsigned long long host = 0;
  SMI_CHECK_HIP(hipMemcpyAsync(&host, count, sizeof(host), hipMemcpyDeviceToHost, c->stream));
  SMI_CHECK_HIP(hipStreamSynchronize(c->stream));
  *outside_nonzero = (long long)host;
  return 0;
}

static int flow_step_body(smi_ctx* c, smi_poisson* p, void* vorticity, void* velocity, void* stream_func,
                                  void* buffer_vector, void* buffer_scalar, void* forcing, double dt, double nu,
                                  double dx, int penalty_width, const double* fac_lo, const double* fac_hi,
                                  const double* free_stream, int filter_order, int filter_type,
                                  int* result_in_buffer) {
  SMI_REQUIRE(c && p && vorticity && velocity && stream_func && buffer_vector && buffer_scalar,
              "smi_flow_step_ns3d: null argument");
  // the prefactors are rounded to the real type exactly as the reference does with real_t(...)
  double pref_curl, pref_diff, pref_vel;
  if (c->dtype == SMI_F32) {
    pref_curl = (double)(float)(dt / (2 * dx));
    pref_diff = (double)(float)(nu * dt / dx / dx);
    pref_vel = (double)(float)(0.5 / dx);
  } else {
    pref_curl = dt / (2 * dx);
    pref_diff = nu * dt / dx / dx;
    pref_vel = 0.5 / dx;
  }
  const bool boxed = forcing && c->fbox_valid && c->nranks == 1;
  if (boxed && c->opt_check_forcing_support) {
    long long outside = 0;
    SMI_CHECK(smi_check_forcing_support(c, forcing, &outside));
    SMI_REQUIRE(outside == 0,
                "smi_flow_step_ns3d: %lld non-zero forcing cells lie outside the box promised with "
                "smi_set_forcing_support ([%d..%d] x [%d..%d] x [%d..%d] in z, y, x)",
                outside, c->fbox_lo[0], c->fbox_hi[0], c->fbox_lo[1], c->fbox_hi[1], c->fbox_lo[2], c->fbox_hi[2]);
  }
  if (boxed)
    SMI_CHECK(smi_forcing_update_boxed_impl(c, vorticity, forcing, pref_curl));
  else if (forcing)
    SMI_CHECK(smi_update_vorticity_from_velocity_forcing(c, vorticity, forcing, pref_curl));
  // Advection and diffusion ping-pong between arrays so that each runs as a single pass: the advected vorticity
  // w + dt/(2dx) curl(u x w) lands in buffer_vector (u x w itself only ever exists in registers) and the diffusion
  // step brings it back into vorticity.  Where the fused kernel serves this context both run as ONE pass into
  // buffer_vector; the result then either comes back through the filter pass, or stays there when the caller has
  // said that it can exchange the two arrays.
  const bool filter1 = filter_order == 1 && filter_type == SMI_FILTER_MULTIPLICATIVE;
  void* w_new = vorticity;  // where the advected, diffused, filtered vorticity ends up
  int fused = 0;
  // The x bands of the penalisation are the expensive ones as a pass of their own (a partial line per row and side).
  // When a band is one 16-byte vector (ghost + zone width = 4) they are penalised in the store of the kernel that
  // writes what the penalisation reads - the one-pass kernel, or the one-pass filter behind it - which is the
  // first of the operator's three axis passes either way.
  const int band = c->g + penalty_width;
  const bool pen_x_in_store = penalty_width > 0 && band == 4 && fac_lo && fac_hi;  // (null tables: reported below)
  int pen_x_done = 0;
  double penx[8];
  if (pen_x_in_store)
    for (int e = 0; e < 4; ++e) {
      penx[e] = fac_lo[e];       // axis x is the first row of the factor tables
      penx[4 + e] = fac_hi[e];
    }
  if (filter1 || (filter_order == 0 && result_in_buffer))
    SMI_CHECK(smi_advect_diffuse_impl(c, buffer_vector, velocity, vorticity, pref_curl, pref_diff, &fused,
                                      pen_x_in_store && !filter1 ? penx : nullptr));
  if (fused) {
    if (filter1) {
      SMI_CHECK(smi_filter_apply_impl(c, vorticity, buffer_vector, 3, pen_x_in_store ? penx : nullptr, &pen_x_done));
    } else {
      w_new = buffer_vector;
      *result_in_buffer = 1;
      pen_x_done = pen_x_in_store;
    }
  } else {
    SMI_CHECK(smi_advect_vorticity_impl(c, buffer_vector, velocity, vorticity, pref_curl));
    if (filter1) {
      // the common filter setting runs as one more single pass: diffuse into the (still dead) stream function
      // array, filter from there back into vorticity
      SMI_CHECK(smi_diffusion_apply_impl(c, stream_func, buffer_vector, pref_diff, 3));
      SMI_CHECK(smi_filter_apply_impl(c, vorticity, stream_func, 3, pen_x_in_store ? penx : nullptr, &pen_x_done));
    } else {
      SMI_CHECK(smi_diffusion_apply_impl(c, vorticity, buffer_vector, pref_diff, 3));
      if (filter_order > 0) {
        size_t stride = c->vol * smi_elem_size(c);
        SMI_CHECK(smi_laplacian_filter(c, vorticity, 3, buffer_vector, (char*)buffer_vector + stride, filter_order,
                                       filter_type));
      }
    }
  }
  SMI_CHECK(smi_penalise_boundary_impl(c, w_new, 3, penalty_width, fac_lo, fac_hi, pen_x_done));
  SMI_CHECK(smi_poisson_vector_solve_for(c, p, stream_func, w_new));
  // curl and the free stream in one pass over the velocity array
  SMI_CHECK(smi_curl_plus_const_impl(c, velocity, stream_func, pref_vel, free_stream, /*track_max=*/true));
  if (boxed) {
    SMI_CHECK(smi_forcing_zero_boxed_impl(c, forcing));
  } else if (forcing) {
    double z3[3] = {0, 0, 0};
    SMI_CHECK(smi_set_fixed_val(c, forcing, 3, z3));
  }
  return 0;
}
