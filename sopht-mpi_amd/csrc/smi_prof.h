// Internal: per-kernel timing with HIP events on the context's stream.  Off unless
// smi_prof_enable(ctx, 1); a scope records one event before and one after the launches
// it brackets, durations are resolved in smi_prof_get after a stream synchronise.
#pragma once
#include <vector>

#include "smi_common.h"

enum SmiProfSlot {
  PROF_CROSS = 0,
  PROF_CURL_UPDATE,
  PROF_DIFFUSION_FLUX,
  PROF_EW_SUM,
  PROF_FILTER,
  PROF_PENALISE,
  PROF_CURL,
  PROF_ADD_SET,
  PROF_POISSON_PAD,
  PROF_POISSON_FFT_FWD,
  PROF_POISSON_MUL,
  PROF_POISSON_FFT_INV,
  PROF_POISSON_EXTRACT,
  PROF_FFT_X_FWD,
  PROF_FFT_Y_FWD,
  PROF_FFT_Z_CONV,
  PROF_FFT_Y_INV,
  PROF_FFT_X_INV,
  PROF_TRANSPOSE,
  PROF_HALO,
  PROF_IB_FORCING,
  PROF_IB_SPREAD,
  PROF_FUSED_ADVECT,
  PROF_FUSED_DIFFUSE,
  PROF_ADVECTION,
  PROF_FUSED_ADVECT_DIFFUSE,
  PROF_STEP,  // one whole smi_flow_step_ns3d (brackets the kernel slots above)
  PROF_NUM_SLOTS
};

struct SmiProfRecord {
  int slot;
  hipEvent_t a, b;
  double bytes;
};

struct SmiProf {
  bool on = false;
  std::vector<SmiProfRecord> live;
  std::vector<hipEvent_t> pool;
  long long launches[PROF_NUM_SLOTS] = {};
  double total_ms[PROF_NUM_SLOTS] = {};
  double bytes[PROF_NUM_SLOTS] = {};
  // individual durations (ms) of the first kProfMaxSamples scopes of each slot since the last reset
  std::vector<float> samples[PROF_NUM_SLOTS];
};
constexpr size_t kProfMaxSamples = 4096;

const char* smi_prof_slot_name(int slot);
// bytes of one interior scalar field (the unit S of the algorithmic byte counts)
static inline double smi_S(const smi_ctx* c) { return (double)c->nz * c->ny * c->nx * (double)smi_elem_size(c); }

struct SmiProfScope {
  smi_ctx* c;
  SmiProfRecord rec;
  bool active;
  SmiProfScope(smi_ctx* ctx, int slot, double alg_bytes);
  ~SmiProfScope();
};
