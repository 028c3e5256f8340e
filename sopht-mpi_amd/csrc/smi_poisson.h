// Internal: Poisson solver object shared by the rocFFT and the pruned back ends.
#pragma once
#include "smi_common.h"

struct RocfftState;
struct PrunedState;

struct smi_poisson {
  smi_ctx* ctx;
  int backend;
  int gnz, gny, gnx;  // global interior sizes
  double dx;          // grid spacing (value of the context's real type)
  double x_range;
  RocfftState* rocfft;
  PrunedState* pruned;
};

bool smi_pruned_supports(int gnz, int gny, int gnx);
int smi_pruned_create(smi_poisson* p);
int smi_pruned_destroy(smi_poisson* p);
int smi_pruned_solve(smi_poisson* p, void* solution, const void* rhs, int ncomp);
size_t smi_pruned_greens_size(smi_poisson* p);
int smi_pruned_greens_download(smi_poisson* p, void* host);
// the solve on the library's fixed per-GPU scratch array (smi_fft.hip)
int smi_pruned_solve_noalloc(smi_poisson* p, void* solution, const void* rhs);
size_t smi_pruned_noalloc_capacity(void);
