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
  // smi_poisson_set_pencil_route: the pencil context whose fields this solver also serves, the slab arrays the
  // right-hand side and the solution pass through, and two work arrays of the redistribution
  smi_ctx* pencil;
  void* route_rhs;
  void* route_sol;
  void* route_work[2];
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
// the vector solve for fields of context c: p's own z-slab context, or the pencil context routed to it
int smi_poisson_vector_solve_for(smi_ctx* c, smi_poisson* p, void* solution, const void* rhs);
