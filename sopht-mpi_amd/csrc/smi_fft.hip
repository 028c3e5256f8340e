// Pruned Poisson back end (placeholder until the LDS FFT kernels land).
#include "smi_common.h"
#include "smi_poisson.h"

int smi_pruned_create(smi_poisson*) {
  smi_set_error("SMI_POISSON_PRUNED is not available in this build");
  return 2;
}
int smi_pruned_destroy(smi_poisson*) { return 0; }
int smi_pruned_solve(smi_poisson*, void*, const void*, int) {
  smi_set_error("SMI_POISSON_PRUNED is not available in this build");
  return 2;
}
size_t smi_pruned_greens_size(smi_poisson*) { return 0; }
int smi_pruned_greens_download(smi_poisson*, void*) {
  smi_set_error("SMI_POISSON_PRUNED is not available in this build");
  return 2;
}
