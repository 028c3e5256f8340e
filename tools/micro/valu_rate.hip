// Micro-benchmark: issue rate of scalar (non-packed) f32 VALU instructions on gfx950 as a
// function of the waves resident per SIMD.  Build: hipcc --offload-arch=gfx950 -O3
//   -Xclang -target-feature -Xclang -packed-fp32-ops tools/micro/valu_rate.hip -o build/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ACC>
__global__ void __launch_bounds__(1024) k_fma(float* out, float a, float b, int iters) {
  float x[ACC];
#pragma unroll
  for (int i = 0; i < ACC; ++i) x[i] = (float)threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < ACC; ++i) x[i] = fmaf(x[i], a, b);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < ACC; ++i) s += x[i];
  if (s == 12345.678f) out[0] = s;  // never true: keeps the chain alive
}

int main() {
  float* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 4096, ACC = 8;
  for (int threads : {64, 128, 256, 512, 1024}) {
    for (int wgs_per_cu : {1, 2}) {
      if (threads * wgs_per_cu > 2048) continue;
      const int blocks = 256 * wgs_per_cu;
      k_fma<ACC><<<blocks, threads>>>(out, 1.0001f, 0.5f, 16);
      hipEventRecord(e0);
      k_fma<ACC><<<blocks, threads>>>(out, 1.0001f, 0.5f, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double inst_per_wave = (double)iters * 16 * ACC;
      const double waves = (double)blocks * threads / 64;
      const double per_simd = inst_per_wave * waves / 1024.0;  // wave-instructions per SIMD
      printf("threads %4d x %d WG/CU (%4.1f waves/SIMD): %.3f ms  %.2f ns per wave-instr per SIMD  %.1f TFLOP/s\n", threads,
             wgs_per_cu, waves / 1024.0, ms, ms * 1e6 / per_simd, inst_per_wave * waves * 128 / (ms * 1e9));
    }
  }
  return 0;
}
