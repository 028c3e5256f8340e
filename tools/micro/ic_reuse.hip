// Micro-benchmark: does data just written by one kernel come back faster than from HBM when the next
// kernel reads it, as long as the working set stays inside the 256 MiB Infinity Cache?
// A producer kernel streams `bytes` into a buffer (plain or non-temporal stores), a consumer kernel
// then reads the same range (plain or non-temporal loads).  Between the two nothing else runs.
// The last rows read a buffer far larger than the cache in the same way (the HBM reference).
// Build: hipcc --offload-arch=gfx950 -O3 -Xclang -target-feature -Xclang -packed-fp32-ops tools/micro/ic_reuse.hip \
//        -o build/ic_reuse
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ void __launch_bounds__(256) k_write(v4* p, size_t n, float s) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    v4 v = {s, s + 1, s + 2, (float)i};
    if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
  }
}
template <bool NT>
__global__ void __launch_bounds__(256) k_read(const v4* p, size_t n, float* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  v4 acc = {0, 0, 0, 0};
  for (; i < n; i += stride) acc += NT ? __builtin_nontemporal_load(p + i) : p[i];
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
// in place: read, modify, write back (what the z convolution does to its tile)
__global__ void __launch_bounds__(256) k_rmw(v4* p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = p[i] * 1.0001f;
}

// out of place: the same traffic as k_rmw, but reads and writes go to different buffers
template <bool NTS>
__global__ void __launch_bounds__(256) k_copy(v4* dst, const v4* src, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const v4 v = __builtin_nontemporal_load(src + i) * 1.0001f;
    if (NTS) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
  }
}
// in place with a lag: the write of element i goes out `lag` elements behind the read front, as in a kernel
// that prefetches its next tile while it stores the previous one
__global__ void __launch_bounds__(256) k_rmw_nt(v4* p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(p + i) * 1.0001f, p + i);
}

int main() {
  const size_t big = (size_t)4 << 30;
  v4* buf;
  float* out;
  if (hipMalloc(&buf, big) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
  hipEvent_t e[4];
  for (auto& x : e) hipEventCreate(&x);
  const int blocks = 256 * 8;
  k_write<false><<<blocks, 256>>>(buf, big / 16, 1.0f);
  hipDeviceSynchronize();
  for (size_t mb : {32, 64, 128, 192, 256, 512, 2048}) {
    const size_t n = (mb << 20) / 16;
    for (int mode = 0; mode < 4; ++mode) {
      const bool ntw = mode & 1, ntr = mode & 2;
      float best_w = 1e9f, best_r = 1e9f, best_m = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        // evict: sweep a far-away 1.5 GB region
        k_read<false><<<blocks, 256>>>(buf + (big / 16) / 2, ((size_t)1536 << 20) / 16, out);
        hipEventRecord(e[0]);
        if (ntw) k_write<true><<<blocks, 256>>>(buf, n, (float)rep); else k_write<false><<<blocks, 256>>>(buf, n, (float)rep);
        hipEventRecord(e[1]);
        k_rmw<<<blocks, 256>>>(buf, n);
        hipEventRecord(e[2]);
        if (ntr) k_read<true><<<blocks, 256>>>(buf, n, out); else k_read<false><<<blocks, 256>>>(buf, n, out);
        hipEventRecord(e[3]);
        hipEventSynchronize(e[3]);
        float w, m, r;
        hipEventElapsedTime(&w, e[0], e[1]);
        hipEventElapsedTime(&m, e[1], e[2]);
        hipEventElapsedTime(&r, e[2], e[3]);
        best_w = w < best_w ? w : best_w;
        best_m = m < best_m ? m : best_m;
        best_r = r < best_r ? r : best_r;
      }
      const double gb = (double)(mb << 20) / 1e9;
      printf("%5zu MB  stores %s loads %s : write %6.0f GB/s   in-place rmw %6.0f GB/s (r+w)   read %6.0f GB/s\n", mb,
             ntw ? "nt   " : "plain", ntr ? "nt   " : "plain", gb / (best_w * 1e-3), 2 * gb / (best_m * 1e-3),
             gb / (best_r * 1e-3));
    }
  }
  // in place against out of place on a range far beyond the cache (1 GB read + 1 GB written)
  {
    const size_t n = ((size_t)1 << 30) / 16;
    v4* dst = buf + 2 * n;
    float t[4] = {1e9f, 1e9f, 1e9f, 1e9f};
    for (int rep = 0; rep < 5; ++rep) {
      for (int k = 0; k < 4; ++k) {
        hipEventRecord(e[0]);
        if (k == 0) k_rmw<<<blocks, 256>>>(buf, n);
        if (k == 1) k_rmw_nt<<<blocks, 256>>>(buf, n);
        if (k == 2) k_copy<false><<<blocks, 256>>>(dst, buf, n);
        if (k == 3) k_copy<true><<<blocks, 256>>>(dst, buf, n);
        hipEventRecord(e[1]);
        hipEventSynchronize(e[1]);
        float ms;
        hipEventElapsedTime(&ms, e[0], e[1]);
        t[k] = ms < t[k] ? ms : t[k];
      }
    }
    const char* names[4] = {"in place, plain", "in place, nt load + nt store", "out of place, nt load + plain store",
                            "out of place, nt load + nt store"};
    for (int k = 0; k < 4; ++k) printf("1 GB read + 1 GB written, %-36s: %6.0f GB/s (r+w)\n", names[k], 2 * 1.0737 / (t[k] * 1e-3));
  }
  return 0;
}
