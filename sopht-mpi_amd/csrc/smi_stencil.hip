// Stencil operators of libsophtmi (gfx950): diffusion flux, curl / vorticity update,
// divergence, boundary penalisation, Laplacian filter.
//
// Layout: every field is a ghosted [z][y][x] array.  A z-plane is treated as a flat
// array of py*px reals: lane i of a workgroup owns V = 4 consecutive reals (one 16-byte
// access for f32) at flat plane offset q, so loads and stores are fully coalesced and
// the y+-1 neighbours are the same flat vector shifted by +-px (still 16-byte aligned
// because px % 4 == 0 on the vector path).  Each lane marches along z over a chunk of
// planes keeping the z-1 / z / z+1 values in registers, so every plane is fetched from
// HBM once per chunk (+2 halo planes); the y+-1 and x+-1 re-reads hit L1/L2.
//
// Which cells an operator writes follows the reference MPI wrappers exactly (seven
// overlapping boxes per call, then zeroing of the physical boundary bands):
//   D = { z in [zlo_c, zhi_c], y and x in [g+1, p-g-2] }   for flux / curl / filter
//   U = A + B + C                                          for the vorticity update
// with the sets spelled out at each kernel.
#include <cstdlib>

#include "smi_common.h"
#include "smi_prof.h"

namespace {

struct Geo {
  int pz, py, px, g;
  long long plane, vol;
  int zlo_c, zhi_c;  // inclusive z range of the zero-banded compute set D
  int ylo_c, yhi_c;  // the same in y (differs from [g + 1, py - g - 2] only on the inner faces of a pencil)
  int xlo_c, xhi_c;  // and in x (inner faces of a rank grid that divides x)
  // z range over which the one-pass advection + diffusion kernel forms the advected vorticity on the core cells:
  // the planes the vorticity update writes there, [g, pz - g - 1], plus the ghost plane next to a neighbouring
  // rank (what that rank computes for its own edge plane, needed by the diffusion stencil of this rank's edge)
  int za_lo, za_hi;
  // the same across an inner y face of a pencil: the ghost row there is the neighbour's FACE row (on_face in its
  // frame), so the one-pass kernel applies the face part of the update set to it; -1 next to a physical face
  int yf_lo, yf_hi;
  int xf_lo, xf_hi;  // likewise the ghost column across an inner x face
};

Geo make_geo(const smi_ctx* c) {
  Geo G;
  G.pz = c->pz;
  G.py = c->py;
  G.px = c->px;
  G.g = c->g;
  G.plane = (long long)c->plane;
  G.vol = (long long)c->vol;
  // diffusion_flux_mpi_3d.py:161-192: g+1 layers are zeroed next to physical faces;
  // at a rank boundary the first / last interior plane is computed from exchanged ghosts.
  G.zlo_c = c->g + (c->phys_zlo ? 1 : 0);
  G.zhi_c = c->pz - c->g - 1 - (c->phys_zhi ? 1 : 0);
  G.ylo_c = c->g + (c->phys_ylo ? 1 : 0);
  G.yhi_c = c->py - c->g - 1 - (c->phys_yhi ? 1 : 0);
  G.za_lo = c->g - (c->phys_zlo ? 0 : 1);
  G.za_hi = c->pz - c->g - 1 + (c->phys_zhi ? 0 : 1);
  G.yf_lo = c->phys_ylo ? -1 : c->g - 1;
  G.yf_hi = c->phys_yhi ? -1 : c->py - c->g;
  G.xlo_c = c->g + (c->phys_xlo ? 1 : 0);
  G.xhi_c = c->px - c->g - 1 - (c->phys_xhi ? 1 : 0);
  G.xf_lo = c->phys_xlo ? -1 : c->g - 1;
  G.xf_hi = c->phys_xhi ? -1 : c->px - c->g;
  return G;
}

constexpr int kBlock = 256;

struct LaunchShape {
  dim3 grid;
  int zchunk;
};

template <int V, int BLOCK = kBlock>
LaunchShape plan(const smi_ctx* c, int ncomp) {
  long long lanes = ((long long)c->plane + V - 1) / V;
  int bx = (int)((lanes + BLOCK - 1) / BLOCK);
  // aim for >= 2048 workgroups while keeping z chunks long enough to amortise the two
  // halo planes each chunk re-reads
  int zchunk = (int)((long long)c->pz * bx * ncomp / 2048);
  if (zchunk < 8) zchunk = 8;
  if (zchunk > 32) zchunk = 32;
  if (zchunk > c->pz) zchunk = c->pz;
  int nchunk = (c->pz + zchunk - 1) / zchunk;
  LaunchShape s;
  s.grid = dim3(bx, nchunk, ncomp);
  s.zchunk = zchunk;
  return s;
}

template <typename T, int V>
struct Pack {
  T v[V];
};

template <typename T, int V>
__device__ __forceinline__ Pack<T, V> ld(const T* p) {
  Pack<T, V> r;
  if constexpr (V == 4) {
    Vec4<T> t = load4(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) r.v[e] = t.v[e];
  } else {
    r.v[0] = p[0];
  }
  return r;
}
template <typename T, int V>
__device__ __forceinline__ void st(T* p, const Pack<T, V>& r) {
  if constexpr (V == 4) {
    Vec4<T> t;
#pragma unroll
    for (int e = 0; e < 4; ++e) t.v[e] = r.v[e];
    store4(p, t);
  } else {
    p[0] = r.v[0];
  }
}
template <typename T, int V>
__device__ __forceinline__ Pack<T, V> zero_pack() {
  Pack<T, V> r;
#pragma unroll
  for (int e = 0; e < V; ++e) r.v[e] = T(0);
  return r;
}

// Logical block index of this workgroup.  Workgroups are handed to the eight XCDs round
// robin in launch order, and each XCD has its own L2; the stencil halos (rows y+-1, the two
// planes at the ends of a z chunk) belong to the neighbouring blocks.  Giving every XCD one
// contiguous range of the launch-order index keeps those neighbours behind the same L2.
// It pays for the fused advection kernel (six input fields with halos); the lighter kernels
// measured marginally faster in plain launch order.
struct Blk {
  int x, y, z;
};
template <bool REMAP>
__device__ __forceinline__ Blk xcd_block() {
  if (!REMAP) return Blk{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
  constexpr unsigned kXcd = 8;
  const unsigned nx = gridDim.x, ny = gridDim.y, total = nx * ny * gridDim.z;
  const unsigned lin = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
  const unsigned xcd = lin % kXcd, idx = lin / kXcd, per = total / kXcd, rem = total % kXcd;
  const unsigned log = xcd * per + min(xcd, rem) + idx;
  Blk b;
  b.x = (int)(log % nx);
  const unsigned r = log / nx;
  b.y = (int)(r % ny);
  b.z = (int)(r / ny);
  return b;
}

// flat plane offset of this lane, its row and first column; false when past the plane
template <int V>
__device__ __forceinline__ bool lane_pos(const Geo& G, const Blk& B, long long& q, int& y, int& x0) {
  q = ((long long)B.x * blockDim.x + threadIdx.x) * V;
  if (q >= G.plane) return false;
  y = (int)(q / G.px);
  x0 = (int)(q - (long long)y * G.px);
  return true;
}

__device__ __forceinline__ bool in_core(int a, int p, int g) { return a >= g + 1 && a <= p - g - 2; }
// rows of the zero-banded compute set D (in_core(y) unless a y face of this rank borders another rank)
__device__ __forceinline__ bool in_core_y(const Geo& G, int y) { return y >= G.ylo_c && y <= G.yhi_c; }
__device__ __forceinline__ bool in_core_x(const Geo& G, int x) { return x >= G.xlo_c && x <= G.xhi_c; }
__device__ __forceinline__ bool on_face(int a, int p, int g) { return a == g || a == p - g - 1; }
__device__ __forceinline__ bool in_ext(int a, int p) { return a >= 1 && a <= p - 2; }

// ---------------------------------------------------------------- diffusion flux
// flux = c * (sum of the six neighbours - 6 f) on D, 0 elsewhere.
// ADD: the output is f + flux instead (every cell written, a copy of f outside D): the whole
// diffusion timestep in one pass when the result may land in another array than its input.
template <typename T, int V, bool ADD = false>
__global__ void __launch_bounds__(kBlock) k_diffusion_flux(T* __restrict__ flux, const T* __restrict__ f, T c,
                                                           Geo G, int zchunk) {
  long long q;
  int y, x0;
  const Blk B = xcd_block<false>();
  if (!lane_pos<V>(G, B, q, y, x0)) return;
  flux += (long long)B.z * G.vol + q;
  f += (long long)B.z * G.vol + q;
  bool core[V], any = false;
  bool ycore = in_core_y(G, y);
#pragma unroll
  for (int e = 0; e < V; ++e) {
    core[e] = ycore && in_core_x(G, x0 + e);
    any |= core[e];
  }
  int z0 = B.y * zchunk, z1 = min(z0 + zchunk, G.pz);
  int zs = max(z0, G.zlo_c), ze = min(z1 - 1, G.zhi_c);
  const Pack<T, V> zero = zero_pack<T, V>();
  auto outside = [&](int z) { st<T, V>(flux + z * G.plane, ADD ? ld<T, V>(f + z * G.plane) : zero); };
  if (!any || zs > ze) {
    for (int z = z0; z < z1; ++z) outside(z);
    return;
  }
  for (int z = z0; z < zs; ++z) outside(z);
  Pack<T, V> prev = ld<T, V>(f + (zs - 1) * G.plane), cur = ld<T, V>(f + zs * G.plane);
  for (int z = zs; z <= ze; ++z) {
    const T* pc = f + z * G.plane;
    Pack<T, V> nxt = ld<T, V>(pc + G.plane);
    Pack<T, V> ym = ld<T, V>(pc - G.px), yp = ld<T, V>(pc + G.px);
    T xm = pc[-1], xp = pc[V];
    Pack<T, V> o;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      T left = e == 0 ? xm : cur.v[e - 1];
      T right = e == V - 1 ? xp : cur.v[e + 1];
      T val = c * (right + left + yp.v[e] + ym.v[e] + nxt.v[e] + prev.v[e] - T(6) * cur.v[e]);
      o.v[e] = core[e] ? val : T(0);
      if (ADD) o.v[e] = cur.v[e] + o.v[e];
    }
    st<T, V>(flux + z * G.plane, o);
    prev = cur;
    cur = nxt;
  }
  for (int z = ze + 1; z < z1; ++z) outside(z);
}

// ---------------------------------------------------------------- curl / vorticity update
// MODE 0 (curl_mpi_3d.py):            out = p * curl(f) on D, 0 elsewhere.
// MODE 1 (update_vorticity_..._3d.py): out += p * curl(f) on U = A + B + C where, with
//   core(a) = a in [g+1, p-g-2], face(a) = a in {g, p-g-1}, ext(a) = a in [1, p-2]:
//   A = { z in [g, pz-g-1], core(y), core(x) }      interior box and the two z boxes
//   B = { ext(z), face(y), core(x) }                the two y boxes
//   C = { ext(z), ext(y), face(x) }                 the two x boxes
// (no band zeroing: update_vorticity_from_velocity_forcing_mpi_3d.py has no PROC_NULL logic).
// MODE 0 on the 16-byte path is served by k_curl_tiled; this kernel remains its scalar-path form.
// The launch may cover only part of the array (MODE 1 with a known support of f): lanes start at flat
// plane offset q_begin and stop before q_end, planes run from z_begin to z_end - 1.
struct CurlRange {
  long long q_begin, q_end;
  int z_begin, z_end;
};
// DIFF: the curl of f - h (update_vorticity_from_penalised_velocity: f the penalised velocity, h the velocity), the
// difference formed as each value is loaded.
template <typename T, int V, int MODE, bool DIFF = false>
__global__ void __launch_bounds__(kBlock) k_curl(T* out, const T* __restrict__ f, T pref, Geo G, int zchunk,
                                                 CurlRange R, const T* __restrict__ h = nullptr) {
  const Blk B = xcd_block<false>();
  auto ldv = [&](const T* p) {
    Pack<T, V> a = ld<T, V>(p);
    if constexpr (DIFF) {
      const Pack<T, V> b = ld<T, V>(h + (p - f));
#pragma unroll
      for (int e = 0; e < V; ++e) a.v[e] -= b.v[e];
    }
    return a;
  };
  auto ld1 = [&](const T* p) {
    if constexpr (DIFF) return p[0] - h[p - f];
    else return p[0];
  };
  const long long q = R.q_begin + ((long long)B.x * blockDim.x + threadIdx.x) * V;
  if (q >= R.q_end) return;
  const int y = (int)(q / G.px), x0 = (int)(q - (long long)y * G.px);
  const T* fx = f + q;
  const T* fy = f + G.vol + q;
  const T* fz = f + 2 * G.vol + q;
  T* ox = out + q;
  T* oy = out + G.vol + q;
  T* oz = out + 2 * G.vol + q;
  bool mA[V], mB[V], any = false;
  // MODE 0 masks with the zero-banded set D, whose y range depends on the rank's position; the update set U of
  // MODE 1 is the same on every rank
  const bool yc = MODE == 0 ? in_core_y(G, y) : in_core(y, G.py, G.g), yf = on_face(y, G.py, G.g), ye = in_ext(y, G.py);
#pragma unroll
  for (int e = 0; e < V; ++e) {
    bool xc = MODE == 0 ? in_core_x(G, x0 + e) : in_core(x0 + e, G.px, G.g), xf = on_face(x0 + e, G.px, G.g);
    mA[e] = yc && xc;
    mB[e] = MODE != 0 && ((yf && xc) || (ye && xf));
    any |= mA[e] || mB[e];
  }
  int z0 = R.z_begin + B.y * zchunk, z1 = min(z0 + zchunk, R.z_end);
  int zs, ze;
  if (MODE == 0) {
    zs = max(z0, G.zlo_c);
    ze = min(z1 - 1, G.zhi_c);
  } else {
    zs = max(z0, 1);
    ze = min(z1 - 1, G.pz - 2);
  }
  const Pack<T, V> zero = zero_pack<T, V>();
  // cells this lane does not compute: 0 (MODE 0), untouched (MODE 1)
  auto outside = [&](int z) {
    if (MODE == 1) return;
    const long long zo = z * G.plane;
    st<T, V>(ox + zo, zero);
    st<T, V>(oy + zo, zero);
    st<T, V>(oz + zo, zero);
  };
  if (!any || zs > ze) {
    for (int z = z0; z < z1; ++z) outside(z);
    return;
  }
  for (int z = z0; z < zs; ++z) outside(z);
  Pack<T, V> fx_prev = ldv(fx + (zs - 1) * G.plane), fx_cur = ldv(fx + zs * G.plane);
  Pack<T, V> fy_prev = ldv(fy + (zs - 1) * G.plane), fy_cur = ldv(fy + zs * G.plane);
  for (int z = zs; z <= ze; ++z) {
    const long long zo = z * G.plane;
    Pack<T, V> fx_nxt = ldv(fx + zo + G.plane), fy_nxt = ldv(fy + zo + G.plane);
    Pack<T, V> fx_ym = ldv(fx + zo - G.px), fx_yp = ldv(fx + zo + G.px);
    Pack<T, V> fz_ym = ldv(fz + zo - G.px), fz_yp = ldv(fz + zo + G.px);
    Pack<T, V> fz_c = ldv(fz + zo);
    T fy_xm = ld1(fy + zo - 1), fy_xp = ld1(fy + zo + V);
    T fz_xm = ld1(fz + zo - 1), fz_xp = ld1(fz + zo + V);
    Pack<T, V> rx, ry, rz;
    bool m[V], anym = false;
    const bool zA = z >= G.g && z <= G.pz - G.g - 1;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      T fyL = e == 0 ? fy_xm : fy_cur.v[e - 1], fyR = e == V - 1 ? fy_xp : fy_cur.v[e + 1];
      T fzL = e == 0 ? fz_xm : fz_c.v[e - 1], fzR = e == V - 1 ? fz_xp : fz_c.v[e + 1];
      rx.v[e] = pref * (fz_yp.v[e] - fz_ym.v[e] - fy_nxt.v[e] + fy_prev.v[e]);
      ry.v[e] = pref * (fx_nxt.v[e] - fx_prev.v[e] - fzR + fzL);
      rz.v[e] = pref * (fyR - fyL - fx_yp.v[e] + fx_ym.v[e]);
      m[e] = MODE == 0 ? mA[e] : (mB[e] || (mA[e] && zA));
      anym |= m[e];
    }
    if (MODE == 0) {
#pragma unroll
      for (int e = 0; e < V; ++e) {
        rx.v[e] = m[e] ? rx.v[e] : T(0);
        ry.v[e] = m[e] ? ry.v[e] : T(0);
        rz.v[e] = m[e] ? rz.v[e] : T(0);
      }
      st<T, V>(ox + zo, rx);
      st<T, V>(oy + zo, ry);
      st<T, V>(oz + zo, rz);
    } else if (anym) {
      Pack<T, V> wx = ld<T, V>(ox + zo), wy = ld<T, V>(oy + zo), wz = ld<T, V>(oz + zo);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        wx.v[e] = m[e] ? wx.v[e] + rx.v[e] : wx.v[e];
        wy.v[e] = m[e] ? wy.v[e] + ry.v[e] : wy.v[e];
        wz.v[e] = m[e] ? wz.v[e] + rz.v[e] : wz.v[e];
      }
      st<T, V>(ox + zo, wx);
      st<T, V>(oy + zo, wy);
      st<T, V>(oz + zo, wz);
    }
    fx_prev = fx_cur;
    fx_cur = fx_nxt;
    fy_prev = fy_cur;
    fy_cur = fy_nxt;
  }
  for (int z = ze + 1; z < z1; ++z) outside(z);
}

// ---------------------------------------------------------------- fused advection update
// out = w + p * curl(u x w) on U (the update set of MODE 1 above), out = w elsewhere: the
// cross product kernel followed by the vorticity update, without the round trip of u x w
// through HBM.  b = u x w is formed in registers wherever the curl stencil needs it: the
// plane ahead (kept in the z window as it becomes the centre and then the plane behind),
// the rows y-1 / y+1 and the two cells left and right of the lane's vector.
template <typename T, int V>
struct Vec3 {
  Pack<T, V> x, y, z;
};
template <typename T, int V>
__device__ __forceinline__ Vec3<T, V> ld3(const T* p, long long vol) {
  return {ld<T, V>(p), ld<T, V>(p + vol), ld<T, V>(p + 2 * vol)};
}

constexpr int kAdvectBlock = 256;  // larger blocks (fewer y halo rows per tile) measured slower
template <typename T, int V>
__global__ void __launch_bounds__(kAdvectBlock) k_advect(T* __restrict__ out, const T* __restrict__ vel,
                                                  const T* __restrict__ vort, T pref, Geo G, int zchunk) {
  long long q;
  int y, x0;
  const Blk B = xcd_block<true>();
  if (!lane_pos<V>(G, B, q, y, x0)) return;
  const T* u = vel + q;
  const T* w = vort + q;
  T* o = out + q;
  bool mA[V], mB[V], any = false;
  const bool yc = in_core(y, G.py, G.g), yf = on_face(y, G.py, G.g), ye = in_ext(y, G.py);
#pragma unroll
  for (int e = 0; e < V; ++e) {
    bool xc = in_core(x0 + e, G.px, G.g), xf = on_face(x0 + e, G.px, G.g);
    mA[e] = yc && xc;
    mB[e] = (yf && xc) || (ye && xf);
    any |= mA[e] || mB[e];
  }
  const int z0 = B.y * zchunk, z1 = min(z0 + zchunk, G.pz);
  const int zs = max(z0, 1), ze = min(z1 - 1, G.pz - 2);
  auto copy_plane = [&](int z) {
    const long long zo = z * G.plane;
    st<T, V>(o + zo, ld<T, V>(w + zo));
    st<T, V>(o + G.vol + zo, ld<T, V>(w + G.vol + zo));
    st<T, V>(o + 2 * G.vol + zo, ld<T, V>(w + 2 * G.vol + zo));
  };
  if (!any || zs > ze) {
    for (int z = z0; z < z1; ++z) copy_plane(z);
    return;
  }
  for (int z = z0; z < zs; ++z) copy_plane(z);
  // z window: b_x, b_y behind; b_y, b_z, w at the centre; everything ahead
  Pack<T, V> bx_p, by_p, bx_c, by_c, bz_c;
  Vec3<T, V> w_c;
  {
    Vec3<T, V> up = ld3<T, V>(u + (zs - 1) * G.plane, G.vol), wp = ld3<T, V>(w + (zs - 1) * G.plane, G.vol);
    Vec3<T, V> uc = ld3<T, V>(u + zs * G.plane, G.vol);
    w_c = ld3<T, V>(w + zs * G.plane, G.vol);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      bx_p.v[e] = up.y.v[e] * wp.z.v[e] - up.z.v[e] * wp.y.v[e];
      by_p.v[e] = up.z.v[e] * wp.x.v[e] - up.x.v[e] * wp.z.v[e];
      bx_c.v[e] = uc.y.v[e] * w_c.z.v[e] - uc.z.v[e] * w_c.y.v[e];
      by_c.v[e] = uc.z.v[e] * w_c.x.v[e] - uc.x.v[e] * w_c.z.v[e];
      bz_c.v[e] = uc.x.v[e] * w_c.y.v[e] - uc.y.v[e] * w_c.x.v[e];
    }
  }
  for (int z = zs; z <= ze; ++z) {
    const long long zo = z * G.plane;
    // plane ahead
    Vec3<T, V> un = ld3<T, V>(u + zo + G.plane, G.vol), w_n = ld3<T, V>(w + zo + G.plane, G.vol);
    Pack<T, V> bx_n, by_n, bz_n;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      bx_n.v[e] = un.y.v[e] * w_n.z.v[e] - un.z.v[e] * w_n.y.v[e];
      by_n.v[e] = un.z.v[e] * w_n.x.v[e] - un.x.v[e] * w_n.z.v[e];
      bz_n.v[e] = un.x.v[e] * w_n.y.v[e] - un.y.v[e] * w_n.x.v[e];
    }
    // rows y-1, y+1: b_x and b_z
    Pack<T, V> bx_ym, bz_ym, bx_yp, bz_yp;
    {
      Vec3<T, V> a = ld3<T, V>(u + zo - G.px, G.vol), b = ld3<T, V>(w + zo - G.px, G.vol);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        bx_ym.v[e] = a.y.v[e] * b.z.v[e] - a.z.v[e] * b.y.v[e];
        bz_ym.v[e] = a.x.v[e] * b.y.v[e] - a.y.v[e] * b.x.v[e];
      }
    }
    {
      Vec3<T, V> a = ld3<T, V>(u + zo + G.px, G.vol), b = ld3<T, V>(w + zo + G.px, G.vol);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        bx_yp.v[e] = a.y.v[e] * b.z.v[e] - a.z.v[e] * b.y.v[e];
        bz_yp.v[e] = a.x.v[e] * b.y.v[e] - a.y.v[e] * b.x.v[e];
      }
    }
    // cells x0-1 and x0+V: b_y and b_z
    T by_xm, bz_xm, by_xp, bz_xp;
    {
      const T* ul = u + zo - 1;
      const T* wl = w + zo - 1;
      T ax = ul[0], ay = ul[G.vol], az = ul[2 * G.vol], cx = wl[0], cy = wl[G.vol], cz = wl[2 * G.vol];
      by_xm = az * cx - ax * cz;
      bz_xm = ax * cy - ay * cx;
      const T* ur = u + zo + V;
      const T* wr = w + zo + V;
      ax = ur[0], ay = ur[G.vol], az = ur[2 * G.vol], cx = wr[0], cy = wr[G.vol], cz = wr[2 * G.vol];
      by_xp = az * cx - ax * cz;
      bz_xp = ax * cy - ay * cx;
    }
    const bool zA = z >= G.g && z <= G.pz - G.g - 1;
    Pack<T, V> rx, ry, rz;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const T byL = e == 0 ? by_xm : by_c.v[e - 1], byR = e == V - 1 ? by_xp : by_c.v[e + 1];
      const T bzL = e == 0 ? bz_xm : bz_c.v[e - 1], bzR = e == V - 1 ? bz_xp : bz_c.v[e + 1];
      const bool m = mB[e] || (mA[e] && zA);
      const T cx = pref * (bz_yp.v[e] - bz_ym.v[e] - by_n.v[e] + by_p.v[e]);
      const T cy = pref * (bx_n.v[e] - bx_p.v[e] - bzR + bzL);
      const T cz = pref * (byR - byL - bx_yp.v[e] + bx_ym.v[e]);
      rx.v[e] = m ? w_c.x.v[e] + cx : w_c.x.v[e];
      ry.v[e] = m ? w_c.y.v[e] + cy : w_c.y.v[e];
      rz.v[e] = m ? w_c.z.v[e] + cz : w_c.z.v[e];
    }
    st<T, V>(o + zo, rx);
    st<T, V>(o + G.vol + zo, ry);
    st<T, V>(o + 2 * G.vol + zo, rz);
    bx_p = bx_c;
    by_p = by_c;
    bx_c = bx_n;
    by_c = by_n;
    bz_c = bz_n;
    w_c = w_n;
  }
  for (int z = ze + 1; z < z1; ++z) copy_plane(z);
}

// ---------------------------------------------------------------- fused advection, LDS tiled
// Same operator as k_advect on the 16-byte path, organised so that HBM sees every input cell
// once: a workgroup owns a tile of TY rows x 4 TXL columns and marches along z.  b = u x w of
// the plane ahead is formed by the lane that owns the cell (and by a few lanes for the ring of
// halo cells around the tile) and parked in one of two LDS tiles; a step later the curl stencil
// reads its y+-1 / x+-1 neighbours from there.  One barrier per plane: a lane writes plane z+1
// into the other tile only after the barrier of step z, which every lane reaches after its
// reads of step z-1.
struct TileShape {
  int txl, ty, ntx, nty;  // lanes (of 4 cells) per tile row, rows per tile, tiles per plane
  int pitch, rows;        // LDS row pitch in elements (own cells start at column 4), ty + 2
};

template <typename T>
__device__ __forceinline__ void cross3(const Pack<T, 4>& ux, const Pack<T, 4>& uy, const Pack<T, 4>& uz,
                                       const Pack<T, 4>& wx, const Pack<T, 4>& wy, const Pack<T, 4>& wz,
                                       Pack<T, 4>& bx, Pack<T, 4>& by, Pack<T, 4>& bz) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    bx.v[e] = uy.v[e] * wz.v[e] - uz.v[e] * wy.v[e];
    by.v[e] = uz.v[e] * wx.v[e] - ux.v[e] * wz.v[e];
    bz.v[e] = ux.v[e] * wy.v[e] - uy.v[e] * wx.v[e];
  }
}

constexpr int kTileThreads = 512;

template <typename T>
__global__ void __launch_bounds__(kTileThreads) k_advect_tiled(T* __restrict__ out, const T* __restrict__ vel,
                                                              const T* __restrict__ vort, T pref, Geo G,
                                                              int zchunk, TileShape S) {
  extern __shared__ char smi_tile_smem[];
  T* lds = reinterpret_cast<T*>(smi_tile_smem);
  const Blk B = xcd_block<true>();
  const int tyi = B.x / S.ntx, txi = B.x - tyi * S.ntx;
  const int tid = threadIdx.x;
  const int ly = tid / S.txl, lx = tid - ly * S.txl;
  const int ty0 = tyi * S.ty, tx0 = txi * S.txl * 4;
  const int y = ty0 + ly, x0 = tx0 + 4 * lx;
  const bool own = ly < S.ty && y < G.py && x0 < G.px;
  const long long q = (long long)y * G.px + x0;
  const size_t comp = (size_t)S.rows * S.pitch;  // LDS elements per component tile
  auto tile = [&](int buf, int c) { return lds + ((size_t)buf * 3 + c) * comp; };
  const int own_off = (ly + 1) * S.pitch + 4 + 4 * lx;
  // halo ring duty of this lane (lanes 0 .. 2 txl + 2 ty - 1)
  const int nrow = 2 * S.txl, ntask = nrow + 2 * S.ty;
  bool h_on = tid < ntask, h_vec = tid < nrow;
  int hy, hx, h_off;
  if (h_vec) {
    const int top = tid >= S.txl, hl = tid - top * S.txl;
    hy = top ? ty0 + S.ty : ty0 - 1;
    hx = tx0 + 4 * hl;
    h_off = (top ? S.ty + 1 : 0) * S.pitch + 4 + 4 * hl;
    h_on = h_on && hy >= 0 && hy < G.py && hx < G.px;
  } else {
    const int k = tid - nrow, right = k >= S.ty, r = k - right * S.ty;
    hy = ty0 + r;
    hx = right ? tx0 + 4 * S.txl : tx0 - 1;
    h_off = (r + 1) * S.pitch + (right ? 4 + 4 * S.txl : 3);
    h_on = h_on && hy < G.py && hx >= 0 && hx < G.px;
  }
  const long long hq = (long long)hy * G.px + hx;

  bool mA[4], mB[4];
  {
    const bool yc = in_core(y, G.py, G.g), yf = on_face(y, G.py, G.g), ye = in_ext(y, G.py);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool xc = in_core(x0 + e, G.px, G.g), xf = on_face(x0 + e, G.px, G.g);
      mA[e] = own && yc && xc;
      mB[e] = own && ((yf && xc) || (ye && xf));
    }
  }
  const int z0 = B.y * zchunk, z1 = min(z0 + zchunk, G.pz);
  const int zs = max(z0, 1), ze = min(z1 - 1, G.pz - 2);
  auto copy_plane = [&](int z) {
    if (!own) return;
    const long long zo = z * G.plane + q;
#pragma unroll
    for (int c = 0; c < 3; ++c) st<T, 4>(out + c * G.vol + zo, ld<T, 4>(vort + c * G.vol + zo));
  };
  if (zs > ze) {  // uniform over the workgroup
    for (int z = z0; z < z1; ++z) copy_plane(z);
    return;
  }
  for (int z = z0; z < zs; ++z) copy_plane(z);

  // b of one plane into a tile: the lane's own four cells and its halo duty
  auto load_own = [&](int z, Vec3<T, 4>& u, Vec3<T, 4>& w) {
    if (own) {
      u = ld3<T, 4>(vel + z * G.plane + q, G.vol);
      w = ld3<T, 4>(vort + z * G.plane + q, G.vol);
    } else {
      u = w = Vec3<T, 4>{zero_pack<T, 4>(), zero_pack<T, 4>(), zero_pack<T, 4>()};
    }
  };
  auto park = [&](int buf, const Pack<T, 4>& bx, const Pack<T, 4>& by, const Pack<T, 4>& bz) {
    if (!own) return;
    st<T, 4>(tile(buf, 0) + own_off, bx);
    st<T, 4>(tile(buf, 1) + own_off, by);
    st<T, 4>(tile(buf, 2) + own_off, bz);
  };
  auto halo = [&](int z, int buf) {
    if (!h_on) return;
    const T* u = vel + z * G.plane + hq;
    const T* w = vort + z * G.plane + hq;
    if (h_vec) {
      Vec3<T, 4> a = ld3<T, 4>(u, G.vol), b = ld3<T, 4>(w, G.vol);
      Pack<T, 4> bx, by, bz;
      cross3(a.x, a.y, a.z, b.x, b.y, b.z, bx, by, bz);
      st<T, 4>(tile(buf, 0) + h_off, bx);
      st<T, 4>(tile(buf, 1) + h_off, by);
      st<T, 4>(tile(buf, 2) + h_off, bz);
    } else {
      const T ax = u[0], ay = u[G.vol], az = u[2 * G.vol], cx = w[0], cy = w[G.vol], cz = w[2 * G.vol];
      tile(buf, 0)[h_off] = ay * cz - az * cy;
      tile(buf, 1)[h_off] = az * cx - ax * cz;
      tile(buf, 2)[h_off] = ax * cy - ay * cx;
    }
  };

  Pack<T, 4> bx_p, by_p, bx_c, by_c, bz_c;
  Vec3<T, 4> w_c;
  {
    Vec3<T, 4> u, w;
    load_own(zs - 1, u, w);
    Pack<T, 4> bz_unused;
    cross3(u.x, u.y, u.z, w.x, w.y, w.z, bx_p, by_p, bz_unused);
    load_own(zs, u, w_c);
    cross3(u.x, u.y, u.z, w_c.x, w_c.y, w_c.z, bx_c, by_c, bz_c);
    park(zs & 1, bx_c, by_c, bz_c);
    halo(zs, zs & 1);
  }
  for (int z = zs; z <= ze; ++z) {
    Vec3<T, 4> u_n, w_n;
    load_own(z + 1, u_n, w_n);  // in flight across the barrier
    __syncthreads();
    const int buf = z & 1;
    if (own) {
      const T* tx = tile(buf, 0) + own_off;
      const T* ty = tile(buf, 1) + own_off;
      const T* tz = tile(buf, 2) + own_off;
      const Pack<T, 4> bx_ym = ld<T, 4>(tx - S.pitch), bx_yp = ld<T, 4>(tx + S.pitch);
      const Pack<T, 4> bz_ym = ld<T, 4>(tz - S.pitch), bz_yp = ld<T, 4>(tz + S.pitch);
      const T by_xm = ty[-1], by_xp = ty[4], bz_xm = tz[-1], bz_xp = tz[4];
      const bool zA = z >= G.g && z <= G.pz - G.g - 1;
      Pack<T, 4> rx, ry, rz;
      // needs b_x, b_y of the plane ahead: computed below from u_n, w_n, so form it first
      Pack<T, 4> bx_n, by_n, bz_n;
      cross3(u_n.x, u_n.y, u_n.z, w_n.x, w_n.y, w_n.z, bx_n, by_n, bz_n);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const T byL = e == 0 ? by_xm : by_c.v[e - 1], byR = e == 3 ? by_xp : by_c.v[e + 1];
        const T bzL = e == 0 ? bz_xm : bz_c.v[e - 1], bzR = e == 3 ? bz_xp : bz_c.v[e + 1];
        const bool m = mB[e] || (mA[e] && zA);
        const T cx = pref * (bz_yp.v[e] - bz_ym.v[e] - by_n.v[e] + by_p.v[e]);
        const T cy = pref * (bx_n.v[e] - bx_p.v[e] - bzR + bzL);
        const T cz = pref * (byR - byL - bx_yp.v[e] + bx_ym.v[e]);
        rx.v[e] = m ? w_c.x.v[e] + cx : w_c.x.v[e];
        ry.v[e] = m ? w_c.y.v[e] + cy : w_c.y.v[e];
        rz.v[e] = m ? w_c.z.v[e] + cz : w_c.z.v[e];
      }
      const long long zo = z * G.plane + q;
      st<T, 4>(out + zo, rx);
      st<T, 4>(out + G.vol + zo, ry);
      st<T, 4>(out + 2 * G.vol + zo, rz);
      bx_p = bx_c;
      by_p = by_c;
      bx_c = bx_n;
      by_c = by_n;
      bz_c = bz_n;
      w_c = w_n;
      park(buf ^ 1, bx_c, by_c, bz_c);
    }
    if (z < ze) halo(z + 1, buf ^ 1);
  }
  for (int z = ze + 1; z < z1; ++z) copy_plane(z);
}

// ---------------------------------------------------------------- advection + diffusion in one pass
// out = w* + c lap7(w*) on D, w* elsewhere, with w* = w + p curl(u x w) on U, w elsewhere: the rotational
// advection update followed by the diffusion timestep (flow_simulators_mpi_3d.py:395-411) without the round trip
// of the advected vorticity w* through HBM (6 S of 15).  Both stencils have support 1, so an output cell needs w*
// on its six neighbours and those need b = u x w one cell further out.  A workgroup owns a tile of OY rows x 4 OX
// columns and runs one lane per 16-byte vector of the region two rows / one vector larger on every side
// (RY = OY + 4 rows of RX = OX + 2 vectors): every lane loads u and w of its vector and forms b and w*, the lanes of
// the tile proper also form and store the output.  The region marches along z.  In-plane neighbours of b and of w*
// come from two sets of LDS tiles, the z - 1 / z + 1 values stay in the lane's registers.  Per plane: park b of
// plane z + 1 and w* of plane z; barrier; w* of plane z + 1 from b; output of plane z from w*; barrier.
// The arithmetic (operand order included) is that of k_advect_tiled and k_diffuse_tiled.  Lanes outside the array
// or in the outer ring read valid memory and compute values nobody uses; selects, not arithmetic, apply the masks.
struct FusedShape {
  int rx, ry;      // region: vectors per row, rows
  int ox, oy;      // tile proper: rx - 2, ry - 4
  int ntx, nty;    // tiles per plane
  int pitch;       // LDS row pitch in elements (the lane's vector starts at column 4 + 4 * its index)
};
// Lanes per workgroup: 768 in single precision (twelve waves, three per SIMD: up to 168 registers per lane; the kernel
// holds 157).  With 1024 lanes (128 registers) it spilled five to seven registers whose reloads - evicted from the L2
// by the planes streaming through between two uses - showed as 0.4 GB of extra fetches per launch at 512^3; measured
// 1024 -> 768 lanes: 1.33 -> 1.24 ms at 512^3, 0.60 -> 0.49 ms at 384^3, and with the next plane requested a whole step
// ahead, which 168 registers allow, 1.21 and 0.48 ms.  Double precision holds twice the register state per lane and
// runs 512 lanes (256 registers), with LDS tiles of the same byte size.
#ifndef SMI_FUSED_LANES_F32
#define SMI_FUSED_LANES_F32 768
#endif
template <typename T>
constexpr int fused_threads() {
  return sizeof(T) == 4 ? SMI_FUSED_LANES_F32 : 512;
}

// Penalisation of the x bands on the way out (penalise_field_boundary_mpi_3d.py:196-212, the first of the three
// passes of the operator that follows the one-pass kernels in the step): with ghost + zone width = 4 a band is exactly the
// first (last) 16-byte vector of a row and its source cell, layer 3 (layer px - 4), is an element of the same vector.
// As a pass of its own it costs 0.1 ms at 512^3 (one partial line per row and side, read and written back).
template <typename T>
struct PenX {
  int on_lo, on_hi;
  T lo[4], hi[4];
};

// the low / high x band as one vector (v[3] resp. v[0] is its source cell): band = 1 / 2
template <typename T>
__device__ __forceinline__ void pen_x_apply(Pack<T, 4>& o, int band, const PenX<T>& X) {
  if (band) {
    const bool low = band == 1;
    const T sv = low ? o.v[3] : o.v[0];
#pragma unroll
    for (int e = 0; e < 4; ++e) o.v[e] = sv * (low ? X.lo[e] : X.hi[e]);
  }
}

template <typename T>
__global__ void __launch_bounds__(fused_threads<T>()) k_advect_diffuse(T* __restrict__ out, const T* __restrict__ vel,
                                                                 const T* __restrict__ vort, T pref, T cdiff, Geo G,
                                                                 int zchunk, FusedShape S, int zbase, int zjump,
                                                                 int zlimit, PenX<T> X) {
  extern __shared__ char smi_tile_smem[];
  T* lds = reinterpret_cast<T*>(smi_tile_smem);
  const Blk B = xcd_block<true>();
  const int tyi = B.x / S.ntx, txi = B.x - tyi * S.ntx;
  const int tid = threadIdx.x;
  const int ry = tid / S.rx, rxl = tid - ry * S.rx;
  const bool lane_on = ry < S.ry;
  const int y = tyi * S.oy - 2 + ry, x0 = (txi * S.ox - 1 + rxl) * 4;
  const bool in_arr = lane_on && y >= 0 && y < G.py && x0 >= 0 && x0 < G.px;
  const bool own = in_arr && ry >= 2 && ry < S.ry - 2 && rxl >= 1 && rxl < S.rx - 1;
  const long long q = in_arr ? (long long)y * G.px + x0 : 0;
  const size_t comp = (size_t)(S.ry + 2) * S.pitch;
  T* tb = lds;              // b_x, b_y, b_z of the plane ahead
  T* tw = lds + 3 * comp;   // w*_x, w*_y, w*_z of the current plane
  const int off = (ry + 1) * S.pitch + 4 + 4 * rxl;
  // per cell of the vector: in the sets A (z-dependent) and B of the update set (bits 0-3), and, in bits 8-11 of mA,
  // in the rows and columns of the diffusion set D (the same as A except on the inner y faces of a pencil)
  unsigned mA = 0, mB = 0;
  if (in_arr) {
    const bool yc = in_core(y, G.py, G.g), ye = in_ext(y, G.py), yd = in_core_y(G, y);
    const bool yf = on_face(y, G.py, G.g) || y == G.yf_lo || y == G.yf_hi;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool xc = in_core(x0 + e, G.px, G.g), xd = in_core_x(G, x0 + e);
      const bool xf = on_face(x0 + e, G.px, G.g) || x0 + e == G.xf_lo || x0 + e == G.xf_hi;
      mA |= ((unsigned)(yc && xc) << e) | ((unsigned)(yd && xd) << (8 + e));
      mB |= (unsigned)((yf && xc) || (ye && xf)) << e;
    }
    // bits 12, 13: this lane's vector is the low / high x band to be penalised
    mA |= ((unsigned)(X.on_lo && x0 == 0) << 12) | ((unsigned)(X.on_hi && x0 == G.px - 4) << 13);
  }
  // planes of this workgroup: zchunk of them from zbase + zjump * (position along grid y), below zlimit - so that one
  // launch can cover a run of chunks (zjump = zchunk) or a short chunk at either end of the array (zjump = pz - zchunk)
  const int z0 = zbase + zjump * B.y, z1 = min(z0 + zchunk, zlimit);
  auto plane_of = [&](int z) { return (long long)min(max(z, 0), G.pz - 1) * G.plane + q; };
  Vec3<T, 4> lu, lw;  // u and w of the plane two ahead, in flight
  Pack<T, 4> bPx, bPy, bCx, bCy, bCz, dummy;
  Vec3<T, 4> wC, Zm, Wk;
  Zm = Wk = Vec3<T, 4>{zero_pack<T, 4>(), zero_pack<T, 4>(), zero_pack<T, 4>()};
  {
    lu = ld3<T, 4>(vel + plane_of(z0 - 2), G.vol);
    lw = ld3<T, 4>(vort + plane_of(z0 - 2), G.vol);
    cross3(lu.x, lu.y, lu.z, lw.x, lw.y, lw.z, bPx, bPy, dummy);
    lu = ld3<T, 4>(vel + plane_of(z0 - 1), G.vol);
    wC = ld3<T, 4>(vort + plane_of(z0 - 1), G.vol);
    cross3(lu.x, lu.y, lu.z, wC.x, wC.y, wC.z, bCx, bCy, bCz);
    lu = ld3<T, 4>(vel + plane_of(z0), G.vol);
    lw = ld3<T, 4>(vort + plane_of(z0), G.vol);
  }
  for (int k = z0 - 2; k < z1; ++k) {
    if (lane_on) {
      st<T, 4>(tb + off, bCx);
      st<T, 4>(tb + comp + off, bCy);
      st<T, 4>(tb + 2 * comp + off, bCz);
      st<T, 4>(tw + off, Wk.x);
      st<T, 4>(tw + comp + off, Wk.y);
      st<T, 4>(tw + 2 * comp + off, Wk.z);
    }
    __syncthreads();
    // b and w of plane k + 2 from the loads issued a step (double precision: half a step) ago
    Pack<T, 4> bNx, bNy, bNz;
    cross3(lu.x, lu.y, lu.z, lw.x, lw.y, lw.z, bNx, bNy, bNz);
    const Vec3<T, 4> wN = lw;
    Vec3<T, 4> Wn = wC;
    // u and w of plane k + 3: single precision requests them here, a whole curl and diffusion ahead of their use (its
    // 168 registers hold both sets in flight); double precision, which would spill, once the curl's registers are free
    constexpr bool EARLY = sizeof(T) == 4;
    if (EARLY && k + 3 <= z1 + 1) {  // uniform
      lu = ld3<T, 4>(vel + plane_of(k + 3), G.vol);
      lw = ld3<T, 4>(vort + plane_of(k + 3), G.vol);
    }
    if (lane_on) {
      // w* of plane k + 1
      const int j = k + 1;
      const unsigned m = ((j >= 1 && j <= G.pz - 2) ? mB : 0u) | ((j >= G.za_lo && j <= G.za_hi) ? mA : 0u);
      const T* tx = tb + off;
      const T* ty = tb + comp + off;
      const T* tz = tb + 2 * comp + off;
      const Pack<T, 4> bx_ym = ld<T, 4>(tx - S.pitch), bx_yp = ld<T, 4>(tx + S.pitch);
      const Pack<T, 4> bz_ym = ld<T, 4>(tz - S.pitch), bz_yp = ld<T, 4>(tz + S.pitch);
      const T by_xm = ty[-1], by_xp = ty[4], bz_xm = tz[-1], bz_xp = tz[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const T byL = e == 0 ? by_xm : bCy.v[e - 1], byR = e == 3 ? by_xp : bCy.v[e + 1];
        const T bzL = e == 0 ? bz_xm : bCz.v[e - 1], bzR = e == 3 ? bz_xp : bCz.v[e + 1];
        const T cx = pref * (bz_yp.v[e] - bz_ym.v[e] - bNy.v[e] + bPy.v[e]);
        const T cy = pref * (bNx.v[e] - bPx.v[e] - bzR + bzL);
        const T cz = pref * (byR - byL - bx_yp.v[e] + bx_ym.v[e]);
        const bool on = (m >> e) & 1u;
        Wn.x.v[e] = on ? wC.x.v[e] + cx : wC.x.v[e];
        Wn.y.v[e] = on ? wC.y.v[e] + cy : wC.y.v[e];
        Wn.z.v[e] = on ? wC.z.v[e] + cz : wC.z.v[e];
      }
    }
    if (!EARLY && k + 3 <= z1 + 1) {  // uniform
      lu = ld3<T, 4>(vel + plane_of(k + 3), G.vol);
      lw = ld3<T, 4>(vort + plane_of(k + 3), G.vol);
    }
    {
      // output of plane k
      if (own && k >= z0) {
        const unsigned d = (k >= G.zlo_c && k <= G.zhi_c) ? mA >> 8 : 0u;
        const long long zo = (long long)k * G.plane + q;
        const Pack<T, 4>* cur3[3] = {&Wk.x, &Wk.y, &Wk.z};
        const Pack<T, 4>* prv3[3] = {&Zm.x, &Zm.y, &Zm.z};
        const Pack<T, 4>* nxt3[3] = {&Wn.x, &Wn.y, &Wn.z};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const T* tc = tw + c * comp + off;
          const Pack<T, 4> ym = ld<T, 4>(tc - S.pitch), yp = ld<T, 4>(tc + S.pitch);
          const T xm = tc[-1], xp = tc[4];
          const Pack<T, 4>& cur = *cur3[c];
          Pack<T, 4> o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const T left = e == 0 ? xm : cur.v[e - 1], right = e == 3 ? xp : cur.v[e + 1];
            const T val = cdiff * (right + left + yp.v[e] + ym.v[e] + nxt3[c]->v[e] + prv3[c]->v[e] - T(6) * cur.v[e]);
            o.v[e] = cur.v[e] + (((d >> e) & 1u) ? val : T(0));
          }
          pen_x_apply(o, (int)(mA >> 12) & 3, X);
          st<T, 4>(out + c * G.vol + zo, o);
        }
      }
    }
    __syncthreads();
    bPx = bCx;
    bPy = bCy;
    bCx = bNx;
    bCy = bNy;
    bCz = bNz;
    wC = wN;
    Zm = Wk;
    Wk = Wn;
  }
}

// Region shape for a padded plane of py rows x px / 4 vectors: the (rx, ry) with rx * ry <= fused_threads<T>() whose tiles
// (rx - 2 vectors x ry - 4 rows) cover the plane with the fewest lanes, the LDS budget permitting.
template <typename T>
static bool fused_shape(const smi_ctx* c, FusedShape& best) {
  const int nl = c->px / 4;
  double best_cost = 1e300;
  bool found = false;
  for (int rx = 6; rx <= 128; ++rx) {
    const int ry = fused_threads<T>() / rx;
    if (ry < 6) break;
    const int ox = rx - 2, oy = ry - 4;
    const int ntx = (nl + ox - 1) / ox, nty = (c->py + oy - 1) / oy;
    const int pitch = 4 * rx + 8;
    const size_t lds = (size_t)6 * (ry + 2) * pitch * sizeof(T);
    if (lds > 150 * 1024) continue;
    const double cost = (double)ntx * nty * rx * ry;
    if (cost < best_cost) {
      best_cost = cost;
      best = FusedShape{rx, ry, ox, oy, ntx, nty, pitch};
      found = true;
    }
  }
  return found;
}

// Position of a lane in the branch-free tiled kernels: its own four cells and its share of the
// halo around the tile, always as one aligned 16-byte vector (lanes 0 .. 2 txl - 1: the rows
// below / above; the next 2 ty lanes: the vector that ends / starts just outside a row, which
// the LDS pitch of 4 txl + 8 has room for).  Lanes without cells or without halo duty read
// the start of the plane and write a spare LDS row (`rows` + 1 of rows + 2), so that the
// plane loop has no divergent branch apart from the final store.
struct TileLane {
  int y, x0;
  bool own;
  long long q, hq;   // plane offsets of the own vector and of the halo vector
  int own_off, h_off;  // LDS offsets within one component tile
};
// corners: the rows below / above run one vector further on either side (kernels whose stencil reaches diagonally)
__device__ __forceinline__ TileLane tile_lane(const Geo& G, const TileShape& S, const Blk& B, bool corners = false) {
  TileLane L;
  const int tyi = B.x / S.ntx, txi = B.x - tyi * S.ntx;
  const int tid = threadIdx.x;
  const int ly = tid / S.txl, lx = tid - ly * S.txl;
  const int ty0 = tyi * S.ty, tx0 = txi * S.txl * 4;
  const int spare = S.rows * S.pitch + 4;
  L.y = ty0 + ly;
  L.x0 = tx0 + 4 * lx;
  L.own = ly < S.ty && L.y < G.py && L.x0 < G.px;
  L.q = L.own ? (long long)L.y * G.px + L.x0 : 0;
  L.own_off = L.own ? (ly + 1) * S.pitch + 4 + 4 * lx : spare;
  const int rowl = S.txl + (corners ? 2 : 0), first = corners ? -1 : 0;  // vectors per halo row, the first one
  const int nrow = 2 * rowl, ntask = nrow + 2 * S.ty;
  int hy, hx, off;
  if (tid < nrow) {
    const int top = tid >= rowl, hl = tid - top * rowl + first;
    hy = top ? ty0 + S.ty : ty0 - 1;
    hx = tx0 + 4 * hl;
    off = (top ? S.ty + 1 : 0) * S.pitch + 4 + 4 * hl;
  } else {
    const int k = tid - nrow, right = k >= S.ty, r = k - right * S.ty;
    hy = ty0 + r;
    hx = right ? tx0 + 4 * S.txl : tx0 - 4;
    off = (r + 1) * S.pitch + (right ? 4 + 4 * S.txl : 0);
  }
  const bool h_on = tid < ntask && hy >= 0 && hy < G.py && hx >= 0 && hx < G.px;
  L.hq = h_on ? (long long)hy * G.px + hx : 0;
  L.h_off = h_on ? off : spare;
  return L;
}

// ---------------------------------------------------------------- diffusion, LDS tiled
// k_diffusion_flux on the 16-byte path with the tiling of k_advect_tiled: the plane a lane is
// about to need sits in one of two LDS tiles (own cells and the halo), its z-1 / z+1 values in
// registers.  HBM then sees each input cell once per tile instead of once per neighbouring
// row's workgroup (those run behind other XCDs' L2s).  Grid z = component.
// Software pipeline, two planes deep: the loads of plane z + 2 are issued at the top of step z
// and first touched at the end of step z + 1, and they are older than the store of step z,
// so no step waits for a load or a store it has just issued.
template <typename T, bool ADD>
__global__ void __launch_bounds__(kTileThreads) k_diffuse_tiled(T* __restrict__ out, const T* __restrict__ f, T c,
                                                               Geo G, int zchunk, TileShape S) {
  extern __shared__ char smi_tile_smem[];
  T* lds = reinterpret_cast<T*>(smi_tile_smem);
  const Blk B = xcd_block<true>();
  const TileLane L = tile_lane(G, S, B);
  out += (long long)B.z * G.vol;
  f += (long long)B.z * G.vol;
  const int comp = (S.rows + 2) * S.pitch;
  bool core[4];
  {
    const bool yc = L.own && in_core_y(G, L.y);
#pragma unroll
    for (int e = 0; e < 4; ++e) core[e] = yc && in_core_x(G, L.x0 + e);
  }
  const int z0 = B.y * zchunk, z1 = min(z0 + zchunk, G.pz);
  const int zs = max(z0, G.zlo_c), ze = min(z1 - 1, G.zhi_c);
  const bool run = zs <= ze;  // uniform over the workgroup
  for (int z = z0; z < (run ? zs : z1); ++z) {
    if (L.own) st<T, 4>(out + z * G.plane + L.q, ADD ? ld<T, 4>(f + z * G.plane + L.q) : zero_pack<T, 4>());
  }
  if (!run) return;
  const T* fo = f + L.q;
  const T* fh = f + L.hq;
  // The planes rotate through four register sets (and the halo through two) by the argument
  // order of step(), not by copies: a copy of a register with a load in flight would wait for it.
  Pack<T, 4> pa = ld<T, 4>(fo + (zs - 1) * G.plane), pb = ld<T, 4>(fo + zs * G.plane);
  Pack<T, 4> pc = ld<T, 4>(fo + (zs + 1) * G.plane), pd, ha = ld<T, 4>(fh + (zs + 1) * G.plane), hb;
  st<T, 4>(lds + (zs & 1) * comp + L.own_off, pb);
  st<T, 4>(lds + (zs & 1) * comp + L.h_off, ld<T, 4>(fh + zs * G.plane));
  auto step = [&](int z, const Pack<T, 4>& prev, const Pack<T, 4>& cur, const Pack<T, 4>& nxt, Pack<T, 4>& nxt2,
                  const Pack<T, 4>& h1, Pack<T, 4>& h2) {
    const long long zn = (long long)min(z + 2, G.pz - 1) * G.plane;
    nxt2 = ld<T, 4>(fo + zn);  // in flight across this step
    h2 = ld<T, 4>(fh + zn);
    __syncthreads();
    const int buf = z & 1;
    const T* tc = lds + buf * comp + L.own_off;
    const Pack<T, 4> ym = ld<T, 4>(tc - S.pitch), yp = ld<T, 4>(tc + S.pitch);
    const T xm = tc[-1], xp = tc[4];
    Pack<T, 4> o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const T left = e == 0 ? xm : cur.v[e - 1], right = e == 3 ? xp : cur.v[e + 1];
      const T val = c * (right + left + yp.v[e] + ym.v[e] + nxt.v[e] + prev.v[e] - T(6) * cur.v[e]);
      o.v[e] = core[e] ? val : T(0);
      if (ADD) o.v[e] = cur.v[e] + o.v[e];
    }
    if (L.own) st<T, 4>(out + z * G.plane + L.q, o);
    st<T, 4>(lds + (buf ^ 1) * comp + L.own_off, nxt);
    st<T, 4>(lds + (buf ^ 1) * comp + L.h_off, h1);
  };
  for (int z = zs; z <= ze; z += 4) {
    step(z, pa, pb, pc, pd, ha, hb);
    if (z + 1 > ze) break;
    step(z + 1, pb, pc, pd, pa, hb, ha);
    if (z + 2 > ze) break;
    step(z + 2, pc, pd, pa, pb, ha, hb);
    if (z + 3 > ze) break;
    step(z + 3, pd, pa, pb, pc, hb, ha);
  }
  for (int z = ze + 1; z < z1; ++z) {
    if (L.own) st<T, 4>(out + z * G.plane + L.q, ADD ? ld<T, 4>(f + z * G.plane + L.q) : zero_pack<T, 4>());
  }
}

// ---------------------------------------------------------------- curl, LDS tiled
// k_curl MODE 0 on the 16-byte path, tiled and pipelined like k_diffuse_tiled: the three
// components of the current plane are parked in LDS for the y+-1 / x+-1 reads, f_x and f_y of
// the planes z-1 / z+1 stay in registers.  `add` is a constant vector added to every cell of the
// output, banded or not (the free stream that the flow step adds to the whole velocity array).
template <typename T>
struct Const3 {
  T x, y, z;
};
// Bit pattern of a non-negative real: these order like the reals, so an integer atomic maximum serves.
__device__ __forceinline__ void atomic_max_nonneg(void* slot, float v) {
  atomicMax(reinterpret_cast<unsigned int*>(slot), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_nonneg(void* slot, double v) {
  atomicMax(reinterpret_cast<unsigned long long*>(slot), (unsigned long long)__double_as_longlong(v));
}

// `umax` (may be null): kUmaxSlots slots that receive the maximum of |out_x| + |out_y| + |out_z| over the
// interior cells this launch computes (the time-step rule's max |u|_1, flow_simulators_mpi_3d.py:426-449),
// one atomic per wave, spread over the slots.
template <typename T>
__global__ void __launch_bounds__(kTileThreads) k_curl_tiled(T* __restrict__ out, const T* __restrict__ f, T pref,
                                                            Geo G, int zchunk, TileShape S, Const3<T> add,
                                                            T* __restrict__ umax, int chunk0, int chunk_step) {
  extern __shared__ char smi_tile_smem[];
  T* lds = reinterpret_cast<T*>(smi_tile_smem);
  const Blk B = xcd_block<true>();
  const TileLane L = tile_lane(G, S, B);
  const int comp = (S.rows + 2) * S.pitch;
  bool core[4];
  {
    const bool yc = L.own && in_core_y(G, L.y);
#pragma unroll
    for (int e = 0; e < 4; ++e) core[e] = yc && in_core_x(G, L.x0 + e);
  }
  // chunk of this workgroup: chunk0 + chunk_step * (position along grid y), as in k_advect_diffuse
  const int z0 = (chunk0 + chunk_step * B.y) * zchunk, z1 = min(z0 + zchunk, G.pz);
  const int zs = max(z0, G.zlo_c), ze = min(z1 - 1, G.zhi_c);
  const bool run = zs <= ze;  // uniform over the workgroup
  Pack<T, 4> band[3];  // value of a cell outside the compute set
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    band[0].v[e] = add.x;
    band[1].v[e] = add.y;
    band[2].v[e] = add.z;
  }
  for (int z = z0; z < (run ? zs : z1); ++z) {
    if (L.own) {
#pragma unroll
      for (int k = 0; k < 3; ++k) st<T, 4>(out + k * G.vol + z * G.plane + L.q, band[k]);
    }
  }
  if (!run) return;
  const T* fo = f + L.q;
  const T* fh = f + L.hq;
  T lmax = T(0);
  bool inner[4];  // interior cells of this lane's row (ghost rows and columns do not count)
  {
    const bool yi = L.own && L.y >= G.g && L.y <= G.py - G.g - 1;
#pragma unroll
    for (int e = 0; e < 4; ++e) inner[e] = yi && L.x0 + e >= G.g && L.x0 + e <= G.px - G.g - 1;
  }
  auto park = [&](int buf, int off, const Vec3<T, 4>& v) {
    T* t = lds + buf * 3 * comp + off;
    st<T, 4>(t, v.x);
    st<T, 4>(t + comp, v.y);
    st<T, 4>(t + 2 * comp, v.z);
  };
  // register rotation by argument order, as in k_diffuse_tiled
  Vec3<T, 4> pa = ld3<T, 4>(fo + (zs - 1) * G.plane, G.vol), pb = ld3<T, 4>(fo + zs * G.plane, G.vol);
  Vec3<T, 4> pc = ld3<T, 4>(fo + (zs + 1) * G.plane, G.vol), pd;
  Vec3<T, 4> ha = ld3<T, 4>(fh + (zs + 1) * G.plane, G.vol), hb;
  park(zs & 1, L.own_off, pb);
  park(zs & 1, L.h_off, ld3<T, 4>(fh + zs * G.plane, G.vol));
  auto step = [&](int z, const Vec3<T, 4>& prev, const Vec3<T, 4>& cur, const Vec3<T, 4>& nxt, Vec3<T, 4>& nxt2,
                  const Vec3<T, 4>& h1, Vec3<T, 4>& h2) {
    const long long zn = (long long)min(z + 2, G.pz - 1) * G.plane;
    nxt2 = ld3<T, 4>(fo + zn, G.vol);  // in flight across this step
    h2 = ld3<T, 4>(fh + zn, G.vol);
    __syncthreads();
    const int buf = z & 1;
    const T* tx = lds + buf * 3 * comp + L.own_off;
    const T* ty = tx + comp;
    const T* tz = tx + 2 * comp;
    const Pack<T, 4> fx_ym = ld<T, 4>(tx - S.pitch), fx_yp = ld<T, 4>(tx + S.pitch);
    const Pack<T, 4> fz_ym = ld<T, 4>(tz - S.pitch), fz_yp = ld<T, 4>(tz + S.pitch);
    const T fy_xm = ty[-1], fy_xp = ty[4], fz_xm = tz[-1], fz_xp = tz[4];
    Pack<T, 4> rx, ry, rz;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const T fyL = e == 0 ? fy_xm : cur.y.v[e - 1], fyR = e == 3 ? fy_xp : cur.y.v[e + 1];
      const T fzL = e == 0 ? fz_xm : cur.z.v[e - 1], fzR = e == 3 ? fz_xp : cur.z.v[e + 1];
      const T cx = pref * (fz_yp.v[e] - fz_ym.v[e] - nxt.y.v[e] + prev.y.v[e]);
      const T cy = pref * (nxt.x.v[e] - prev.x.v[e] - fzR + fzL);
      const T cz = pref * (fyR - fyL - fx_yp.v[e] + fx_ym.v[e]);
      rx.v[e] = (core[e] ? cx : T(0)) + add.x;
      ry.v[e] = (core[e] ? cy : T(0)) + add.y;
      rz.v[e] = (core[e] ? cz : T(0)) + add.z;
    }
    if (umax && z >= G.g && z <= G.pz - G.g - 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const T s1 = fabs(rx.v[e]) + fabs(ry.v[e]) + fabs(rz.v[e]);
        lmax = inner[e] ? max(lmax, s1) : lmax;
      }
    }
    if (L.own) {
      const long long zo = z * G.plane + L.q;
      st<T, 4>(out + zo, rx);
      st<T, 4>(out + G.vol + zo, ry);
      st<T, 4>(out + 2 * G.vol + zo, rz);
    }
    park(buf ^ 1, L.own_off, nxt);
    park(buf ^ 1, L.h_off, h1);
  };
  for (int z = zs; z <= ze; z += 4) {
    step(z, pa, pb, pc, pd, ha, hb);
    if (z + 1 > ze) break;
    step(z + 1, pb, pc, pd, pa, hb, ha);
    if (z + 2 > ze) break;
    step(z + 2, pc, pd, pa, pb, ha, hb);
    if (z + 3 > ze) break;
    step(z + 3, pd, pa, pb, pc, hb, ha);
  }
  for (int z = ze + 1; z < z1; ++z) {
    if (L.own) {
#pragma unroll
      for (int k = 0; k < 3; ++k) st<T, 4>(out + k * G.vol + z * G.plane + L.q, band[k]);
    }
  }
  if (umax) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) lmax = max(lmax, __shfl_xor(lmax, off, 64));
    if ((threadIdx.x & 63) == 0 && lmax > T(0)) {
      const unsigned wave = (blockIdx.x + gridDim.x * blockIdx.y) * (blockDim.x >> 6) + (threadIdx.x >> 6);
      atomic_max_nonneg(umax + (wave & (kUmaxSlots - 1)), lmax);
    }
  }
}

template <typename T>
static TileShape tile_shape(const smi_ctx* c, int ty_cap = 8) {
  TileShape S;
  const int lanes = c->px / 4;
  S.ntx = (lanes + 31) / 32;
  S.txl = (lanes + S.ntx - 1) / S.ntx;
  S.ty = kTileThreads / S.txl;
  if (S.ty > ty_cap) S.ty = ty_cap;  // four-wave workgroups: several fit on a CU and hide each other's barriers
  S.nty = (c->py + S.ty - 1) / S.ty;
  S.pitch = 4 * S.txl + 8;
  S.rows = S.ty + 2;
  return S;
}

// ---------------------------------------------------------------- conservative advection flux
// flux = c * sum over axes of (F(i+1/2) - F(i-1/2)) on W, 0 elsewhere, with the face flux
//   F(i+1/2) = v_f * (v_f > 0 ? (-f[i-1] + 5 f[i] + 2 f[i+1]) / 6 : (2 f[i] + 5 f[i+1] - f[i+2]) / 6),
//   v_f = (v[i] + v[i+1]) / 2
// (third-order upwind-biased reconstruction, kernel support 2; the body of the reference's
// ENO3 flux lives in a package that is not part of its tree, so this follows the CPU
// restatement it is tested against).  W is what the seven boxes of advection_flux_mpi_3d.py
// :39-193 cover once two layers are taken off each: with core(a) = a in [g+2, p-g-3],
// face(a) = a in [g, g+1] or [p-g-2, p-g-1], ext(a) = a in [2, p-3]:
//   W = { z in [g, pz-g-1], core(y), core(x) } + { ext(z), face(y), core(x) } + { ext(z), ext(y), face(x) }.
template <typename T>
__device__ __forceinline__ T upwind3_face(const T* f, const T* v, long long stride) {
  // face between the cell at f / v and the next one along stride
  const T vf = T(0.5) * (v[0] + v[stride]);
  const T left = T(-1.0 / 6) * f[-stride] + T(5.0 / 6) * f[0] + T(1.0 / 3) * f[stride];
  const T right = T(1.0 / 3) * f[0] + T(5.0 / 6) * f[stride] + T(-1.0 / 6) * f[2 * stride];
  return vf * (vf > T(0) ? left : right);
}

// KEEP: cells outside W keep their value (the stand-alone flux operator writes W only).
template <typename T, bool KEEP = false>
__global__ void __launch_bounds__(kBlock) k_advection_flux(T* __restrict__ flux, const T* __restrict__ f,
                                                          const T* __restrict__ vel, T c, Geo G) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G.vol) return;
  const int z = (int)(i / G.plane);
  const long long r = i - (long long)z * G.plane;
  const int y = (int)(r / G.px), x = (int)(r - (long long)y * G.px);
  auto core = [&](int a, int p) { return a >= G.g + 2 && a <= p - G.g - 3; };
  auto face = [&](int a, int p) { return (a >= G.g && a <= G.g + 1) || (a >= p - G.g - 2 && a <= p - G.g - 1); };
  auto ext = [&](int a, int p) { return a >= 2 && a <= p - 3; };
  const bool in_w = (z >= G.g && z <= G.pz - G.g - 1 && core(y, G.py) && core(x, G.px)) ||
                    (ext(z, G.pz) && face(y, G.py) && core(x, G.px)) ||
                    (ext(z, G.pz) && ext(y, G.py) && face(x, G.px));
  T val = T(0);
  if (in_w) {
    const T* fc = f + i;
    const T* vx = vel + i;
    const T* vy = vel + G.vol + i;
    const T* vz = vel + 2 * G.vol + i;
    const T dx = upwind3_face(fc, vx, 1) - upwind3_face(fc - 1, vx - 1, 1);
    const T dy = upwind3_face(fc, vy, (long long)G.px) - upwind3_face(fc - G.px, vy - G.px, (long long)G.px);
    const T dz = upwind3_face(fc, vz, G.plane) - upwind3_face(fc - G.plane, vz - G.plane, G.plane);
    val = c * (dx + dy + dz);
  }
  if (!KEEP || in_w) flux[i] = val;
}

// ---------------------------------------------------------------- divergence (diagnostic)
// div = (0.5 * inv_dx) * (dfx/dx + dfy/dy + dfz/dz central differences) on D, 0 elsewhere.
template <typename T>
__global__ void __launch_bounds__(kBlock) k_divergence(T* __restrict__ div, const T* __restrict__ f, T half_inv_dx,
                                                       Geo G) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G.vol) return;
  int z = (int)(i / G.plane);
  long long r = i - (long long)z * G.plane;
  int y = (int)(r / G.px), x = (int)(r - (long long)y * G.px);
  bool ok = z >= G.zlo_c && z <= G.zhi_c && in_core_y(G, y) && in_core_x(G, x);
  T v = T(0);
  if (ok) {
    const T* fx = f + i;
    const T* fy = f + G.vol + i;
    const T* fz = f + 2 * G.vol + i;
    v = half_inv_dx * (fx[1] - fx[-1] + fy[G.px] - fy[-G.px] + fz[G.plane] - fz[-G.plane]);
  }
  div[i] = v;
}

// ---------------------------------------------------------------- Laplacian filter pass
// flux = 0.25 * (-f(+1) - f(-1) + 2 f) along AXIS (0 = x, 1 = y, 2 = z) on D, 0 elsewhere
// (laplacian_filter_mpi_3d.py:62-99 followed by the band clearing of :110-143).
template <typename T, int AXIS>
__global__ void __launch_bounds__(kBlock) k_filter_pass(T* __restrict__ flux, const T* __restrict__ f, Geo G) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G.vol) return;
  int z = (int)(i / G.plane);
  long long r = i - (long long)z * G.plane;
  int y = (int)(r / G.px), x = (int)(r - (long long)y * G.px);
  bool ok = z >= G.zlo_c && z <= G.zhi_c && in_core_y(G, y) && in_core_x(G, x);
  T v = T(0);
  if (ok) {
    const long long d = AXIS == 0 ? 1 : (AXIS == 1 ? (long long)G.px : G.plane);
    v = T(0.25) * (-f[i + d] - f[i - d] + T(2) * f[i]);
  }
  flux[i] = v;
}

// One-pass form of the order-1 multiplicative filter: out = f - Lz(Ly(Lx f)), where each
// directional pass L_a g = 0.25 (-g(+1) - g(-1) + 2 g) is taken on D and is 0 elsewhere, exactly
// as the three passes above leave it (so the intermediate fields vanish outside D when the next
// pass reads them).  F2 = Ly(Lx f) of a plane only needs f of that plane, so the kernel marches
// along z with F2 of the planes behind, at and ahead in registers.  Planes (and, in a pencil, rows) one past a rank
// boundary count as inside for the intermediate fields (there the neighbour rank computes them
// from the same exchanged values); [zlo_e, zhi_e] and [ylo_e, yhi_e] are those extended ranges.
template <typename T, int V>
__global__ void __launch_bounds__(kBlock) k_filter_fused(T* __restrict__ out, const T* __restrict__ f, Geo G,
                                                        int zchunk, int zlo_e, int zhi_e, PenX<T> X) {
  const Blk B = xcd_block<false>();
  long long q;
  int y, x0;
  if (!lane_pos<V>(G, B, q, y, x0)) return;
  f += (long long)B.z * G.vol + q;
  out += (long long)B.z * G.vol + q;
  // x bands penalised on the way out (V = 4 only): every store of this kernel goes through put()
  const int band = V == 4 ? ((X.on_lo && x0 == 0) ? 1 : ((X.on_hi && x0 == G.px - 4) ? 2 : 0)) : 0;
  auto put = [&](int z, Pack<T, V> o) {
    if constexpr (V == 4) pen_x_apply(o, band, X);
    st<T, V>(out + z * G.plane, o);
  };
  bool xc[V], anyx = false;
#pragma unroll
  for (int e = 0; e < V; ++e) {
    xc[e] = in_core_x(G, x0 + e);
    anyx |= xc[e];
  }
  // rows of D, and the rows whose Lx the y pass may read: D plus the ghost row across an inner y face of a pencil
  const bool yc = in_core_y(G, y);
  const int ylo_e = G.ylo_c - (G.yf_lo >= 0 ? 1 : 0), yhi_e = G.yhi_c + (G.yf_hi >= 0 ? 1 : 0);
  const bool ycm = y - 1 >= ylo_e && y - 1 <= yhi_e, ycp = y + 1 >= ylo_e && y + 1 <= yhi_e;
  const int z0 = B.y * zchunk, z1 = min(z0 + zchunk, G.pz);
  const int zs = max(z0, G.zlo_c), ze = min(z1 - 1, G.zhi_c);
  if (!(yc && anyx) || zs > ze) {
    for (int z = z0; z < z1; ++z) put(z, ld<T, V>(f + z * G.plane));
    return;
  }
  // Lx f of one row of plane zp (masked to the row's cells inside D)
  auto lx_row = [&](const T* row, bool row_in_core) {
    Pack<T, V> r = zero_pack<T, V>();
    if (!row_in_core) return r;
    const Pack<T, V> c = ld<T, V>(row);
    const T xm = row[-1], xp = row[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const T left = e == 0 ? xm : c.v[e - 1], right = e == V - 1 ? xp : c.v[e + 1];
      r.v[e] = xc[e] ? T(0.25) * (-right - left + T(2) * c.v[e]) : T(0);
    }
    return r;
  };
  // F2 = Ly(Lx f) of this lane's cells in plane zp
  auto f2_plane = [&](int zp) {
    Pack<T, V> r = zero_pack<T, V>();
    if (zp < zlo_e || zp > zhi_e) return r;
    const T* pc = f + zp * G.plane;
    const Pack<T, V> a = lx_row(pc - G.px, ycm), b = lx_row(pc, true), c = lx_row(pc + G.px, ycp);
#pragma unroll
    for (int e = 0; e < V; ++e) r.v[e] = xc[e] ? T(0.25) * (-c.v[e] - a.v[e] + T(2) * b.v[e]) : T(0);
    return r;
  };
  for (int z = z0; z < zs; ++z) put(z, ld<T, V>(f + z * G.plane));
  Pack<T, V> prev = f2_plane(zs - 1), cur = f2_plane(zs);
  for (int z = zs; z <= ze; ++z) {
    const Pack<T, V> nxt = f2_plane(z + 1);
    const Pack<T, V> fc = ld<T, V>(f + z * G.plane);
    Pack<T, V> o;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const T flux = xc[e] ? T(0.25) * (-nxt.v[e] - prev.v[e] + T(2) * cur.v[e]) : T(0);
      o.v[e] = fc.v[e] - flux;
    }
    put(z, o);
    prev = cur;
    cur = nxt;
  }
  for (int z = ze + 1; z < z1; ++z) put(z, ld<T, V>(f + z * G.plane));
}

// k_filter_fused on the 16-byte path with the LDS tiling of k_diffuse_tiled: the plane whose F2 = Ly(Lx f) is due sits
// in one of two LDS tiles (own cells and a ring of one cell, corners included: F2 reaches diagonally), F2 of the
// planes behind, at and ahead and f of the plane being written stay in registers.  The non-tiled form reads three
// rows and six scalars per plane and lane through the L1; this one reads each cell of the plane once per tile.
// The arithmetic, masks and their order are those of k_filter_fused.  Grid z = component.
template <typename T>
__global__ void __launch_bounds__(kTileThreads) k_filter_tiled(T* __restrict__ out, const T* __restrict__ f, Geo G,
                                                              int zchunk, TileShape S, int zlo_e, int zhi_e,
                                                              PenX<T> X) {
  extern __shared__ char smi_tile_smem[];
  T* lds = reinterpret_cast<T*>(smi_tile_smem);
  const Blk B = xcd_block<true>();
  const TileLane L = tile_lane(G, S, B, true);
  out += (long long)B.z * G.vol;
  f += (long long)B.z * G.vol;
  const int comp = (S.rows + 2) * S.pitch;
  bool xc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) xc[e] = in_core_x(G, L.x0 + e);
  const bool yc = L.own && in_core_y(G, L.y);
  const int ylo_e = G.ylo_c - (G.yf_lo >= 0 ? 1 : 0), yhi_e = G.yhi_c + (G.yf_hi >= 0 ? 1 : 0);
  const bool ycm = L.y - 1 >= ylo_e && L.y - 1 <= yhi_e, ycp = L.y + 1 >= ylo_e && L.y + 1 <= yhi_e;
  const int band = (X.on_lo && L.x0 == 0) ? 1 : ((X.on_hi && L.x0 == G.px - 4) ? 2 : 0);
  auto put = [&](int z, Pack<T, 4> o) {
    pen_x_apply(o, band, X);
    if (L.own) st<T, 4>(out + z * G.plane + L.q, o);
  };
  const int z0 = B.y * zchunk, z1 = min(z0 + zchunk, G.pz);
  const int zs = max(z0, G.zlo_c), ze = min(z1 - 1, G.zhi_c);
  const bool run = zs <= ze;  // uniform over the workgroup
  const T* fo = f + L.q;
  const T* fh = f + L.hq;
  for (int z = z0; z < (run ? zs : z1); ++z) put(z, ld<T, 4>(fo + z * G.plane));
  if (!run) return;
  // Lx of the row at `row` of the LDS tile (masked to the row's cells inside D), then F2 of the plane in tile `buf`
  auto lx_row = [&](const T* row, bool row_in_core) {
    Pack<T, 4> r = zero_pack<T, 4>();
    if (!row_in_core) return r;
    const Pack<T, 4> c = ld<T, 4>(row);
    const T xm = row[-1], xp = row[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const T left = e == 0 ? xm : c.v[e - 1], right = e == 3 ? xp : c.v[e + 1];
      r.v[e] = xc[e] ? T(0.25) * (-right - left + T(2) * c.v[e]) : T(0);
    }
    return r;
  };
  auto f2_tile = [&](int buf, int zp) {
    Pack<T, 4> r = zero_pack<T, 4>();
    if (zp < zlo_e || zp > zhi_e) return r;  // uniform
    const T* tc = lds + buf * comp + L.own_off;
    const Pack<T, 4> a = lx_row(tc - S.pitch, ycm), b = lx_row(tc, true), c = lx_row(tc + S.pitch, ycp);
#pragma unroll
    for (int e = 0; e < 4; ++e) r.v[e] = xc[e] ? T(0.25) * (-c.v[e] - a.v[e] + T(2) * b.v[e]) : T(0);
    return r;
  };
  // The tile holds plane w = z + 1 while plane z is written: w runs from zs - 1 (two steps that only form F2) to
  // ze + 1.  Planes rotate through four register sets and the halo through two by the argument order of step(), as
  // in k_diffuse_tiled; the loads of plane w + 2 are issued at the top of step w.
  auto plane = [&](int z) { return (long long)min(max(z, 0), G.pz - 1) * G.plane; };
  const int w0 = zs - 1;
  Pack<T, 4> pa = ld<T, 4>(fo + plane(w0 - 1)), pb = ld<T, 4>(fo + plane(w0));
  Pack<T, 4> pc = ld<T, 4>(fo + plane(w0 + 1)), pd, ha = ld<T, 4>(fh + plane(w0 + 1)), hb;
  st<T, 4>(lds + (w0 & 1) * comp + L.own_off, pb);
  st<T, 4>(lds + (w0 & 1) * comp + L.h_off, ld<T, 4>(fh + plane(w0)));
  Pack<T, 4> f2p = zero_pack<T, 4>(), f2c = zero_pack<T, 4>();
  auto step = [&](int w, const Pack<T, 4>& prev, const Pack<T, 4>& nxt, Pack<T, 4>& nxt2, const Pack<T, 4>& h1,
                  Pack<T, 4>& h2) {
    nxt2 = ld<T, 4>(fo + plane(w + 2));  // in flight across this step
    h2 = ld<T, 4>(fh + plane(w + 2));
    __syncthreads();
    const int buf = w & 1;
    const Pack<T, 4> f2n = f2_tile(buf, w);
    const int z = w - 1;
    if (z >= zs) {  // uniform
      Pack<T, 4> o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const T flux = (yc && xc[e]) ? T(0.25) * (-f2n.v[e] - f2p.v[e] + T(2) * f2c.v[e]) : T(0);
        o.v[e] = prev.v[e] - flux;
      }
      put(z, o);
    }
    f2p = f2c;
    f2c = f2n;
    st<T, 4>(lds + (buf ^ 1) * comp + L.own_off, nxt);
    st<T, 4>(lds + (buf ^ 1) * comp + L.h_off, h1);
  };
  for (int w = w0; w <= ze + 1; w += 4) {
    step(w, pa, pc, pd, ha, hb);
    if (w + 1 > ze + 1) break;
    step(w + 1, pb, pd, pa, hb, ha);
    if (w + 2 > ze + 1) break;
    step(w + 2, pc, pa, pb, ha, hb);
    if (w + 3 > ze + 1) break;
    step(w + 3, pd, pb, pc, hb, ha);
  }
  for (int z = ze + 1; z < z1; ++z) put(z, ld<T, 4>(fo + z * G.plane));
}

// zero the outer (g + width) layers next to physical faces (zero_bands of the wrappers)
template <typename T>
__global__ void __launch_bounds__(kBlock) k_zero_bands(T* __restrict__ f, Geo G, int w, int zlo, int zhi, int ylo,
                                                      int yhi, int xlo = 1, int xhi = 1) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G.vol) return;
  int z = (int)(i / G.plane);
  long long r = i - (long long)z * G.plane;
  int y = (int)(r / G.px), x = (int)(r - (long long)y * G.px);
  bool band = (xlo && x < w) || (xhi && x >= G.px - w) || (ylo && y < w) || (yhi && y >= G.py - w) ||
              (zlo && z < w) || (zhi && z >= G.pz - w);
  if (band) f[i] = T(0);
}

// ---------------------------------------------------------------- penalise boundary
struct PenFac {
  double lo[3][16];  // [axis x,y,z][band index from the outermost cell inward]
  double hi[3][16];  // [axis][band index from the first cell of the high band outward]
};

// AXIS 0: one lane per (comp, z, y) line; AXIS 1: per (comp, z, x); AXIS 2: per (comp, y, x).
// Copies layer w-1 (resp. p-w) across the band, then multiplies by the sine factor
// (penalise_field_boundary_mpi_3d.py:196-242).
template <typename T, int AXIS>
__global__ void __launch_bounds__(kBlock) k_penalise(T* __restrict__ f, Geo G, int ncomp, int w, PenFac fac,
                                                     int do_lo, int do_hi) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long lines = AXIS == 0 ? (long long)G.pz * G.py : (AXIS == 1 ? (long long)G.pz * G.px : G.plane);
  if (i >= lines * ncomp) return;
  int comp = (int)(i / lines);
  long long l = i - comp * lines;
  T* base;
  long long stride;
  int p;
  if (AXIS == 0) {
    base = f + comp * G.vol + l * G.px;
    stride = 1;
    p = G.px;
  } else if (AXIS == 1) {
    long long z = l / G.px, x = l - z * G.px;
    base = f + comp * G.vol + z * G.plane + x;
    stride = G.px;
    p = G.py;
  } else {
    base = f + comp * G.vol + l;
    stride = G.plane;
    p = G.pz;
  }
  if (do_lo) {
    T src = base[(w - 1) * stride];
    for (int k = 0; k < w; ++k) base[k * stride] = src * (T)fac.lo[AXIS][k];
  }
  if (do_hi) {
    T src = base[(p - w) * stride];
    for (int k = 0; k < w; ++k) base[(p - w + k) * stride] = src * (T)fac.hi[AXIS][k];
  }
}

template <typename T>
bool vec_ok(const smi_ctx* c, const void* a, const void* b) {
  return c->px % 4 == 0 && ((uintptr_t)a % (4 * sizeof(T))) == 0 && ((uintptr_t)b % (4 * sizeof(T))) == 0 &&
         (c->vol % 4) == 0;
}

int blocks_for(long long n) { return (int)((n + kBlock - 1) / kBlock); }

#ifndef SMI_TILE_ZCHUNK
#define SMI_TILE_ZCHUNK 16
#endif
#ifndef SMI_TILE_MIN_WGS
#define SMI_TILE_MIN_WGS 4096
#endif
#ifndef SMI_TILE_ROWS
#define SMI_TILE_ROWS 8
#endif
// Planes per workgroup of a tiled kernel.  Each chunk re-reads two halo planes, so chunks are
// SMI_TILE_ZCHUNK planes long unless that leaves the chip (1024 - 2048 resident workgroups)
// with fewer than a few rounds of work: then they are halved once (256^3: advection 0.195 -> 0.163 ms).
static int tiled_zchunk(const smi_ctx* c, int tiles_per_plane_set) {
  int zchunk = SMI_TILE_ZCHUNK;
  while (zchunk > 8 && (long long)tiles_per_plane_set * ((c->pz + zchunk - 1) / zchunk) < SMI_TILE_MIN_WGS) zchunk /= 2;
  return c->pz < zchunk ? c->pz : zchunk;
}

// launch geometry of the LDS-tiled stencil kernels (ncomp_lds component tiles, two buffers)
struct TiledLaunch {
  TileShape S;
  dim3 grid;
  int threads, zchunk;
  size_t lds_bytes;
};
template <typename T>
TiledLaunch tiled_launch(const smi_ctx* c, int ncomp_lds, int grid_z) {
  TiledLaunch t;
  t.S = tile_shape<T>(c, SMI_TILE_ROWS);
  t.zchunk = tiled_zchunk(c, t.S.ntx * t.S.nty * grid_z);
  t.grid = dim3(t.S.ntx * t.S.nty, (c->pz + t.zchunk - 1) / t.zchunk, grid_z);
  t.threads = (t.S.txl * t.S.ty + 63) / 64 * 64;  // whole waves covering the tile and its halo duty
  t.lds_bytes = (size_t)2 * ncomp_lds * (t.S.rows + 2) * t.S.pitch * sizeof(T);  // two spare rows per tile
  return t;
}
// The tiled kernels may need more than the default 64 KB of dynamic LDS (double precision).  The
// attribute belongs to the current device's copy of the function and costs next to nothing to set,
// so it is set before every launch rather than cached per process.
template <typename F>
int allow_lds(F fn) {
  SMI_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  return 0;
}

}  // namespace

// ------------------------------------------------------------------ C ABI
extern "C" int smi_diffusion_flux(smi_ctx* c, void* flux, const void* field, double prefactor, int ncomp) {
  SMI_REQUIRE(c && flux && field && ncomp >= 1, "smi_diffusion_flux: bad argument");
  SMI_REQUIRE(flux != field, "smi_diffusion_flux: flux and field must not alias");
  if (smi_multi_rank(c)) SMI_CHECK(smi_halo_exchange_impl(c, (void*)field, ncomp));
  Geo G = make_geo(c);
  SmiProfScope prof(c, PROF_DIFFUSION_FLUX, 2.0 * ncomp * smi_S(c));
  SMI_DISPATCH(c, {
    if (vec_ok<T>(c, flux, field)) {
      const TiledLaunch t = tiled_launch<T>(c, 1, ncomp);
      SMI_CHECK(allow_lds(k_diffuse_tiled<T, false>));
      k_diffuse_tiled<T, false><<<t.grid, t.threads, t.lds_bytes, c->stream>>>((T*)flux, (const T*)field, (T)prefactor,
                                                                              G, t.zchunk, t.S);
    } else {
      LaunchShape s = plan<1>(c, ncomp);
      k_diffusion_flux<T, 1><<<s.grid, kBlock, 0, c->stream>>>((T*)flux, (const T*)field, (T)prefactor, G, s.zchunk);
    }
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int smi_diffusion_timestep(smi_ctx* c, void* field, void* flux, double nu_dt_by_dx2, int ncomp) {
  SMI_REQUIRE(c && flux && field && ncomp >= 1, "smi_diffusion_timestep: bad argument");
  for (int k = 0; k < ncomp; ++k) {
    void* fk = (char*)field + (size_t)k * c->vol * smi_elem_size(c);
    SMI_CHECK(smi_diffusion_flux(c, flux, fk, nu_dt_by_dx2, 1));
    SMI_CHECK(smi_elementwise_sum(c, fk, fk, flux, 1));
  }
  return 0;
}

// One-pass diffusion timestep into another array: out = in + c * laplacian(in) on D, in elsewhere.
int smi_diffusion_apply_impl(smi_ctx* c, void* out, void* in, double prefactor, int ncomp) {
  SMI_REQUIRE(out != in, "smi_diffusion_apply: output and input must not alias");
  if (smi_multi_rank(c)) SMI_CHECK(smi_halo_exchange_impl(c, in, ncomp));
  Geo G = make_geo(c);
  SmiProfScope prof(c, PROF_FUSED_DIFFUSE, 2.0 * ncomp * smi_S(c));
  SMI_DISPATCH(c, {
    if (vec_ok<T>(c, out, in)) {
      const TiledLaunch t = tiled_launch<T>(c, 1, ncomp);
      SMI_CHECK(allow_lds(k_diffuse_tiled<T, true>));
      k_diffuse_tiled<T, true><<<t.grid, t.threads, t.lds_bytes, c->stream>>>((T*)out, (const T*)in, (T)prefactor, G,
                                                                             t.zchunk, t.S);
    } else {
      LaunchShape s = plan<1>(c, ncomp);
      k_diffusion_flux<T, 1, true><<<s.grid, kBlock, 0, c->stream>>>((T*)out, (const T*)in, (T)prefactor, G, s.zchunk);
    }
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

// `add` (MODE 0 only, may be null): constant vector added to the whole output array, i.e. the
// add_fixed_val that follows the curl in compute_flow_velocity (flow_simulators_mpi_3d.py:382-393),
// folded into the store of the tiled kernel; the other paths run it as a second pass.
template <int MODE>
static int run_curl(smi_ctx* c, void* out, void* f, double prefactor, const double* add = nullptr,
                    bool track_max = false) {
  SMI_REQUIRE(out != f, "curl operators: output and input must not alias");
  Geo G = make_geo(c);
  bool tiled = false;
  SMI_DISPATCH(c, { tiled = MODE == 0 && vec_ok<T>(c, out, f); });
  // Across z-slabs over RCCL the tiled curl runs the way the one-pass advection + diffusion kernel does: the ghost
  // planes of `f` travel on the communication stream while the z chunks that cannot reach them are computed, and
  // the chunks at the two ends of the slab follow behind the exchange (curl_mpi_3d.py:47-60: interior before halo).
  // A chunk reads planes z0 - 1 .. z1, so the chunks out of reach are those with z0 >= g + 1 and z1 <= pz - g - 1.
  TiledLaunch tl{};
  int n_lo = 0, n_hi = 0, nchunks = 0;
  bool overlap = false;
  if (tiled) {
    SMI_DISPATCH(c, { tl = tiled_launch<T>(c, 3, 1); });
    nchunks = (int)tl.grid.y;
    while (n_lo < nchunks && n_lo * tl.zchunk < c->g + 1) ++n_lo;
    while (n_hi < nchunks - n_lo && (nchunks - n_hi) * tl.zchunk > c->pz - c->g - 1) ++n_hi;
    overlap = smi_multi_rank(c) && !c->exchange_cb && c->nranks_y == 1 && c->nranks_x == 1 && nchunks - n_lo - n_hi >= 1;
  }
  if (smi_multi_rank(c)) {
    if (overlap) {
      // no RCCL call may be issued on c->stream while this exchange is pending on the communication stream: the
      // launches below are plain kernels, and the stream waits for ev_b before anything else follows
      SMI_CHECK_HIP(hipEventRecord(c->ev_a, c->stream));
      SMI_CHECK_HIP(hipStreamWaitEvent(c->comm_stream, c->ev_a, 0));
      SMI_CHECK(smi_halo_exchange2_on(c, f, 3, nullptr, 0, c->comm_stream));
    } else {
      SMI_CHECK(smi_halo_exchange_impl(c, f, 3));
    }
  }
  bool added = add == nullptr;
  c->umax_valid = 0;
  if (track_max)  // a fill kernel, not hipMemsetAsync: the step may be captured into a graph by a caller
    SMI_CHECK(smi_memset_zero(c, c->umax_slots, sizeof(unsigned long long) * kUmaxSlots));
  {
    SmiProfScope prof(c, MODE == 0 ? PROF_CURL : PROF_CURL_UPDATE, (MODE == 0 ? 6.0 : 9.0) * smi_S(c));
    SMI_DISPATCH(c, {
    if (tiled) {
      const TiledLaunch& t = tl;
      SMI_CHECK(allow_lds(k_curl_tiled<T>));
      const Const3<T> a = add ? Const3<T>{(T)add[0], (T)add[1], (T)add[2]} : Const3<T>{T(0), T(0), T(0)};
      auto launch = [&](int chunk0, int count, hipStream_t stream) {
        if (count <= 0) return;
        k_curl_tiled<T><<<dim3(t.grid.x, count, 1), t.threads, t.lds_bytes, stream>>>(
            (T*)out, (const T*)f, (T)prefactor, G, t.zchunk, t.S, a, track_max ? (T*)c->umax_slots : nullptr, chunk0, 1);
      };
      if (overlap) {
        // the end chunks follow the exchange on its stream, beside the interior chunks (see smi_advect_diffuse_impl)
        // (they also follow what this stream has queued since the exchange began: the reset of the maximum slots)
        SMI_CHECK_HIP(hipEventRecord(c->ev_a, c->stream));
        SMI_CHECK_HIP(hipStreamWaitEvent(c->comm_stream, c->ev_a, 0));
        launch(n_lo, nchunks - n_lo - n_hi, c->stream);
        launch(0, n_lo, c->comm_stream);
        launch(nchunks - n_hi, n_hi, c->comm_stream);
        SMI_CHECK_HIP(hipEventRecord(c->ev_b, c->comm_stream));
        SMI_CHECK_HIP(hipStreamWaitEvent(c->stream, c->ev_b, 0));
      } else {
        launch(0, nchunks, c->stream);
      }
      added = true;
      if (track_max) {
        // cells outside the compute set (the zeroed bands) hold the constant alone
        c->umax_band = (double)(fabs(a.x) + fabs(a.y) + fabs(a.z));
        c->umax_valid = 1;
        c->umax_field = out;
      }
    } else if (vec_ok<T>(c, out, f)) {
      LaunchShape s = plan<4>(c, 1);
      k_curl<T, 4, MODE><<<s.grid, kBlock, 0, c->stream>>>((T*)out, (const T*)f, (T)prefactor, G, s.zchunk,
                                                           CurlRange{0, G.plane, 0, G.pz});
    } else {
      LaunchShape s = plan<1>(c, 1);
      k_curl<T, 1, MODE><<<s.grid, kBlock, 0, c->stream>>>((T*)out, (const T*)f, (T)prefactor, G, s.zchunk,
                                                           CurlRange{0, G.plane, 0, G.pz});
    }
    });
    SMI_LAUNCH_CHECK();
  }
  if (!added) SMI_CHECK(smi_add_fixed_val(c, out, out, 3, add));
  return 0;
}

// u = prefactor * curl(field) + add on the whole array (add may be null)
int smi_curl_plus_const_impl(smi_ctx* c, void* curl, void* field, double prefactor, const double* add,
                             bool track_max) {
  return run_curl<0>(c, curl, field, prefactor, add, track_max);
}

namespace {
// rows [y0, y1] x planes [z0, z1] of every component := 0
template <typename T>
__global__ void __launch_bounds__(kBlock) k_zero_rows(T* __restrict__ f, Geo G, int y0, int nrows, int z0) {
  // grid.y = plane, grid.z = component; the rows of one plane are one contiguous run
  T* p = f + (long long)blockIdx.z * G.vol + (long long)(z0 + blockIdx.y) * G.plane + (long long)y0 * G.px;
  const long long n = (long long)nrows * G.px;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    p[i] = T(0);
}

// the support box of the forcing promised for this step, clipped to the array: false when it is empty
bool forcing_box(const smi_ctx* c, int margin, int (&lo)[3], int (&hi)[3]) {
  const int p[3] = {c->pz, c->py, c->px};
  for (int a = 0; a < 3; ++a) {
    lo[a] = c->fbox_lo[a] - margin < 0 ? 0 : c->fbox_lo[a] - margin;
    hi[a] = c->fbox_hi[a] + margin > p[a] - 1 ? p[a] - 1 : c->fbox_hi[a] + margin;
    if (c->fbox_lo[a] > c->fbox_hi[a] || lo[a] > hi[a]) return false;
  }
  return true;
}
}  // namespace

// w += p curl(f) where f is zero outside the promised box: only the cells within one cell of the box
// can change, so the kernel runs over those rows and planes only (whole rows: they are contiguous).
int smi_forcing_update_boxed_impl(smi_ctx* c, void* vorticity, void* forcing, double prefactor) {
  SMI_REQUIRE(c->fbox_valid && c->nranks == 1, "boxed forcing update without a support box");
  int lo[3], hi[3];
  if (!forcing_box(c, 1, lo, hi)) return 0;  // zero forcing: nothing to add
  Geo G = make_geo(c);
  bool ok = false;
  SMI_DISPATCH(c, { ok = vec_ok<T>(c, vorticity, forcing); });
  if (!ok) return smi_update_vorticity_from_velocity_forcing(c, vorticity, forcing, prefactor);
  const int nz = hi[0] - lo[0] + 1, nrows = hi[1] - lo[1] + 1;
  const long long lanes = (long long)nrows * G.px / 4;
  const int zchunk = nz < 16 ? nz : 16;
  dim3 grid((unsigned)((lanes + kBlock - 1) / kBlock), (unsigned)((nz + zchunk - 1) / zchunk), 1);
  const CurlRange R{(long long)lo[1] * G.px, (long long)(hi[1] + 1) * G.px, lo[0], hi[0] + 1};
  SmiProfScope prof(c, PROF_CURL_UPDATE, 9.0 * nz * nrows * G.px * smi_elem_size(c));
  SMI_DISPATCH(c, {
    k_curl<T, 4, 1><<<grid, kBlock, 0, c->stream>>>((T*)vorticity, (const T*)forcing, (T)prefactor, G, zchunk, R);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

int smi_forcing_zero_boxed_impl(smi_ctx* c, void* forcing) {
  SMI_REQUIRE(c->fbox_valid && c->nranks == 1, "boxed forcing reset without a support box");
  int lo[3], hi[3];
  if (!forcing_box(c, 0, lo, hi)) return 0;
  Geo G = make_geo(c);
  const int nz = hi[0] - lo[0] + 1, nrows = hi[1] - lo[1] + 1;
  const long long n = (long long)nrows * G.px;
  const int bx = (int)((n + kBlock - 1) / kBlock < 64 ? (n + kBlock - 1) / kBlock : 64);
  SMI_DISPATCH(c, {
    k_zero_rows<T><<<dim3(bx, nz, 3), kBlock, 0, c->stream>>>((T*)forcing, G, lo[1], nrows, lo[0]);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int smi_update_vorticity_from_velocity_forcing(smi_ctx* c, void* vorticity, void* forcing,
                                                          double prefactor) {
  SMI_REQUIRE(c && vorticity && forcing, "smi_update_vorticity_from_velocity_forcing: bad argument");
  return run_curl<1>(c, vorticity, forcing, prefactor);
}

// gen_update_vorticity_from_penalised_velocity_pyst_mpi_kernel_3d
// (update_vorticity_from_velocity_forcing_mpi_3d.py:181-330): vorticity += p * curl(penalised_velocity - velocity) on
// the update set of smi_update_vorticity_from_velocity_forcing, both inputs exchanged first.
extern "C" int smi_update_vorticity_from_penalised_velocity(smi_ctx* c, void* vorticity, void* penalised_velocity,
                                                            void* velocity, double prefactor) {
  SMI_REQUIRE(c && vorticity && penalised_velocity && velocity,
              "smi_update_vorticity_from_penalised_velocity: bad argument");
  SMI_REQUIRE(vorticity != penalised_velocity && vorticity != velocity,
              "smi_update_vorticity_from_penalised_velocity: the vorticity must not alias its inputs");
  if (smi_multi_rank(c)) SMI_CHECK(smi_halo_exchange2_impl(c, velocity, 3, penalised_velocity, 3));
  Geo G = make_geo(c);
  SmiProfScope prof(c, PROF_CURL_UPDATE, 12.0 * smi_S(c));
  SMI_DISPATCH(c, {
    const CurlRange R{0, G.plane, 0, G.pz};
    if (vec_ok<T>(c, vorticity, penalised_velocity) && vec_ok<T>(c, vorticity, velocity)) {
      LaunchShape s = plan<4>(c, 1);
      k_curl<T, 4, 1, true><<<s.grid, kBlock, 0, c->stream>>>((T*)vorticity, (const T*)penalised_velocity, (T)prefactor,
                                                             G, s.zchunk, R, (const T*)velocity);
    } else {
      LaunchShape s = plan<1>(c, 1);
      k_curl<T, 1, 1, true><<<s.grid, kBlock, 0, c->stream>>>((T*)vorticity, (const T*)penalised_velocity, (T)prefactor,
                                                             G, s.zchunk, R, (const T*)velocity);
    }
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- pointwise helpers of the penalisation method
// gen_brinkmann_penalise_pyst_mpi_kernel_3d / 2d (brinkmann_penalise_mpi_3d.py:7-19): implicit Brinkmann
// penalisation, penalised = (field + lambda chi penalty) / (1 + lambda chi), on every cell of the arrays (kernel
// support 0: no exchange, ghost cells included); n cells per component, ncomp components, chi one component.
// gen_char_func_from_level_set_via_sine_heaviside_pyst_mpi_kernel_3d / 2d (char_func_from_level_set_mpi_3d.py:8-28):
// chi = 0 below -w, 1 above w, (1 + phi / w + sin(pi phi / w) / pi) / 2 in between.
namespace {
template <typename T>
__global__ void __launch_bounds__(kBlock) k_brinkmann(T* __restrict__ out, const T* __restrict__ f,
                                                      const T* __restrict__ chi, const T* __restrict__ pen, T lambda,
                                                      long long n, int ncomp) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const T lc = lambda * chi[i];
  for (int k = 0; k < ncomp; ++k) out[k * n + i] = (f[k * n + i] + lc * pen[k * n + i]) / (T(1) + lc);
}
template <typename T>
__global__ void __launch_bounds__(kBlock) k_sine_heaviside(T* __restrict__ chi, const T* __restrict__ phi, T w,
                                                           long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const T p = phi[i];
  const T pi = (T)3.14159265358979323846;
  T v = T(0.5) * (T(1) + p / w + sin(pi * p / w) / pi);
  v = p < -w ? T(0) : (p > w ? T(1) : v);
  chi[i] = v;
}
}  // namespace

extern "C" int smi_brinkmann_penalise(smi_ctx* c, void* penalised_field, const void* field, const void* char_field,
                                      const void* penalty_field, double penalty_factor, int ncomp) {
  SMI_REQUIRE(c && penalised_field && field && char_field && penalty_field && ncomp >= 1 && ncomp <= 3,
              "smi_brinkmann_penalise: bad argument");
  const long long n = c->nz == 1 ? (long long)c->plane : (long long)c->vol;
  SMI_DISPATCH(c, {
    k_brinkmann<T><<<blocks_for(n), kBlock, 0, c->stream>>>((T*)penalised_field, (const T*)field, (const T*)char_field,
                                                           (const T*)penalty_field, (T)penalty_factor, n, ncomp);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int smi_char_func_from_level_set(smi_ctx* c, void* char_func_field, const void* level_set_field,
                                            double blend_width) {
  SMI_REQUIRE(c && char_func_field && level_set_field, "smi_char_func_from_level_set: bad argument");
  SMI_REQUIRE(blend_width > 0.0, "smi_char_func_from_level_set: the blend width must be positive");
  const long long n = c->nz == 1 ? (long long)c->plane : (long long)c->vol;
  SMI_DISPATCH(c, {
    k_sine_heaviside<T><<<blocks_for(n), kBlock, 0, c->stream>>>((T*)char_func_field, (const T*)level_set_field,
                                                                (T)blend_width, n);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

// out = vorticity + p * curl(velocity x vorticity) on the update set, vorticity elsewhere
int smi_advect_vorticity_impl(smi_ctx* c, void* out, void* velocity, void* vorticity, double prefactor) {
  SMI_REQUIRE(out != velocity && out != vorticity, "advection update: output and input must not alias");
  if (smi_multi_rank(c)) {
    // the ghost planes of u x w are the cross product of the exchanged ghost planes
    SMI_CHECK(smi_halo_exchange2_impl(c, velocity, 3, vorticity, 3));
  }
  Geo G = make_geo(c);
  SmiProfScope prof(c, PROF_FUSED_ADVECT, 9.0 * smi_S(c));
  SMI_DISPATCH(c, {
    if (vec_ok<T>(c, out, velocity) && vec_ok<T>(c, out, vorticity)) {
      const TileShape S = tile_shape<T>(c);
      const size_t lds_bytes = (size_t)2 * 3 * S.rows * S.pitch * sizeof(T);
      SMI_CHECK(allow_lds(k_advect_tiled<T>));
      // short z chunks (two halo planes per 16) measured faster than long ones: more, shorter workgroups
      const int zchunk = tiled_zchunk(c, S.ntx * S.nty);
      dim3 grid(S.ntx * S.nty, (c->pz + zchunk - 1) / zchunk, 1);
      const int threads = (S.txl * S.ty + 63) / 64 * 64;  // whole waves covering the tile and its halo duty
      k_advect_tiled<T><<<grid, threads, lds_bytes, c->stream>>>((T*)out, (const T*)velocity,
                                                                      (const T*)vorticity, (T)prefactor, G, zchunk, S);
    } else {
      LaunchShape s = plan<1, kAdvectBlock>(c, 1);
      k_advect<T, 1><<<s.grid, kAdvectBlock, 0, c->stream>>>((T*)out, (const T*)velocity, (const T*)vorticity,
                                                       (T)prefactor, G, s.zchunk);
    }
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

// Planes per workgroup of the fused kernel over a run of `planes` planes: one workgroup is resident per CU (its LDS
// tiles take two thirds of the CU's 160 KB), so the launch should fill whole rounds of the chip, and every chunk
// re-reads four halo planes.
static int fused_zchunk(const smi_ctx* c, int tiles_per_plane, int planes) {
  static int cus[64] = {};
  int& n_cu = cus[c->device & 63];
  if (n_cu == 0 && hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess) n_cu = 256;
  if (c->opt_fused_zchunk > 0) return c->opt_fused_zchunk < planes ? c->opt_fused_zchunk : planes;
  int best = planes;
  double best_eff = 0.0;
  for (int n = 1; n <= planes; ++n) {  // (down to one plane per chunk: small grids want workgroups, not long marches)
    const int zc = (planes + n - 1) / n;
    const long long wgs = (long long)tiles_per_plane * ((planes + zc - 1) / zc);
    const long long rounds = (wgs + n_cu - 1) / n_cu;
    const double eff = (double)wgs / (double)(rounds * n_cu) * zc / (zc + 4.0);
    if (eff > best_eff) {
      best_eff = eff;
      best = zc;
    }
  }
  return best;
}

// out = advection update followed by the diffusion timestep of `vorticity`, in one pass (k_advect_diffuse).
// *ran = 0 and nothing is launched when this context is not served by the fused kernel (rows that are not whole
// 4-element vectors; fewer than two ghost layers across ranks): the caller then runs the two operators one
// after the other.
// Across ranks the kernel needs u and w on BOTH ghost planes (and, in a pencil, ghost rows) next to a neighbour (the
// advected vorticity of the neighbour's edge plane or row is formed here from them, see Geo::za_lo and Geo::yf_lo),
// which one exchange of depth 2 delivers, in place
// of the two exchanges of depth 1 of the two-pass path.  That exchange runs on the communication stream while the
// chunks of planes that are out of reach of the ghost planes are already being computed; the first and the last
// chunk follow once it has arrived.
// penx (may be null): sine factors of the x bands (4 low, 4 high) of the boundary penalisation that follows, to be
// applied on the way out; only valid for ghost + zone width = 4.
int smi_advect_diffuse_impl(smi_ctx* c, void* out, void* velocity, void* vorticity, double pref_curl, double pref_diff,
                            int* ran, const double* penx) {
  *ran = 0;
  SMI_REQUIRE(out != velocity && out != vorticity, "advection-diffusion: output and input must not alias");
  if (c->opt_no_fused) return 0;
  if (smi_multi_rank(c) && c->g < 2) return 0;
  bool aligned = false;
  FusedShape S;
  bool shaped = false;
  SMI_DISPATCH(c, {
    aligned = vec_ok<T>(c, out, velocity) && vec_ok<T>(c, out, vorticity);
    shaped = fused_shape<T>(c, S);
  });
  if (!aligned || !shaped) return 0;
  Geo G = make_geo(c);
  const size_t lds_bytes = (size_t)6 * (S.ry + 2) * S.pitch * smi_elem_size(c);
  SMI_DISPATCH(c, {
    SMI_CHECK_HIP(hipFuncSetAttribute((const void*)k_advect_diffuse<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds_bytes));
  });
  // planes [zbase, zlimit) in `count` chunks of zchunk planes, the second and later ones zjump planes apart
  auto launch = [&](int zbase, int zjump, int zchunk, int zlimit, int count, hipStream_t stream) {
    dim3 grid(S.ntx * S.nty, count, 1);
    SMI_DISPATCH(c, {
      PenX<T> X{};
      if (penx) {
        X.on_lo = c->phys_xlo;
        X.on_hi = c->phys_xhi;
        for (int e = 0; e < 4; ++e) {
          X.lo[e] = (T)penx[e];
          X.hi[e] = (T)penx[4 + e];
        }
      }
      k_advect_diffuse<T><<<grid, fused_threads<T>(), lds_bytes, stream>>>(
          (T*)out, (const T*)velocity, (const T*)vorticity, (T)pref_curl, (T)pref_diff, G, zchunk, S, zbase, zjump,
          zlimit, X);
    });
  };
  auto launch_all = [&](hipStream_t stream) {
    const int zchunk = fused_zchunk(c, S.ntx * S.nty, c->pz);
    launch(0, zchunk, zchunk, c->pz, (c->pz + zchunk - 1) / zchunk, stream);
  };
  // The overlapped multi-rank schedule cuts the slab into the interior - every plane whose chunk cannot reach a ghost
  // plane: a chunk reads planes z0 - 2 .. z1 + 1, so planes [g + 2, pz - g - 2) - and a short chunk of g + 2 planes at
  // either end.  The interior is most of the work and needs nothing from the neighbours.
  const int edge = c->g + 2;
  const int inner = c->pz - 2 * edge;
  const bool splittable = inner >= 8;
  auto launch_interior = [&](hipStream_t stream) {
    const int zchunk = fused_zchunk(c, S.ntx * S.nty, inner);
    launch(edge, zchunk, zchunk, c->pz - edge, (inner + zchunk - 1) / zchunk, stream);
  };
  auto launch_edges = [&](hipStream_t stream) { launch(0, c->pz - edge, edge, c->pz, 2, stream); };
  if (!smi_multi_rank(c)) {
    SmiProfScope prof(c, PROF_FUSED_ADVECT_DIFFUSE, 9.0 * smi_S(c));
    // split_launch (SMI_FUSED_SPLIT_LAUNCH, smi_ctx_set_option): the two launches of the overlapped multi-rank schedule below (interior chunks, then the
    // two end chunks) without the exchange between them - a single GPU can then check that they cover the
    // array exactly like the one launch
    if (c->opt_split_launch && splittable) {
      launch_interior(c->stream);
      launch_edges(c->stream);
    } else {
      launch_all(c->stream);
    }
    SMI_LAUNCH_CHECK();
    *ran = 1;
    return 0;
  }
  const int saved_depth = c->halo_depth;
  c->halo_depth = c->g > 2 ? 2 : 0;  // two planes (0 = all g of them)
  // (the ghost rows of a pencil's y faces reach every chunk: no overlap there)
  const bool overlap = !c->exchange_cb && c->nranks_y == 1 && c->nranks_x == 1 && splittable;
  auto exchange = [&]() -> int {
    if (!overlap) return smi_halo_exchange2_impl(c, velocity, 3, vorticity, 3);
    // No RCCL call may be issued on c->stream while this exchange is pending on the communication stream (one
    // communicator, two streams): the launches below are plain kernels, and the stream waits for ev_b before
    // anything else follows.
    SMI_CHECK_HIP(hipEventRecord(c->ev_a, c->stream));
    SMI_CHECK_HIP(hipStreamWaitEvent(c->comm_stream, c->ev_a, 0));
    SMI_CHECK(smi_halo_exchange2_on(c, velocity, 3, vorticity, 3, c->comm_stream));
    return 0;
  };
  const int rc = exchange();
  c->halo_depth = saved_depth;
  SMI_CHECK(rc);
  {
    SmiProfScope prof(c, PROF_FUSED_ADVECT_DIFFUSE, 9.0 * smi_S(c));
    if (overlap) {
      // The end chunks follow the exchange on ITS stream, next to the interior chunks on the compute stream.  They are
      // short on purpose: whatever a launch of up to one workgroup per CU holds takes the time of its longest
      // workgroup, so end chunks as long as the interior ones (a 68-plane slab of 512^3 in three chunks of 23) made the
      // step wait for the exchange AND for a third of the slab's planes behind it; now the interior chunks carry all
      // planes but 2 (g + 2) and the ends cost 2 (g + 4) plane passes behind the exchange.
      launch_interior(c->stream);
      launch_edges(c->comm_stream);
      SMI_CHECK_HIP(hipEventRecord(c->ev_b, c->comm_stream));
      SMI_CHECK_HIP(hipStreamWaitEvent(c->stream, c->ev_b, 0));
    } else {
      launch_all(c->stream);
    }
    SMI_LAUNCH_CHECK();
  }
  *ran = 1;
  return 0;
}

extern "C" int smi_advection_diffusion_update(smi_ctx* c, void* out, void* velocity, void* vorticity, void* scratch,
                                              double prefactor, double nu_dt_by_dx2) {
  SMI_REQUIRE(c && out && velocity && vorticity && scratch, "smi_advection_diffusion_update: bad argument");
  SMI_REQUIRE(scratch != out && scratch != velocity && scratch != vorticity,
              "smi_advection_diffusion_update: the scratch array must not alias the others");
  int ran = 0;
  SMI_CHECK(smi_advect_diffuse_impl(c, out, velocity, vorticity, prefactor, nu_dt_by_dx2, &ran, nullptr));
  if (ran) return 0;
  SMI_CHECK(smi_advect_vorticity_impl(c, scratch, velocity, vorticity, prefactor));
  return smi_diffusion_apply_impl(c, out, scratch, nu_dt_by_dx2, 3);
}

extern "C" int smi_rotational_advection_update(smi_ctx* c, void* out, void* velocity, void* vorticity,
                                               double prefactor) {
  SMI_REQUIRE(c && out && velocity && vorticity, "smi_rotational_advection_update: bad argument");
  return smi_advect_vorticity_impl(c, out, velocity, vorticity, prefactor);
}

extern "C" int smi_curl(smi_ctx* c, void* curl, void* field, double prefactor) {
  SMI_REQUIRE(c && curl && field, "smi_curl: bad argument");
  return run_curl<0>(c, curl, field, prefactor);
}

extern "C" int smi_divergence(smi_ctx* c, void* divergence, void* field, double inv_dx) {
  SMI_REQUIRE(c && divergence && field, "smi_divergence: bad argument");
  SMI_REQUIRE(divergence != field, "smi_divergence: output and input must not alias");
  if (smi_multi_rank(c)) SMI_CHECK(smi_halo_exchange_impl(c, field, 3));
  Geo G = make_geo(c);
  SMI_DISPATCH(c, {
    T h = (T)0.5 * (T)inv_dx;
    k_divergence<T><<<blocks_for(G.vol), kBlock, 0, c->stream>>>((T*)divergence, (const T*)field, h, G);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int smi_penalise_boundary(smi_ctx* c, void* field, int ncomp, int width, const double* fac_lo,
                                     const double* fac_hi) {
  return smi_penalise_boundary_impl(c, field, ncomp, width, fac_lo, fac_hi, 0);
}

// skip_x: the x bands have been penalised already (by the store of the kernel that produced `field`)
int smi_penalise_boundary_impl(smi_ctx* c, void* field, int ncomp, int width, const double* fac_lo,
                               const double* fac_hi, int skip_x) {
  SMI_REQUIRE(c && field && ncomp >= 1, "smi_penalise_boundary: bad argument");
  SMI_REQUIRE(width >= 0, "invalid zone width");
  if (width == 0) return 0;
  int w = c->g + width;
  SMI_REQUIRE(w <= 16, "smi_penalise_boundary: ghost + width = %d exceeds 16", w);
  SMI_REQUIRE(fac_lo && fac_hi, "smi_penalise_boundary: null factor tables");
  SMI_REQUIRE(2 * w <= c->px && 2 * w <= c->py && 2 * w <= c->pz,
              "smi_penalise_boundary: penalty bands overlap on this grid");
  PenFac fac;
  for (int a = 0; a < 3; ++a)
    for (int k = 0; k < w; ++k) {
      fac.lo[a][k] = fac_lo[a * w + k];
      fac.hi[a][k] = fac_hi[a * w + k];
    }
  Geo G = make_geo(c);
  SmiProfScope prof(c, PROF_PENALISE, 0.0);
  SMI_DISPATCH(c, {
    if (!skip_x && (c->phys_xlo || c->phys_xhi))
      k_penalise<T, 0><<<blocks_for((long long)G.pz * G.py * ncomp), kBlock, 0, c->stream>>>((T*)field, G, ncomp, w,
                                                                                           fac, c->phys_xlo, c->phys_xhi);
    if (c->phys_ylo || c->phys_yhi)
      k_penalise<T, 1><<<blocks_for((long long)G.pz * G.px * ncomp), kBlock, 0, c->stream>>>((T*)field, G, ncomp, w,
                                                                                           fac, c->phys_ylo, c->phys_yhi);
    if (c->phys_zlo || c->phys_zhi)
      k_penalise<T, 2><<<blocks_for(G.plane * ncomp), kBlock, 0, c->stream>>>((T*)field, G, ncomp, w, fac,
                                                                             c->phys_zlo, c->phys_zhi);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

template <typename T>
static int filter_pass(smi_ctx* c, const Geo& G, int axis, T* flux, T* src) {
  if (smi_multi_rank(c)) SMI_CHECK(smi_halo_exchange_impl(c, src, 1));
  int b = blocks_for(G.vol);
  if (axis == 0)
    k_filter_pass<T, 0><<<b, kBlock, 0, c->stream>>>(flux, src, G);
  else if (axis == 1)
    k_filter_pass<T, 1><<<b, kBlock, 0, c->stream>>>(flux, src, G);
  else
    k_filter_pass<T, 2><<<b, kBlock, 0, c->stream>>>(flux, src, G);
  SMI_LAUNCH_CHECK();
  return 0;
}

// out = in - Lz(Ly(Lx in)): the multiplicative filter of order 1 in one pass per component
// penx (may be null): factors of the x bands of the penalisation that follows (4 low, 4 high; ghost + zone width = 4),
// applied on the way out where the kernel stores 16-byte vectors; *pen_done says whether that happened.
int smi_filter_apply_impl(smi_ctx* c, void* out, void* in, int ncomp, const double* penx, int* pen_done) {
  SMI_REQUIRE(out != in, "filter: output and input must not alias");
  if (pen_done) *pen_done = 0;
  if (smi_multi_rank(c)) SMI_CHECK(smi_halo_exchange_impl(c, in, ncomp));
  Geo G = make_geo(c);
  const int zlo_e = G.zlo_c - (c->phys_zlo ? 0 : 1), zhi_e = G.zhi_c + (c->phys_zhi ? 0 : 1);
  SmiProfScope prof(c, PROF_FILTER, 2.0 * ncomp * smi_S(c));
  SMI_DISPATCH(c, {
    PenX<T> X{};
    if (vec_ok<T>(c, out, in)) {
      if (penx && pen_done) {
        X.on_lo = c->phys_xlo;
        X.on_hi = c->phys_xhi;
        for (int e = 0; e < 4; ++e) {
          X.lo[e] = (T)penx[e];
          X.hi[e] = (T)penx[4 + e];
        }
        *pen_done = 1;
      }
      // (chunks of 16 planes like the other tiled kernels: 8, 32 and 64 measured 0.351, 0.360, 0.393 against 0.344 ms
      // at 384^3, although every chunk forms F2 of two planes before it writes its first one)
      const TiledLaunch t = tiled_launch<T>(c, 1, ncomp);
      SMI_CHECK(allow_lds(k_filter_tiled<T>));
      k_filter_tiled<T><<<t.grid, t.threads, t.lds_bytes, c->stream>>>((T*)out, (const T*)in, G, t.zchunk, t.S, zlo_e,
                                                                       zhi_e, X);
    } else {
      LaunchShape s = plan<1>(c, ncomp);
      k_filter_fused<T, 1><<<s.grid, kBlock, 0, c->stream>>>((T*)out, (const T*)in, G, s.zchunk, zlo_e, zhi_e, X);
    }
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int smi_laplacian_filter(smi_ctx* c, void* field, int ncomp, void* filter_flux_buffer,
                                    void* field_buffer, int order, int filter_type) {
  return smi_laplacian_filter_w(c, field, ncomp, filter_flux_buffer, field_buffer, order, filter_type, 1);
}

extern "C" int smi_laplacian_filter_w(smi_ctx* c, void* field, int ncomp, void* filter_flux_buffer,
                                      void* field_buffer, int order, int filter_type, int boundary_width) {
  SMI_REQUIRE(c && field && filter_flux_buffer && field_buffer && ncomp >= 1, "smi_laplacian_filter: bad argument");
  SMI_REQUIRE(order >= 0, "Invalid filter order");
  SMI_REQUIRE(filter_type == SMI_FILTER_MULTIPLICATIVE || filter_type == SMI_FILTER_CONVOLUTION,
              "Invalid filter type");
  SMI_REQUIRE(boundary_width >= 1, "Invalid value for filter flux buffer boundary zone");
  const int w = boundary_width;
  SMI_REQUIRE(c->g + w < c->py - c->g - w && c->g + w < c->px - c->g - w && c->g + w < c->pz - c->g - w,
              "smi_laplacian_filter: boundary width %d leaves no cells", w);
  Geo G = make_geo(c);
  // the filter flux lives on the cells at least g + w away from a physical face (laplacian_filter_mpi_3d.py:110-143);
  // the directional passes take their x / y limits from G.g and their z limits from zlo_c / zhi_c
  G.g = c->g + w - 1;
  G.zlo_c = c->g + (c->phys_zlo ? w : 0);
  G.zhi_c = c->pz - c->g - 1 - (c->phys_zhi ? w : 0);
  G.ylo_c = c->g + (c->phys_ylo ? w : 0);
  G.yhi_c = c->py - c->g - 1 - (c->phys_yhi ? w : 0);
  G.xlo_c = c->g + (c->phys_xlo ? w : 0);
  G.xhi_c = c->px - c->g - 1 - (c->phys_xhi ? w : 0);
  if (filter_type == SMI_FILTER_MULTIPLICATIVE && order == 1 && w == 1) {
    // one fused pass into the scratch field, then back (the scratch fields hold one component)
    const size_t stride = c->vol * smi_elem_size(c);
    for (int k = 0; k < ncomp; ++k) {
      void* fk = (char*)field + (size_t)k * stride;
      SMI_CHECK(smi_filter_apply_impl(c, field_buffer, fk, 1, nullptr, nullptr));
      SMI_CHECK(smi_elementwise_copy(c, fk, field_buffer, 1));
    }
    return 0;
  }
  SmiProfScope prof(c, PROF_FILTER, 2.0 * ncomp * smi_S(c));
  SMI_DISPATCH(c, {
    T* flux = (T*)filter_flux_buffer;
    T* buf = (T*)field_buffer;
    for (int k = 0; k < ncomp; ++k) {
      T* f = (T*)field + (size_t)k * c->vol;
      // The reference copies the flux back into field_buffer after every directional pass;
      // here the two scratch fields alternate roles instead.  Each pass writes every cell
      // (0 outside D), so the result is the same and two field copies per pass are saved.
      if (filter_type == SMI_FILTER_MULTIPLICATIVE) {
        if (order == 0) {
          // degenerate: field -= (stale flux buffer with its physical bands cleared)
          k_zero_bands<T><<<blocks_for(G.vol), kBlock, 0, c->stream>>>(flux, G, c->g + w, c->phys_zlo, c->phys_zhi,
                                                                       c->phys_ylo, c->phys_yhi, c->phys_xlo, c->phys_xhi);
          SMI_LAUNCH_CHECK();
          SMI_CHECK(smi_elementwise_saxpby(c, f, f, 1.0, flux, -1.0, 1));
          continue;
        }
        T* src = f;
        T* dst = flux;
        for (int it = 0; it < order; ++it)
          for (int axis = 0; axis < 3; ++axis) {
            SMI_CHECK(filter_pass<T>(c, G, axis, dst, src));
            src = dst;
            dst = (dst == flux) ? buf : flux;
          }
        SMI_CHECK(smi_elementwise_saxpby(c, f, f, 1.0, src, -1.0, 1));
      } else {
        for (int axis = 0; axis < 3; ++axis) {
          if (order == 0) {
            k_zero_bands<T><<<blocks_for(G.vol), kBlock, 0, c->stream>>>(flux, G, c->g + w, c->phys_zlo, c->phys_zhi,
                                                                       c->phys_ylo, c->phys_yhi, c->phys_xlo,
                                                                       c->phys_xhi);
            SMI_LAUNCH_CHECK();
            SMI_CHECK(smi_elementwise_saxpby(c, f, f, 1.0, flux, -1.0, 1));
            continue;
          }
          T* src = f;
          T* dst = flux;
          for (int it = 0; it < order; ++it) {
            SMI_CHECK(filter_pass<T>(c, G, axis, dst, src));
            src = dst;
            dst = (dst == flux) ? buf : flux;
          }
          SMI_CHECK(smi_elementwise_saxpby(c, f, f, 1.0, src, -1.0, 1));
        }
      }
    }
  });
  return 0;
}

// gen_advection_flux_conservative_eno3_pyst_mpi_kernel_3d (advection_flux_mpi_3d.py:8-195): the flux
// alone, written on the cells the wrapper's seven boxes cover; the rest of `flux` is left as it is.
extern "C" int smi_advection_flux_eno3(smi_ctx* c, void* flux, void* field, void* velocity, double inv_dx) {
  SMI_REQUIRE(c && field && flux && velocity, "smi_advection_flux_eno3: bad argument");
  SMI_REQUIRE(flux != field && flux != velocity, "smi_advection_flux_eno3: flux must not alias its inputs");
  SMI_REQUIRE(c->g >= 2, "smi_advection_flux_eno3: kernel support 2 needs ghost_size >= 2 (got %d)", c->g);
  if (smi_multi_rank(c)) {
    SMI_CHECK(smi_halo_exchange_impl(c, field, 1));
    SMI_CHECK(smi_halo_exchange_impl(c, velocity, 3));
  }
  Geo G = make_geo(c);
  SmiProfScope prof(c, PROF_ADVECTION, 5.0 * smi_S(c));
  SMI_DISPATCH(c, {
    k_advection_flux<T, true><<<blocks_for(G.vol), kBlock, 0, c->stream>>>((T*)flux, (const T*)field,
                                                                          (const T*)velocity, (T)inv_dx, G);
  });
  SMI_LAUNCH_CHECK();
  return 0;
}

extern "C" int smi_advection_timestep_eno3(smi_ctx* c, void* field, void* flux, void* velocity, double dt_by_dx,
                                           int ncomp) {
  SMI_REQUIRE(c && field && flux && velocity && ncomp >= 1, "smi_advection_timestep_eno3: bad argument");
  SMI_REQUIRE(flux != field && flux != velocity, "smi_advection_timestep_eno3: flux must not alias its inputs");
  SMI_REQUIRE(c->g >= 2, "smi_advection_timestep_eno3: kernel support 2 needs ghost_size >= 2 (got %d)", c->g);
  if (smi_multi_rank(c)) SMI_CHECK(smi_halo_exchange_impl(c, velocity, 3));
  Geo G = make_geo(c);
  for (int k = 0; k < ncomp; ++k) {
    void* fk = (char*)field + (size_t)k * c->vol * smi_elem_size(c);
    if (smi_multi_rank(c)) SMI_CHECK(smi_halo_exchange_impl(c, fk, 1));
    {
      SmiProfScope prof(c, PROF_ADVECTION, 5.0 * smi_S(c));
      SMI_DISPATCH(c, {
        k_advection_flux<T><<<blocks_for(G.vol), kBlock, 0, c->stream>>>((T*)flux, (const T*)fk, (const T*)velocity,
                                                                        (T)(-dt_by_dx), G);
      });
      SMI_LAUNCH_CHECK();
    }
    SMI_CHECK(smi_elementwise_sum(c, fk, fk, flux, 1));
  }
  return 0;
}
