// Shared device/host helpers for libpiso_hip.so (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/piso_hip.h"

namespace piso {

constexpr int kWave = 64;            // CDNA wavefront
constexpr int kBlock = 256;          // 4 waves, one per SIMD
constexpr int kXcds = 8;             // MI355X: 8 XCDs, block b is observed on XCD b % 8 (speed only, never correctness)
constexpr int kMaxPartials = 2048;   // upper bound on the grid of any kernel that publishes per-block partial sums

void set_error(const char* what, hipError_t err);
void set_error_msg(const char* what);

#define PISO_HIP_CHECK(expr)                                  \
  do {                                                        \
    hipError_t _e = (expr);                                   \
    if (_e != hipSuccess) {                                   \
      ::piso::set_error(#expr, _e);                           \
      return PISO_ERR_HIP;                                    \
    }                                                         \
  } while (0)

#define PISO_LAUNCH_CHECK()                                   \
  do {                                                        \
    hipError_t _e = hipGetLastError();                        \
    if (_e != hipSuccess) {                                   \
      ::piso::set_error("kernel launch", _e);                 \
      return PISO_ERR_HIP;                                    \
    }                                                         \
  } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Bump allocator over the caller-provided workspace.
struct Arena {
  char* base;
  size_t size, used;
  Arena(void* p, size_t n) : base(static_cast<char*>(p)), size(n), used(0) {}
  template <typename T>
  T* take(size_t count) {
    used = align_up(used, 256);
    T* p = reinterpret_cast<T*>(base + used);
    used += count * sizeof(T);
    return p;
  }
  bool ok() const { return used <= size; }
};

// ---- wavefront / block reductions (wave = 64 lanes; __shfl_xor butterflies) --------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    T o = __shfl_xor(v, off, kWave);
    v = o > v ? o : v;
  }
  return v;
}

// NaN-propagating max (a NaN residual must never look "converged")
template <typename T>
__device__ __forceinline__ T nanmax(T a, T b) { return (b > a || b != b) ? b : a; }
template <typename T>
__device__ __forceinline__ T wave_max_nan(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = nanmax(v, __shfl_xor(v, off, kWave));
  return v;
}
// max over a kBlock-thread block, valid in every thread; smem holds 4 values
template <typename T>
__device__ __forceinline__ T block_max_nan(T v, T* smem) {
  v = wave_max_nan(v);
  __syncthreads();
  if ((threadIdx.x & (kWave - 1)) == 0) smem[threadIdx.x >> 6] = v;
  __syncthreads();
  return nanmax(nanmax(smem[0], smem[1]), nanmax(smem[2], smem[3]));
}

// Sum NV values per thread across a kBlock-thread block; result valid in every thread. `smem` holds NV * 4 values.
template <typename T, int NV>
__device__ __forceinline__ void block_sum(T (&v)[NV], T* smem) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = wave_sum(v[k]);
  __syncthreads();   // protect smem reuse
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) smem[k * 4 + wave] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = (smem[k * 4 + 0] + smem[k * 4 + 1]) + (smem[k * 4 + 2] + smem[k * 4 + 3]);
}

// Deterministic reduction of `count` per-block partial records (NV values each, SoA: part[k * kMaxPartials + b]) written
// by the PREVIOUS kernel on the stream; every block of the consuming kernel calls this redundantly (L2-served, tiny).
template <typename T, int NV>
__device__ __forceinline__ void reduce_partials(const T* __restrict__ part, int count, T (&out)[NV], T* smem) {
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    T s = 0;
    for (int b = threadIdx.x; b < count; b += kBlock) s += part[k * kMaxPartials + b];
    out[k] = s;
  }
  block_sum<T, NV>(out, smem);
}

// XCD-aware work split: work items [0, n) are cut into kXcds contiguous chunks; block b (observed on XCD b % 8) walks
// chunk b % 8 with the other blocks of that XCD, so neighbouring items share one L2.  Pure speed; any placement is correct.
struct XcdRange {
  int begin, end, step;
};
__device__ __forceinline__ XcdRange xcd_range(int n) {
  const int nb = gridDim.x, b = blockIdx.x;
  XcdRange r;
  if (nb % kXcds != 0 || nb < kXcds) {
    r.begin = b; r.end = n; r.step = nb;
    return r;
  }
  const int chunk = (n + kXcds - 1) / kXcds;
  const int x = b % kXcds;
  r.begin = x * chunk + b / kXcds;
  r.end = min((x + 1) * chunk, n);
  r.step = nb / kXcds;
  return r;
}

// ---- rows of a slab-decomposed STEP (round 5: LOCAL storage).  Every kernel of the step keeps the index arithmetic of
// the whole grid - boundary rules, periodic seams, the duplicate face row v[ny], the reference's adjoint quirks are decided on GLOBAL
// (i, j) - and only the last step, "where does element (i, j) of this array live", goes through the rank's RowMap: a rank stores its
// own rows plus the rows either side that its gathers read, contiguous in ring order, and nothing else (memory and element-wise
// work per rank are 1 / ranks of the grid's).  on = 0 (one GPU): every array is the whole grid's and every map is the identity.
//   cells, u rows   ring of ny rows:      stored rows cb, cb + 1, ... (mod ny), cr of them          = [j0 - 2, j1 + 2)
//   v rows          ring of ny + 1 rows (the duplicate row v[ny] sits between v[ny - 1] and v[0]): vb, ..., vr of them = [j0 - 3, j1 + 3)
//   padded masks    [ny + 2][nx + 2], rows [mb, mb + mr) = [j0, j1 + 3) (they carry their own border rows: no ring)
//   padded velocity u [ny + 2][nx + 3] rows [j0, j1 + 2), v [ny + 3][nx + 2] rows [j0, j1 + 3)
//   CSR             the rows of the stored face rows, component after component; a row's slots are the whole grid's closed form
//                   minus the first stored row's (plus the component's nnz behind the ring's seam); columns stay GLOBAL (component-
//                   local row numbers of the whole grid) and are mapped where they are used
struct RowMap {
  int on, nx, ny;
  int j0, j1, last;          // owned cell rows [j0, j1); last: this rank also owns the duplicate face row v[ny]
  int cb, cr, vb, vr;        // stored ring rows of cells / u faces and of v faces
  int mb, mr;                // stored rows of the padded masks
  int pur, pvr;              // stored rows of the padded velocities (from row j0)
  int n_u, n_v;              // elements of the two segments of a flat face vector as STORED (one GPU: (nx + 1) ny, nx (ny + 1))
  int csr0[2], csrn[2];      // whole-grid CSR slot of the first stored row of each component; nnz of each whole component
  int nnz_l[2];              // CSR entries stored per component
  __host__ __device__ int urow(int j) const { int d = j - cb; if (d < 0) d += ny; return d; }
  __host__ __device__ int vrow(int j) const { int d = j - vb; if (d < 0) d += ny + 1; return d; }
  // element (row j, column i) of a u-face / v-face / cell array -> its place in the stored array (flat face vectors: u first)
  __host__ __device__ int u(int j, int i) const { return on ? urow(j) * (nx + 1) + i : j * (nx + 1) + i; }
  __host__ __device__ int v(int j, int i) const { return on ? n_u + vrow(j) * nx + i : n_u + j * nx + i; }
  __host__ __device__ int c(int j, int i) const { return on ? urow(j) * nx + i : j * nx + i; }
  // whole-grid flat indices
  __host__ __device__ int face(int f) const {                // u-first face vector
    if (!on) return f;
    const int nug = (nx + 1) * ny;
    if (f < nug) { const int j = f / (nx + 1); return u(j, f - j * (nx + 1)); }
    const int q = f - nug, j = q / nx;
    return v(j, q - j * nx);
  }
  __host__ __device__ int face_vfirst(int k) const {         // v-first face vector (A0)
    if (!on) return k;
    const int nvg = nx * (ny + 1);
    if (k < nvg) { const int j = k / nx; return vrow(j) * nx + (k - j * nx); }
    const int q = k - nvg, j = q / (nx + 1);
    return n_v + urow(j) * (nx + 1) + (q - j * (nx + 1));
  }
  __host__ __device__ int cell(int cidx) const { if (!on) return cidx; const int j = cidx / nx; return c(j, cidx - j * nx); }
  __host__ __device__ int mask(int m) const { return on ? m - mb * (nx + 2) : m; }
  // padded velocities (flat: padded u, then padded v)
  __host__ __device__ int pad(int k) const {
    if (!on) return k;
    const int pu = (ny + 2) * (nx + 3);
    if (k < pu) return k - j0 * (nx + 3);
    return pur * (nx + 3) + (k - pu) - j0 * (nx + 2);
  }
  // CSR: slot `s` (whole-grid numbering, component-local) of an entry of face row j of component comp -> stored slot (component-local)
  __host__ __device__ int slot(int comp, int j, int s) const {
    if (!on) return s;
    const int base = comp ? vb : cb;
    return s - csr0[comp] + (j < base ? csrn[comp] : 0);
  }
  // row pointer arrays: [stored rows of u + 1][stored rows of v + 1]; component-local row number of the whole grid -> stored row number
  __host__ __device__ int frow(int comp, int row) const {
    if (!on) return row;
    const int W = nx + (comp == 0);
    const int j = row / W;
    return (comp ? vrow(j) : urow(j)) * W + (row - j * W);
  }
};
// the owned elements of a launch, walked as one index space of WHOLE-GRID indices (flat u-first face vector: two intervals)
struct FaceWin {
  int u_lo, cu, v_lo, cv;
  __host__ __device__ int count() const { return cu + cv; }
  __device__ __forceinline__ int map(int w) const { return w < cu ? u_lo + w : v_lo + (w - cu); }
};
struct CellWin {
  int lo, n;
};
// closed-form CSR end offset of `row` (calcCsrRowPtrGpu, central_difference_csr_op.cu.cc:472-505, 2-D branch); W, H: the component's dims
__host__ __device__ inline int csr_row_end(int row, int i, int j, int W, int H, int per_x, int per_y) {
  int r = (row + 1) * 5;
  const int j1 = j < 1 ? j : 1;
  r -= j1 * (W * (1 - per_y));
  const int hi_j = (j + 1 - H) > -1 ? (j + 1 - H) : -1, hi_i = (i + 1 - W) > -1 ? (i + 1 - W) : -1;
  r -= ((1 - j1) + (1 + hi_j)) * (i + 1) * (1 - per_y);
  r -= (j * 2 + 1 + (1 + hi_i)) * (1 - per_x);
  return r;
}
// slab = NULL: the whole grid.  per_x / per_y only matter for the CSR numbers.
inline RowMap make_row_map(const piso_slab_t* slab, int nx, int ny, int per_x = 0, int per_y = 0) {
  RowMap m;
  memset(&m, 0, sizeof(m));
  m.nx = nx; m.ny = ny; m.j0 = 0; m.j1 = ny; m.last = 1;
  m.n_u = (nx + 1) * ny; m.n_v = nx * (ny + 1);
  m.cr = ny; m.vr = ny + 1; m.mr = ny + 2; m.pur = ny + 2; m.pvr = ny + 3;
  int nnz[2];
  nnz[0] = 5 * (nx + 1) * ny - 2 * ny * (1 - per_x) - 2 * (nx + 1) * (1 - per_y);      // diffpiso/piso_tf.py:102-106
  nnz[1] = 5 * nx * (ny + 1) - 2 * (ny + 1) * (1 - per_x) - 2 * nx * (1 - per_y);
  m.csrn[0] = nnz[0]; m.csrn[1] = nnz[1]; m.nnz_l[0] = nnz[0]; m.nnz_l[1] = nnz[1];
  if (!slab) return m;
  m.on = 1;
  m.j0 = slab->row_begin; m.j1 = slab->row_end; m.last = slab->owns_last_face_row ? 1 : 0;
  const int nyl = m.j1 - m.j0;
  m.cb = ((m.j0 - 2) % ny + ny) % ny; m.cr = nyl + 4;
  m.vb = ((m.j0 - 3) % (ny + 1) + (ny + 1)) % (ny + 1); m.vr = nyl + 6;
  m.mb = m.j0; m.mr = nyl + 3;
  m.pur = nyl + 2; m.pvr = nyl + 3;
  m.n_u = m.cr * (nx + 1); m.n_v = m.vr * nx;
  for (int comp = 0; comp < 2; ++comp) {
    const int W = nx + (comp == 0), H = ny + (comp == 1), base = comp ? m.vb : m.cb, rows = comp ? m.vr : m.cr;
    const int first = base * W;                             // first stored row (component-local row number)
    m.csr0[comp] = first > 0 ? csr_row_end(first - 1, W - 1, base - 1, W, H, per_x, per_y) : 0;
    // entries stored: from the first stored row to the last, around the ring
    const int jl = (base + rows - 1) % H;                   // last stored face row
    const int endl = csr_row_end(jl * W + W - 1, W - 1, jl, W, H, per_x, per_y);
    m.nnz_l[comp] = (base + rows <= H) ? endl - m.csr0[comp] : (nnz[comp] - m.csr0[comp]) + endl;
  }
  return m;
}
inline bool slab_ok(const piso_slab_t* s, int ny) {
  return !s || (s->ny_global == ny && s->row_begin >= 0 && s->row_end > s->row_begin && s->row_end <= ny && s->row_end - s->row_begin >= 4 &&
                (s->row_end - s->row_begin) + 6 <= ny);
}
inline FaceWin face_window(const RowMap& m) {
  const int nx = m.nx, ny = m.ny, n_u = (nx + 1) * ny;
  return FaceWin{m.j0 * (nx + 1), (m.j1 - m.j0) * (nx + 1), n_u + m.j0 * nx, (m.j1 - m.j0 + (m.last ? 1 : 0)) * nx};
}
inline CellWin cell_window(const RowMap& m) { return CellWin{m.j0 * m.nx, (m.j1 - m.j0) * m.nx}; }

inline int grid_for(long long work_items, int per_block, int cap = kMaxPartials) {
  long long g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  if (g >= kXcds) g = g / kXcds * kXcds;   // keep the XCD split exact
  return static_cast<int>(g);
}

}  // namespace piso
