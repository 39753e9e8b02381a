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

// ---- row window of a slab-decomposed STEP (one process per GPU, so this is a process-wide setting:
// piso_set_row_window).  Arrays stay globally indexed; the element-wise / gather kernels of the step (assembly, glue, Laplacian,
// CSR product) then work on the face / cell rows of this rank's y-slab only and read their neighbours' rows from halo rows that
// piso_comm_exchange filled.  on = 0: the whole grid (one GPU).
struct RowWin {
  int on, j0, j1, last;      // cell rows [j0, j1); last: this rank also owns the duplicate face row v[ny]
};
RowWin row_window();
// flat u-first face vector (u [ny][nx+1] then v [ny+1][nx]): the windowed elements are two intervals, walked as one index space
struct FaceWin {
  int u_lo, cu, v_lo, cv;
  __host__ __device__ int count() const { return cu + cv; }
  __device__ __forceinline__ int map(int w) const { return w < cu ? u_lo + w : v_lo + (w - cu); }
};
inline FaceWin face_window(int nx, int ny) {
  const RowWin r = row_window();
  const int n_u = (nx + 1) * ny;
  if (!r.on) return FaceWin{0, n_u, n_u, nx * (ny + 1)};
  return FaceWin{r.j0 * (nx + 1), (r.j1 - r.j0) * (nx + 1), n_u + r.j0 * nx, (r.j1 - r.j0 + (r.last ? 1 : 0)) * nx};
}
struct CellWin {
  int lo, n;
};
inline CellWin cell_window(int nx, int ny) {
  const RowWin r = row_window();
  if (!r.on) return CellWin{0, nx * ny};
  return CellWin{r.j0 * nx, (r.j1 - r.j0) * nx};
}

inline int grid_for(long long work_items, int per_block, int cap = kMaxPartials) {
  long long g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  if (g >= kXcds) g = g / kXcds * kXcds;   // keep the XCD split exact
  return static_cast<int>(g);
}

}  // namespace piso
