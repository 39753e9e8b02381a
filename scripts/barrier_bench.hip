// Micro-benchmark of grid-wide "barrier + 3 sums" exchanges on MI355X (diagnostics for cg_persist.h; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/barrier_bench scripts/barrier_bench.hip && /tmp/barrier_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

constexpr int kThreads = 512, kWaves = 8, kMaxG = 256;
typedef unsigned long long u64;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double sum16(double v) {
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

struct Ctl {
  unsigned* bar; double* parts;       // V0
  u64* rec;                           // V1/V2: [G][8] tagged words (6 used), 64-byte records
  u64* grp;                           // V2: [G/16][8]
  u64* loc;                           // V3: XCC-local
  int* xcc;                           // out: XCC id per block
  int* xcnt;                          // V30: arrivals per XCD
};

// ---------------- V0: counter + fence + partials (the first version of cg_persist.h; kept as the baseline of the comparison)
__device__ void exch_v0(const Ctl& c, double (&v)[3], unsigned epoch, double* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int q = 0; q < 3; ++q) v[q] = wave_sum(v[q]);
  __syncthreads();
  if (lane == 0) for (int q = 0; q < 3; ++q) smem[q * kWaves + wave] = v[q];
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int q = 0; q < 3; ++q) { double s = 0; for (int w = 0; w < kWaves; ++w) s += smem[q * kWaves + w]; c.parts[q * kMaxG + blockIdx.x] = s; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(c.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(c.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * gridDim.x) __builtin_amdgcn_s_sleep(2);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (wave == 0) {
    double s[3] = {0, 0, 0};
    for (int b = lane; b < (int)gridDim.x; b += 64) for (int q = 0; q < 3; ++q) s[q] += c.parts[q * kMaxG + b];
    for (int q = 0; q < 3; ++q) s[q] = wave_sum(s[q]);
    if (lane == 0) for (int q = 0; q < 3; ++q) smem[3 * kWaves + q] = s[q];
  }
  __syncthreads();
  for (int q = 0; q < 3; ++q) v[q] = smem[3 * kWaves + q];
  __syncthreads();
}

// tagged words: a double travels as two 8-byte words {32 payload bits, 32-bit epoch}; 8-byte accesses are single-copy atomic
__device__ __forceinline__ void put_tagged(u64* rec, const double (&s)[3], unsigned epoch) {
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const u64 bits = __double_as_longlong(s[q]);
    __hip_atomic_store(rec + 2 * q, ((bits & 0xffffffffull) << 32) | epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(rec + 2 * q + 1, (bits & 0xffffffff00000000ull) | epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// poll one record until all six words carry `epoch`; `active` lanes only; returns the three doubles
__device__ __forceinline__ void get_tagged(const u64* rec, bool active, unsigned epoch, double (&s)[3]) {
  u64 w[6];
  bool ok = !active;
  for (int q = 0; q < 6; ++q) w[q] = 0;
  while (true) {
    if (!ok) {
#pragma unroll
      for (int q = 0; q < 6; ++q) w[q] = __hip_atomic_load(rec + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ok = true;
#pragma unroll
      for (int q = 0; q < 6; ++q) ok = ok && ((unsigned)(w[q] & 0xffffffffull) == epoch);
    }
    if (__all(ok)) break;
    __builtin_amdgcn_s_sleep(1);
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) s[q] = active ? __longlong_as_double((w[2 * q] >> 32) | (w[2 * q + 1] & 0xffffffff00000000ull)) : 0.0;
}

// workgroup partial with two __syncthreads (parity double-buffered LDS); result valid in wave 0
__device__ __forceinline__ void wg_partial(double (&v)[3], double* smem, unsigned epoch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* sm = smem + (epoch & 1) * 32;
  for (int q = 0; q < 3; ++q) v[q] = wave_sum(v[q]);
  if (lane == 0) for (int q = 0; q < 3; ++q) sm[q * kWaves + wave] = v[q];
  __syncthreads();
  if (wave == 0) for (int q = 0; q < 3; ++q) { double s = 0; for (int w = 0; w < kWaves; ++w) s += sm[q * kWaves + w]; v[q] = s; }
}
__device__ __forceinline__ void wg_broadcast(double (&v)[3], double* smem, unsigned epoch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* sm = smem + (epoch & 1) * 32 + 24;
  if (wave == 0 && lane == 0) for (int q = 0; q < 3; ++q) sm[q] = v[q];
  __syncthreads();
  for (int q = 0; q < 3; ++q) v[q] = sm[q];
}

// ---------------- V1: flat all-to-all of tagged records
__device__ void exch_v1(const Ctl& c, double (&v)[3], unsigned epoch, double* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  wg_partial(v, smem, epoch);
  if (wave == 0) {
    if (lane == 0) put_tagged(c.rec + (size_t)blockIdx.x * 8, v, epoch);
    double tot[3] = {0, 0, 0};
    for (int m = 0; m < ((int)gridDim.x + 63) / 64; ++m) {
      const int b = m * 64 + lane;
      double s[3];
      get_tagged(c.rec + (size_t)b * 8, b < (int)gridDim.x, epoch, s);
      for (int q = 0; q < 3; ++q) tot[q] += s[q];
    }
    for (int q = 0; q < 3; ++q) v[q] = wave_sum(tot[q]);
  }
  wg_broadcast(v, smem, epoch);
}

// ---------------- V3: flat all-to-all, every lane polls its (up to) NR records concurrently
template <int NR>
__device__ void exch_v3(const Ctl& c, double (&v)[3], unsigned epoch, double* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  wg_partial(v, smem, epoch);
  if (wave == 0) {
    if (lane == 0) put_tagged(c.rec + (size_t)blockIdx.x * 8, v, epoch);
    u64 w[NR][6];
    bool ok[NR];
#pragma unroll
    for (int m = 0; m < NR; ++m) ok[m] = (m * 64 + lane) >= (int)gridDim.x;
    while (true) {
      bool all = true;
#pragma unroll
      for (int m = 0; m < NR; ++m) {
        if (!ok[m]) {
          const u64* rec = c.rec + (size_t)(m * 64 + lane) * 8;
#pragma unroll
          for (int q = 0; q < 6; ++q) w[m][q] = __hip_atomic_load(rec + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
#pragma unroll
      for (int m = 0; m < NR; ++m) {
        if (!ok[m]) {
          bool o = true;
#pragma unroll
          for (int q = 0; q < 6; ++q) o = o && ((unsigned)(w[m][q] & 0xffffffffull) == epoch);
          ok[m] = o;
        }
        all = all && ok[m];
      }
      if (__all(all)) break;
    }
    double tot[3] = {0, 0, 0};
#pragma unroll
    for (int m = 0; m < NR; ++m) {
      const bool act = (m * 64 + lane) < (int)gridDim.x;
#pragma unroll
      for (int q = 0; q < 3; ++q) tot[q] += act ? __longlong_as_double((w[m][2 * q] >> 32) | (w[m][2 * q + 1] & 0xffffffff00000000ull)) : 0.0;
    }
    for (int q = 0; q < 3; ++q) v[q] = wave_sum(tot[q]);
  }
  wg_broadcast(v, smem, epoch);
}

// ---------------- V4: as V3 but word-major layout: word q of workgroup b at rec[q * kMaxG + b] (coalesced polls)
template <int NR, int SLEEP>
__device__ void exch_v4(const Ctl& c, double (&v)[3], unsigned epoch, double* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  wg_partial(v, smem, epoch);
  if (wave == 0) {
    if (lane < 6) {
      const u64 bits = __double_as_longlong(v[lane >> 1 == 0 ? 0 : (lane >> 1 == 1 ? 1 : 2)]);
      const u64 word = (lane & 1) ? ((bits & 0xffffffff00000000ull) | epoch) : (((bits & 0xffffffffull) << 32) | epoch);
      __hip_atomic_store(c.rec + (size_t)lane * kMaxG + blockIdx.x, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    u64 w[NR][6];
    bool ok[NR];
#pragma unroll
    for (int m = 0; m < NR; ++m) ok[m] = (m * 64 + lane) >= (int)gridDim.x;
    while (true) {
      bool all = true;
#pragma unroll
      for (int m = 0; m < NR; ++m) {
        if (!ok[m]) {
#pragma unroll
          for (int q = 0; q < 6; ++q) w[m][q] = __hip_atomic_load(c.rec + (size_t)q * kMaxG + m * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
#pragma unroll
      for (int m = 0; m < NR; ++m) {
        if (!ok[m]) {
          bool o = true;
#pragma unroll
          for (int q = 0; q < 6; ++q) o = o && ((unsigned)(w[m][q] & 0xffffffffull) == epoch);
          ok[m] = o;
        }
        all = all && ok[m];
      }
      if (__all(all)) break;
      if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
    }
    double tot[3] = {0, 0, 0};
#pragma unroll
    for (int m = 0; m < NR; ++m) {
      const bool act = (m * 64 + lane) < (int)gridDim.x;
#pragma unroll
      for (int q = 0; q < 3; ++q) tot[q] += act ? __longlong_as_double((w[m][2 * q] >> 32) | (w[m][2 * q + 1] & 0xffffffff00000000ull)) : 0.0;
    }
    for (int q = 0; q < 3; ++q) v[q] = wave_sum(tot[q]);
  }
  wg_broadcast(v, smem, epoch);
}

// ---------------- V7: AoS records of 8 words, polled cooperatively: lane l reads word l % 8 of record 8 i + l / 8 (one
// coalesced 512-byte load per 8 records, all G / 8 loads in flight at once); 6 lanes publish the record with one store
template <int NI>   // NI = max records / 8
__device__ void exch_v7(const Ctl& c, double (&v)[3], unsigned epoch, double* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* sm = smem + (epoch & 1) * 32;
  for (int q = 0; q < 3; ++q) v[q] = wave_sum(v[q]);
  if (lane == 0) for (int q = 0; q < 3; ++q) sm[q * kWaves + wave] = v[q];
  __syncthreads();
  if (wave == 0) {
    const int wq = lane & 7, grp = lane >> 3;
    if (lane < 6) {
      double s = 0;
      for (int w = 0; w < kWaves; ++w) s += sm[(lane >> 1) * kWaves + w];
      const u64 bits = __double_as_longlong(s);
      const u64 word = (lane & 1) ? ((bits & 0xffffffff00000000ull) | epoch) : (((bits & 0xffffffffull) << 32) | epoch);
      __hip_atomic_store(c.rec + (size_t)blockIdx.x * 8 + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int ni = ((int)gridDim.x + 7) / 8;
    u64 w[NI];
    bool ok[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) { w[i] = 0; ok[i] = !(i < ni && wq < 6 && (i * 8 + grp) < (int)gridDim.x); }
    while (true) {
      bool all = true;
#pragma unroll
      for (int i = 0; i < NI; ++i)
        if (!ok[i]) w[i] = __hip_atomic_load(c.rec + (size_t)i * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        if (!ok[i]) ok[i] = ((unsigned)(w[i] & 0xffffffffull) == epoch);
        all = all && ok[i];
      }
      if (__all(all)) break;
    }
    double acc = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const u64 nxt = __shfl_down(w[i], 1, 64);        // the hi word sits in the next lane
      const bool act = i < ni && wq < 6 && !(wq & 1) && (i * 8 + grp) < (int)gridDim.x;
      acc += act ? __longlong_as_double((w[i] >> 32) | (nxt & 0xffffffff00000000ull)) : 0.0;
    }
    acc += __shfl_xor(acc, 8, 64); acc += __shfl_xor(acc, 16, 64); acc += __shfl_xor(acc, 32, 64);
    for (int q = 0; q < 3; ++q) v[q] = __shfl(acc, 2 * q, 64);
  }
  wg_broadcast(v, smem, epoch);
}

// ---------------- V8: flat, every lane re-reads ALL its NR records until all carry the epoch (branch-free inner body)
template <int NR>
__device__ void exch_v8(const Ctl& c, double (&v)[3], unsigned epoch, double* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  wg_partial(v, smem, epoch);
  if (wave == 0) {
    if (lane == 0) put_tagged(c.rec + (size_t)blockIdx.x * 8, v, epoch);
    u64 w[NR][6];
    const u64* rp[NR];
#pragma unroll
    for (int m = 0; m < NR; ++m) { const int b = m * 64 + lane; rp[m] = c.rec + (size_t)(b < (int)gridDim.x ? b : blockIdx.x) * 8; }
    while (true) {
#pragma unroll
      for (int m = 0; m < NR; ++m)
#pragma unroll
        for (int q = 0; q < 6; ++q) w[m][q] = __hip_atomic_load(rp[m] + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned bad = 0;
#pragma unroll
      for (int m = 0; m < NR; ++m)
#pragma unroll
        for (int q = 0; q < 6; ++q) bad |= ((unsigned)(w[m][q] & 0xffffffffull)) ^ epoch;
      if (__all(bad == 0)) break;
    }
    double tot[3] = {0, 0, 0};
#pragma unroll
    for (int m = 0; m < NR; ++m) {
      const bool act = (m * 64 + lane) < (int)gridDim.x;
#pragma unroll
      for (int q = 0; q < 3; ++q) tot[q] += act ? __longlong_as_double((w[m][2 * q] >> 32) | (w[m][2 * q + 1] & 0xffffffff00000000ull)) : 0.0;
    }
    for (int q = 0; q < 3; ++q) v[q] = wave_sum(tot[q]);
  }
  wg_broadcast(v, smem, epoch);
}

// ---------------- V10: as V1, but a lane polls ONE word of its record (the last one stored) and reads the other five only
// once that carries the epoch; SLEEP = back-off between polls
template <int SLEEP, bool ONEWORD>
__device__ void exch_v10(const Ctl& c, double (&v)[3], unsigned epoch, double* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  wg_partial(v, smem, epoch);
  if (wave == 0) {
    if (lane == 0) put_tagged(c.rec + (size_t)blockIdx.x * 8, v, epoch);
    double tot[3] = {0, 0, 0};
    for (int m = 0; m < ((int)gridDim.x + 63) / 64; ++m) {
      const int b = m * 64 + lane;
      const bool active = b < (int)gridDim.x;
      const u64* rec = c.rec + (size_t)(active ? b : 0) * 8;
      u64 w[6] = {0, 0, 0, 0, 0, 0};
      bool ok = !active;
      while (true) {
        if (!ok) {
          if (ONEWORD) {
            w[5] = __hip_atomic_load(rec + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(w[5] & 0xffffffffull) == epoch) {
#pragma unroll
              for (int q = 0; q < 5; ++q) w[q] = __hip_atomic_load(rec + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              ok = true;
#pragma unroll
              for (int q = 0; q < 5; ++q) ok = ok && ((unsigned)(w[q] & 0xffffffffull) == epoch);
            }
          } else {
#pragma unroll
            for (int q = 0; q < 6; ++q) w[q] = __hip_atomic_load(rec + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = true;
#pragma unroll
            for (int q = 0; q < 6; ++q) ok = ok && ((unsigned)(w[q] & 0xffffffffull) == epoch);
          }
        }
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(SLEEP);
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) tot[q] += active ? __longlong_as_double((w[2 * q] >> 32) | (w[2 * q + 1] & 0xffffffff00000000ull)) : 0.0;
    }
    for (int q = 0; q < 3; ++q) v[q] = wave_sum(tot[q]);
  }
  wg_broadcast(v, smem, epoch);
}

// ---------------- V15: as V1, but wave m polls records [64 m, 64 m + 64) - the rounds run concurrently on different waves
__device__ void exch_v15(const Ctl& c, double (&v)[3], unsigned epoch, double* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* sm = smem + (epoch & 1) * 48;
  for (int q = 0; q < 3; ++q) v[q] = wave_sum(v[q]);
  if (lane == 0) for (int q = 0; q < 3; ++q) sm[q * kWaves + wave] = v[q];
  __syncthreads();
  const int nm = ((int)gridDim.x + 63) / 64;
  if (wave < nm) {
    if (wave == 0 && lane == 0) {
      double s[3];
      for (int q = 0; q < 3; ++q) { s[q] = 0; for (int w = 0; w < kWaves; ++w) s[q] += sm[q * kWaves + w]; }
      put_tagged(c.rec + (size_t)blockIdx.x * 8, s, epoch);
    }
    const int b = wave * 64 + lane;
    double s[3];
    get_tagged(c.rec + (size_t)(b < (int)gridDim.x ? b : 0) * 8, b < (int)gridDim.x, epoch, s);
    for (int q = 0; q < 3; ++q) s[q] = wave_sum(s[q]);
    if (lane == 0) for (int q = 0; q < 3; ++q) sm[24 + q * 4 + wave] = s[q];
  }
  __syncthreads();
  for (int q = 0; q < 3; ++q) { double t = 0; for (int m = 0; m < nm; ++m) t += sm[24 + q * 4 + m]; v[q] = t; }
}

// ---------------- V16: V1 (sequential rounds, sleep 1) on the word-major layout rec[q * kMaxG + b]; BLK > 0: blocks of BLK
// records, word-major inside a block
template <int BLK>
__device__ void exch_v16(const Ctl& c, double (&v)[3], unsigned epoch, double* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  wg_partial(v, smem, epoch);
  auto addr = [&](int b, int q) -> u64* {
    if (BLK == 0) return c.rec + (size_t)q * kMaxG + b;
    return c.rec + (size_t)(b / BLK) * (BLK * 8) + q * BLK + (b % BLK);
  };
  if (wave == 0) {
    if (lane < 6) {
      const u64 bits = __double_as_longlong(v[lane >> 1 == 0 ? 0 : (lane >> 1 == 1 ? 1 : 2)]);
      const u64 word = (lane & 1) ? ((bits & 0xffffffff00000000ull) | epoch) : (((bits & 0xffffffffull) << 32) | epoch);
      __hip_atomic_store(addr(blockIdx.x, lane), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    double tot[3] = {0, 0, 0};
    for (int m = 0; m < ((int)gridDim.x + 63) / 64; ++m) {
      const int b = m * 64 + lane;
      const bool active = b < (int)gridDim.x;
      u64 w[6] = {0, 0, 0, 0, 0, 0};
      bool ok = !active;
      while (true) {
        if (!ok) {
#pragma unroll
          for (int q = 0; q < 6; ++q) w[q] = __hip_atomic_load(addr(active ? b : 0, q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = true;
#pragma unroll
          for (int q = 0; q < 6; ++q) ok = ok && ((unsigned)(w[q] & 0xffffffffull) == epoch);
        }
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) tot[q] += active ? __longlong_as_double((w[2 * q] >> 32) | (w[2 * q + 1] & 0xffffffff00000000ull)) : 0.0;
    }
    for (int q = 0; q < 3; ++q) v[q] = wave_sum(tot[q]);
  }
  wg_broadcast(v, smem, epoch);
}

// ---------------- V19: poll the lane's FIRST record like V1; once that round passes, read the remaining records concurrently
// (arrivals cluster: they are almost always there) and fall back to polling only for the late ones.  BLK layout as V16.
template <int BLK, int NR>
__device__ void exch_v19(const Ctl& c, double (&v)[3], unsigned epoch, double* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  wg_partial(v, smem, epoch);
  auto addr = [&](int b, int q) -> u64* {
    if (BLK == 0) return c.rec + (size_t)b * 8 + q;
    return c.rec + (size_t)(b / BLK) * (BLK * 8) + q * BLK + (b % BLK);
  };
  if (wave == 0) {
    if (lane < 6) {
      const u64 bits = __double_as_longlong(v[lane >> 1 == 0 ? 0 : (lane >> 1 == 1 ? 1 : 2)]);
      const u64 word = (lane & 1) ? ((bits & 0xffffffff00000000ull) | epoch) : (((bits & 0xffffffffull) << 32) | epoch);
      __hip_atomic_store(addr(blockIdx.x, lane), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    u64 w[NR][6];
    bool ok[NR];
#pragma unroll
    for (int m = 0; m < NR; ++m) ok[m] = (m * 64 + lane) >= (int)gridDim.x;
    // round 0: first record only
    while (true) {
      if (!ok[0]) {
#pragma unroll
        for (int q = 0; q < 6; ++q) w[0][q] = __hip_atomic_load(addr(lane, q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool o = true;
#pragma unroll
        for (int q = 0; q < 6; ++q) o = o && ((unsigned)(w[0][q] & 0xffffffffull) == epoch);
        ok[0] = o;
      }
      if (__all(ok[0])) break;
      __builtin_amdgcn_s_sleep(1);
    }
    // the rest: all at once, then poll the stragglers
    while (true) {
      bool all = true;
#pragma unroll
      for (int m = 1; m < NR; ++m)
        if (!ok[m]) {
#pragma unroll
          for (int q = 0; q < 6; ++q) w[m][q] = __hip_atomic_load(addr(m * 64 + lane, q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
      for (int m = 1; m < NR; ++m) {
        if (!ok[m]) {
          bool o = true;
#pragma unroll
          for (int q = 0; q < 6; ++q) o = o && ((unsigned)(w[m][q] & 0xffffffffull) == epoch);
          ok[m] = o;
        }
        all = all && ok[m];
      }
      if (__all(all)) break;
      __builtin_amdgcn_s_sleep(1);
    }
    double tot[3] = {0, 0, 0};
#pragma unroll
    for (int m = 0; m < NR; ++m) {
      const bool act = (m * 64 + lane) < (int)gridDim.x;
#pragma unroll
      for (int q = 0; q < 3; ++q) tot[q] += act ? __longlong_as_double((w[m][2 * q] >> 32) | (w[m][2 * q + 1] & 0xffffffff00000000ull)) : 0.0;
    }
    for (int q = 0; q < 3; ++q) v[q] = wave_sum(tot[q]);
  }
  wg_broadcast(v, smem, epoch);
}

// ---------------- V21: NW waves poll disjoint slices of the records (sequential rounds inside a wave, sleep 1), one-store publish
template <int NW>
__device__ void exch_v21(const Ctl& c, double (&v)[3], unsigned epoch, double* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* sm = smem + (epoch & 1) * 48;
  for (int q = 0; q < 3; ++q) v[q] = wave_sum(v[q]);
  if (lane == 0) for (int q = 0; q < 3; ++q) sm[q * kWaves + wave] = v[q];
  __syncthreads();
  const int rounds = ((int)gridDim.x + 63) / 64;
  if (wave < NW) {
    if (wave == 0) {
      double s[3];
      for (int q = 0; q < 3; ++q) { s[q] = 0; for (int w = 0; w < kWaves; ++w) s[q] += sm[q * kWaves + w]; }
      const int vq = lane >> 1;
      const u64 bits = __double_as_longlong(vq == 0 ? s[0] : (vq == 1 ? s[1] : s[2]));
      const u64 word = (lane & 1) ? ((bits & 0xffffffff00000000ull) | epoch) : (((bits & 0xffffffffull) << 32) | epoch);
      if (lane < 6) __hip_atomic_store(c.rec + (size_t)blockIdx.x * 8 + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    double tot[3] = {0, 0, 0};
    for (int m = wave; m < rounds; m += NW) {
      const int b = m * 64 + lane;
      double s[3];
      get_tagged(c.rec + (size_t)(b < (int)gridDim.x ? b : 0) * 8, b < (int)gridDim.x, epoch, s);
      for (int q = 0; q < 3; ++q) tot[q] += s[q];
    }
    for (int q = 0; q < 3; ++q) tot[q] = wave_sum(tot[q]);
    if (lane == 0) for (int q = 0; q < 3; ++q) sm[24 + q * 4 + wave] = tot[q];
  }
  __syncthreads();
  for (int q = 0; q < 3; ++q) { double t = 0; for (int m = 0; m < NW; ++m) t += sm[24 + q * 4 + m]; v[q] = t; }
}

// ---------------- V2: two levels of 16 (leader = first workgroup of each group of 16)
__device__ void exch_v2(const Ctl& c, double (&v)[3], unsigned epoch, double* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int G = gridDim.x, ngrp = (G + 15) / 16, g = blockIdx.x / 16;
  wg_partial(v, smem, epoch);
  if (wave == 0) {
    if (lane == 0) put_tagged(c.rec + (size_t)blockIdx.x * 8, v, epoch);
    if ((blockIdx.x & 15) == 0) {
      const int b = g * 16 + lane;
      double s[3];
      get_tagged(c.rec + (size_t)b * 8, lane < 16 && b < G, epoch, s);
      for (int q = 0; q < 3; ++q) s[q] = sum16(s[q]);
      if (lane == 0) put_tagged(c.grp + (size_t)g * 8, s, epoch);
    }
    double s[3];
    get_tagged(c.grp + (size_t)lane * 8, lane < ngrp, epoch, s);
    for (int q = 0; q < 3; ++q) v[q] = sum16(s[q]);
  }
  wg_broadcast(v, smem, epoch);
}


// ---------------- V30: two levels that follow the hardware: level 1 inside an XCD through that XCD's L2 (records stored without
// sc1 - they stay in the L2 - and polled with sc1 loads, which bypass the L1 and are served by the L2), level 2 = the eight XCD
// records through the fabric (sc1 stores, sc1 loads), published by the first arrival of every XCD.  xcd / rank: the XCC id of the
// workgroup and its arrival rank there (one atomic per launch).  AoS records of 8 words, coalesced cooperative polls as V7.
template <bool LEADER_ONLY>   // V31: only the first arrival of an XCD reads the level-1 records (a tree: 256 -> 8 leaders -> everybody)
__device__ void exch_v30(const Ctl& c, double (&v)[3], unsigned epoch, double* smem, int xcd, int rank, int per_xcd) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* sm = smem + (epoch & 1) * 32;
  for (int q = 0; q < 3; ++q) v[q] = wave_sum(v[q]);
  if (lane == 0) for (int q = 0; q < 3; ++q) sm[q * kWaves + wave] = v[q];
  __syncthreads();
  if (wave == 0) {
    const int wq = lane & 7, grp = lane >> 3;
    u64* loc = c.loc + (size_t)(epoch & 1) * kMaxG * 8 + (size_t)xcd * 32 * 8;
    if (lane < 6) {
      double s = 0;
      for (int w = 0; w < kWaves; ++w) s += sm[(lane >> 1) * kWaves + w];
      const u64 bits = __double_as_longlong(s);
      const u64 word = (lane & 1) ? ((bits & 0xffffffff00000000ull) | epoch) : (((bits & 0xffffffffull) << 32) | epoch);
      __hip_atomic_store(loc + (size_t)rank * 8 + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    u64 w[4];
    bool ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { w[i] = 0; ok[i] = !(wq < 6 && (i * 8 + grp) < per_xcd) || (LEADER_ONLY && rank != 0); }
    unsigned spins = 0;
    while (true) {
      bool all = true;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (!ok[i]) w[i] = __hip_atomic_load(loc + (size_t)i * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!ok[i]) ok[i] = ((unsigned)(w[i] & 0xffffffffull) == epoch);
        all = all && ok[i];
      }
      if (__all(all) || ++spins > (1u << 20)) break;
    }
    double acc = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u64 nxt = __shfl_down(w[i], 1, 64);
      const bool act = wq < 6 && !(wq & 1) && (i * 8 + grp) < per_xcd;
      acc += act ? __longlong_as_double((w[i] >> 32) | (nxt & 0xffffffff00000000ull)) : 0.0;
    }
    acc += __shfl_xor(acc, 8, 64); acc += __shfl_xor(acc, 16, 64); acc += __shfl_xor(acc, 32, 64);   // lane 2 q: XCD total of value q
    u64* grpr = c.grp + (size_t)(epoch & 1) * 64;
    if (rank == 0 && lane < 6) {
      const double s = __shfl(acc, lane & 6, 64);
      const u64 bits = __double_as_longlong(s);
      const u64 word = (lane & 1) ? ((bits & 0xffffffff00000000ull) | epoch) : (((bits & 0xffffffffull) << 32) | epoch);
      __hip_atomic_store(grpr + (size_t)xcd * 8 + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    u64 g = 0;
    const bool need = wq < 6;
    while (true) {
      g = __hip_atomic_load(grpr + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__all(!need || (unsigned)(g & 0xffffffffull) == epoch) || ++spins > (1u << 21)) break;
    }
    const u64 nxt = __shfl_down(g, 1, 64);
    double t = (wq < 6 && !(wq & 1)) ? __longlong_as_double((g >> 32) | (nxt & 0xffffffff00000000ull)) : 0.0;
    t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
    for (int q = 0; q < 3; ++q) v[q] = __shfl(t, 2 * q, 64);
  }
  wg_broadcast(v, smem, epoch);
}

template <int VAR>
__global__ __launch_bounds__(kThreads) void bench(Ctl c, int iters, double* out, u64* ticks) {
  __shared__ double smem[96];
  double acc = 0;
  if (threadIdx.x == 0) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    c.xcc[blockIdx.x] = (int)(x & 0xf);
  }
  __shared__ int xr[2];
  if (VAR == 30 || VAR == 31) {
    if (threadIdx.x == 0) {
      unsigned x;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
      xr[0] = (int)(x & 7);
      xr[1] = __hip_atomic_fetch_add(c.xcnt + (x & 7), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  }
  const int my_xcd = VAR >= 30 ? xr[0] : 0, my_rank = VAR >= 30 ? xr[1] : 0;
  const u64 t0 = wall_clock64();
  for (int it = 1; it <= iters; ++it) {
    double v[3] = {1.0, (double)(threadIdx.x & 3), (double)it * 1e-3};
    if (VAR == 0) exch_v0(c, v, (unsigned)it, smem);
    if (VAR == 1) exch_v1(c, v, (unsigned)it, smem);
    if (VAR == 2) exch_v2(c, v, (unsigned)it, smem);
    if (VAR == 3) exch_v3<4>(c, v, (unsigned)it, smem);
    if (VAR == 4) exch_v4<4, 0>(c, v, (unsigned)it, smem);
    if (VAR == 5) exch_v4<4, 1>(c, v, (unsigned)it, smem);
    if (VAR == 6) exch_v4<4, 4>(c, v, (unsigned)it, smem);
    if (VAR == 7) exch_v7<32>(c, v, (unsigned)it, smem);
    if (VAR == 8) exch_v8<4>(c, v, (unsigned)it, smem);
    if (VAR == 15) exch_v15(c, v, (unsigned)it, smem);
    if (VAR == 21) exch_v21<1>(c, v, (unsigned)it, smem);
    if (VAR == 22) exch_v21<2>(c, v, (unsigned)it, smem);
    if (VAR == 19) exch_v19<0, 4>(c, v, (unsigned)it, smem);
    if (VAR == 20) exch_v19<16, 4>(c, v, (unsigned)it, smem);
    if (VAR == 16) exch_v16<0>(c, v, (unsigned)it, smem);
    if (VAR == 17) exch_v16<8>(c, v, (unsigned)it, smem);
    if (VAR == 18) exch_v16<16>(c, v, (unsigned)it, smem);
    if (VAR == 10) exch_v10<1, true>(c, v, (unsigned)it, smem);
    if (VAR == 11) exch_v10<4, true>(c, v, (unsigned)it, smem);
    if (VAR == 12) exch_v10<12, true>(c, v, (unsigned)it, smem);
    if (VAR == 13) exch_v10<4, false>(c, v, (unsigned)it, smem);
    if (VAR == 14) exch_v10<12, false>(c, v, (unsigned)it, smem);
    if (VAR == 9) exch_v8<1>(c, v, (unsigned)it, smem);
    if (VAR == 30) exch_v30<false>(c, v, (unsigned)it, smem, my_xcd, my_rank, (int)gridDim.x / 8);
    if (VAR == 31) exch_v30<true>(c, v, (unsigned)it, smem, my_xcd, my_rank, (int)gridDim.x / 8);
    acc += v[0] + v[1] + v[2];
  }
  const u64 t1 = wall_clock64();
  if (threadIdx.x == 0) { out[blockIdx.x] = acc; ticks[blockIdx.x] = t1 - t0; }
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 2000;
  Ctl c;
  CK(hipMalloc((void**)&c.bar, 256)); CK(hipMalloc((void**)&c.parts, 3 * kMaxG * 8));
  CK(hipMalloc((void**)&c.rec, kMaxG * 64)); CK(hipMalloc((void**)&c.grp, 16 * 64)); CK(hipMalloc((void**)&c.loc, 2 * kMaxG * 64));
  CK(hipMalloc((void**)&c.xcc, kMaxG * 4)); CK(hipMalloc((void**)&c.xcnt, 64));
  double* out; u64* ticks;
  CK(hipMalloc((void**)&out, kMaxG * 8)); CK(hipMalloc((void**)&ticks, kMaxG * 8));
  const double expect = (double)G * kThreads * 1.0 + (double)G * (kThreads / 4) * 6.0;
  for (int var = 0; var < 32; ++var) {
    if (var > 22 && var < 30) continue;
    if (var >= 30 && G % 8 != 0) continue;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemset(c.bar, 0, 256)); CK(hipMemset(c.rec, 0, kMaxG * 64)); CK(hipMemset(c.grp, 0, 16 * 64)); CK(hipMemset(c.loc, 0, 2 * kMaxG * 64)); CK(hipMemset(c.xcnt, 0, 64));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      if (var == 0) bench<0><<<G, kThreads>>>(c, iters, out, ticks);
      if (var == 1) bench<1><<<G, kThreads>>>(c, iters, out, ticks);
      if (var == 2) bench<2><<<G, kThreads>>>(c, iters, out, ticks);
      if (var == 3) bench<3><<<G, kThreads>>>(c, iters, out, ticks);
      if (var == 4) bench<4><<<G, kThreads>>>(c, iters, out, ticks);
      if (var == 5) bench<5><<<G, kThreads>>>(c, iters, out, ticks);
      if (var == 6) bench<6><<<G, kThreads>>>(c, iters, out, ticks);
      if (var == 7) bench<7><<<G, kThreads>>>(c, iters, out, ticks);
      if (var == 8) bench<8><<<G, kThreads>>>(c, iters, out, ticks);
      if (var == 15) bench<15><<<G, kThreads>>>(c, iters, out, ticks);
      if (var == 21) bench<21><<<G, kThreads>>>(c, iters, out, ticks);
      if (var == 22) bench<22><<<G, kThreads>>>(c, iters, out, ticks);
      if (var == 19) bench<19><<<G, kThreads>>>(c, iters, out, ticks);
      if (var == 20) bench<20><<<G, kThreads>>>(c, iters, out, ticks);
      if (var == 16) bench<16><<<G, kThreads>>>(c, iters, out, ticks);
      if (var == 17) bench<17><<<G, kThreads>>>(c, iters, out, ticks);
      if (var == 18) bench<18><<<G, kThreads>>>(c, iters, out, ticks);
      if (var == 10) bench<10><<<G, kThreads>>>(c, iters, out, ticks);
      if (var == 11) bench<11><<<G, kThreads>>>(c, iters, out, ticks);
      if (var == 12) bench<12><<<G, kThreads>>>(c, iters, out, ticks);
      if (var == 13) bench<13><<<G, kThreads>>>(c, iters, out, ticks);
      if (var == 14) bench<14><<<G, kThreads>>>(c, iters, out, ticks);
      if (var == 30) bench<30><<<G, kThreads>>>(c, iters, out, ticks);
      if (var == 31) bench<31><<<G, kThreads>>>(c, iters, out, ticks);
      if (var == 9 && G <= 64) bench<9><<<G, kThreads>>>(c, iters, out, ticks);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<double> h(G); CK(hipMemcpy(h.data(), out, G * 8, hipMemcpyDeviceToHost));
      double want = 0; for (int it = 1; it <= iters; ++it) want += expect + (double)G * kThreads * it * 1e-3;
      bool same = true; for (int b = 1; b < G; ++b) same = same && (h[b] == h[0]);
      printf("variant %d G=%d: %.3f us per exchange; acc[0]=%.6e want %.6e identical across workgroups: %d\n", var, G, 1e3 * ms / iters, h[0], want, (int)same);
    }
  }
  std::vector<int> xc(G); CK(hipMemcpy(xc.data(), c.xcc, G * 4, hipMemcpyDeviceToHost));
  printf("XCC id of blocks 0..31:"); for (int b = 0; b < 32 && b < G; ++b) printf(" %d", xc[b]); printf("\n");
  int mism = 0; for (int b = 0; b < G; ++b) mism += (xc[b] != xc[b % 8]); printf("blocks whose XCC differs from block (b %% 8): %d\n", mism);
  return 0;
}
