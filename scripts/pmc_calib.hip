// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on the access patterns of cg_persist (MI355X_MICROARCH.md, HBM section:
// "calibrate on a known byte count in your own access pattern"): known byte counts per launch, streamed with
//   read8   8 B per lane  (raw buffer_load_b64, the float2 coefficient rows)        read16  16 B per lane (buffer_load_b128)
//   read8a  8 B per lane at agent scope (sc1, the perimeter loads)
//   write8a 8 B per lane write-through at agent scope (sc1, the perimeter stores)    write16 16 B per lane plain stores
// Build: hipcc --offload-arch=gfx950 -O3 scripts/pmc_calib.hip -o scripts/_bin/pmc_calib ; run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- scripts/_bin/pmc_calib     (and once more with WRITE_SIZE)
// Each kernel moves BYTES = 256 MiB per launch (printed); the counters' mean per dispatch / BYTES is the calibration factor.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
constexpr int kAgent = 16;

template <int AUX>
__global__ void read8(const float* src, float* sink, size_t n8) {   // n8 = number of 8-byte elements
  const rsrc_t R = make_rsrc(src, (unsigned)(n8 * 8));
  float acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const auto t = __builtin_amdgcn_raw_buffer_load_b64(R, (unsigned)(i * 8), 0, AUX);
    float f[2];
    __builtin_memcpy(f, &t, 8);
    acc += f[0] + f[1];
  }
  if (acc == 12345.678f) sink[0] = acc;
}
__global__ void read16(const float* src, float* sink, size_t n16) {
  const rsrc_t R = make_rsrc(src, (unsigned)(n16 * 16));
  float acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const auto t = __builtin_amdgcn_raw_buffer_load_b128(R, (unsigned)(i * 16), 0, 0);
    float f[4];
    __builtin_memcpy(f, &t, 16);
    acc += f[0] + f[1] + f[2] + f[3];
  }
  if (acc == 12345.678f) sink[0] = acc;
}
__global__ void write8a(float* dst, size_t n8) {
  const rsrc_t R = make_rsrc(dst, (unsigned)(n8 * 8));
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    __attribute__((ext_vector_type(2))) unsigned int t = {(unsigned)i, 7u};
    __builtin_amdgcn_raw_buffer_store_b64(t, R, (unsigned)(i * 8), 0, kAgent);
  }
}
__global__ void write16(float* dst, size_t n16) {
  const rsrc_t R = make_rsrc(dst, (unsigned)(n16 * 16));
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    __attribute__((ext_vector_type(4))) unsigned int t = {(unsigned)i, 7u, 8u, 9u};
    __builtin_amdgcn_raw_buffer_store_b128(t, R, (unsigned)(i * 16), 0, 0);
  }
}

int main() {
  const size_t bytes = (size_t)256 << 20;
  float *a, *sink;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(a, 0, bytes);
  for (int rep = 0; rep < 3; ++rep) {
    read8<0><<<2048, 256>>>(a, sink, bytes / 8);
    read8<kAgent><<<2048, 256>>>(a, sink, bytes / 8);
    read16<<<2048, 256>>>(a, sink, bytes / 16);
    write8a<<<2048, 256>>>(a, bytes / 8);
    write16<<<2048, 256>>>(a, bytes / 16);
  }
  hipDeviceSynchronize();
  printf("BYTES per launch of every kernel: %zu\n", bytes);
  return 0;
}
