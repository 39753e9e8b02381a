// Issue rate of fp64 / fp32 VALU instructions on one CU (diagnostics): N independent FMA chains per lane, W waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/_bin/fp64_rate scripts/fp64_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

template <typename T, int CHAINS, int OP>
__global__ void rate(T* out, int iters, unsigned long long* cycles) {
  T a[CHAINS];
  for (int c = 0; c < CHAINS; ++c) a[c] = (T)(threadIdx.x + c);
  const T m = (T)1.0000001, b = (T)1e-9;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      if (OP == 0) a[c] = __builtin_fma(a[c], m, b);
      if (OP == 1) a[c] = a[c] + b;
      if (OP == 2) { float f = (float)threadIdx.x + c; asm volatile("" : "+v"(f)); a[c] = a[c] + (T)f; }          // v_cvt_f64_f32 + v_add_f64
      if (OP == 3) { int lo = (int)i + c; asm volatile("" : "+v"(lo)); lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, false); a[c] = a[c] + (T)__int_as_float(lo & 0x3f800000); }   // + wave_shr DPP (+ and, cvt)
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  T s = 0;
  for (int c = 0; c < CHAINS; ++c) s += a[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <typename T, int CHAINS, int OP>
int run(const char* name, int threads) {
  T* out; unsigned long long* cyc;
  CK(hipMalloc((void**)&out, 256 * 1024 * sizeof(T))); CK(hipMalloc((void**)&cyc, 1024 * 8));
  const int iters = 20000;
  rate<T, CHAINS, OP><<<256, threads>>>(out, iters, cyc);
  CK(hipDeviceSynchronize());
  unsigned long long h[256]; CK(hipMemcpy(h, cyc, 256 * 8, hipMemcpyDeviceToHost));
  const double per_instr = (double)h[0] / ((double)iters * CHAINS);
  const int waves_per_simd = threads / 256;
  printf("%-28s %d chains, %d wave(s)/SIMD: %.2f shader cycles per instruction per wave -> %.2f cycles of SIMD time per wave-instruction\n",
         name, CHAINS, waves_per_simd, per_instr, per_instr / waves_per_simd);
  CK(hipFree(out)); CK(hipFree(cyc));
  return 0;
}

int main() {
  run<double, 8, 0>("v_fma_f64", 256); run<double, 8, 0>("v_fma_f64", 512);
  run<double, 2, 0>("v_fma_f64", 512); run<double, 1, 0>("v_fma_f64", 512);
  run<double, 8, 1>("v_add_f64", 512); run<double, 1, 1>("v_add_f64", 512);
  run<float, 8, 0>("v_fma_f32", 512); run<float, 1, 0>("v_fma_f32", 512);
  run<double, 8, 2>("v_cvt_f64_f32 + v_add_f64", 512); run<double, 8, 2>("v_cvt_f64_f32 + v_add_f64", 256);
  run<double, 8, 3>("dpp + and + cvt + add", 512);
  return 0;
}
