#include <hip/hip_runtime.h>
__device__ __forceinline__ double xor32_swap(double c) {
  unsigned long long b = (unsigned long long)__double_as_longlong(c);
  unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
  auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  double a = __longlong_as_double((long long)(((unsigned long long)r1[0] << 32) | r0[0]));
  double d = __longlong_as_double((long long)(((unsigned long long)r1[1] << 32) | r0[1]));
  return a + d;
}
__device__ __forceinline__ double xor16_swap(double c) {
  unsigned long long b = (unsigned long long)__double_as_longlong(c);
  unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
  auto r0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto r1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  double a = __longlong_as_double((long long)(((unsigned long long)r1[0] << 32) | r0[0]));
  double d = __longlong_as_double((long long)(((unsigned long long)r1[1] << 32) | r0[1]));
  return a + d;
}
__global__ void k(const double* in, double* out) {
  double c = in[threadIdx.x];
  out[threadIdx.x] = xor32_swap(xor16_swap(c));
  out[64 + threadIdx.x] = (c + __shfl_xor(c, 16, 64));
  double e = c + __shfl_xor(c, 16, 64);
  out[128 + threadIdx.x] = e + __shfl_xor(e, 32, 64);
}
int main() {
  double h[64], o[192]; for (int i = 0; i < 64; ++i) h[i] = 1.0 / (i + 3) + i * 1e-7;
  double *d, *e; hipMalloc(&d, sizeof(h)); hipMalloc(&e, sizeof(o)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, e); hipMemcpy(o, e, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 64; ++i) if (o[i] != o[128 + i]) ++bad;
  printf("mismatches %d  (%.17g vs %.17g)\n", bad, o[5], o[133]);
  return bad != 0;
}
