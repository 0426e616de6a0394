// GPU box: worst relative error of V_RCP_F64 (__builtin_amdgcn_rcp) over 2^26 operands spread over the exponent range.
// build: hipcc --offload-arch=gfx950 -O2 scripts/rcp_accuracy.hip -o gpurun_out/rcp_accuracy    (group_box_hit and cube_shadow_approx budget 2^-23)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(double* worst, unsigned long long n) {
  unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
  double w = 0.0;
  for (; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
    unsigned long long h = i * 0x9E3779B97F4A7C15ull;
    h ^= h >> 31; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 29;
    const double mant = 1.0 + (double)(h >> 12) * (1.0 / 4503599627370496.0);
    const int e = (int)((h & 0xfff) % 600) - 300;
    double x = ldexp(mant, e);
    if (h & 0x1000) x = -x;
    const double a = __builtin_amdgcn_rcp(x), b = 1.0 / x;
    const double rel = fabs(a - b) / fabs(b);
    if (rel > w) w = rel;
  }
  atomicMax((unsigned long long*)worst, __double_as_longlong(w));  // (non-negative doubles order like their bits)
}
int main() {
  double* d; double h = 0.0;
  hipMalloc(&d, 8); hipMemcpy(d, &h, 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, d, 1ull << 26);
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("V_RCP_F64: worst relative error over 2^26 operands = %.3e = 2^%.2f\n", h, log2(h));
  return 0;
}
