// pmc_calibrate.hip — known byte counts in this path's access patterns, to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on
// gfx950 (MI355X_MICROARCH.md §HBM: FETCH_SIZE reports half the bytes of a 16-B-per-lane stream; "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern").  Each kernel moves exactly BYTES bytes of a buffer
// far larger than the 256 MiB Infinity Cache, once:
//   read8 / write8     8 B per lane, coalesced (the wavefront path's SoA f64 queue rows)
//   read16 / write16   16 B per lane, coalesced (the guide's reference pattern)
//   write24s           three 8-B stores per lane at a 24-B stride (the one-kernel path's rgb[3q .. 3q+2] framebuffer write)
//   gather128          one 128-B record per lane at a random index (node / intersection-record fetches that miss the caches)
// build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/pmc_calibrate scripts/pmc_calibrate.hip ; run under rocprofv3 --pmc.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void read8(const double* __restrict__ p, size_t n, double* out) {
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 12345.678) *out = acc;
}
__global__ void write8(double* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)i;
}
__global__ void read16(const double2* __restrict__ p, size_t n, double* out) {
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = p[i]; acc += v.x + v.y; }
  if (acc == 12345.678) *out = acc;
}
__global__ void write16(double2* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_double2((double)i, 1.0);
}
__global__ void write24s(double* __restrict__ p, size_t npx) {
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < npx; q += (size_t)gridDim.x * blockDim.x) { p[3 * q] = 1.0; p[3 * q + 1] = 2.0; p[3 * q + 2] = 3.0; }
}
__global__ void gather128(const double2* __restrict__ p, size_t nrec, size_t n_fetch, double* out) {
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_fetch; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = (i * 0x9E3779B97F4A7C15ull) % nrec;  // every record at most a few times; the table is 8x the Infinity Cache
    const double2* q = p + r * 8;
    double2 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4], f = q[5], g = q[6], h = q[7];
    acc += a.x + b.x + c.x + d.x + e.x + f.x + g.x + h.x;
  }
  if (acc == 12345.678) *out = acc;
}

int main() {
  const size_t BYTES = 2ull << 30;
  void* buf = nullptr;
  double* out = nullptr;
  CHECK(hipMalloc(&buf, BYTES));
  CHECK(hipMalloc((void**)&out, 8));
  CHECK(hipMemset(buf, 0, BYTES));
  CHECK(hipDeviceSynchronize());
  const dim3 grid(256 * 8), block(256);
  hipLaunchKernelGGL(read8, grid, block, 0, 0, (const double*)buf, BYTES / 8, out);
  hipLaunchKernelGGL(write8, grid, block, 0, 0, (double*)buf, BYTES / 8);
  hipLaunchKernelGGL(read16, grid, block, 0, 0, (const double2*)buf, BYTES / 16, out);
  hipLaunchKernelGGL(write16, grid, block, 0, 0, (double2*)buf, BYTES / 16);
  hipLaunchKernelGGL(write24s, grid, block, 0, 0, (double*)buf, BYTES / 24);
  hipLaunchKernelGGL(gather128, grid, block, 0, 0, (const double2*)buf, BYTES / 128, BYTES / 128 / 4, out);
  CHECK(hipDeviceSynchronize());
  std::printf("{\"bytes\": {\"read8\": %zu, \"write8\": %zu, \"read16\": %zu, \"write16\": %zu, \"write24s\": %zu, \"gather128\": %zu}}\n", BYTES, BYTES, BYTES, BYTES,
              BYTES / 24 * 24, BYTES / 128 / 4 * 128);
  return 0;
}
