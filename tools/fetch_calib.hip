// tools/fetch_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access shapes of the alignment kernels.
// MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide (16 B per lane) coalesced streaming
// read; "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  This program
// reads / writes a known number of bytes (1 GiB per kernel, far beyond the Infinity Cache) in three shapes:
//   read16  : 16 bytes per lane, consecutive lanes consecutive (the text / pattern loads)
//   read4   : 4 bytes per lane, consecutive lanes consecutive  (the boundary planes of k_myers_band: [group][plane][thread])
//   write4  : 4 bytes per lane                                  (its plane stores)
//   write16 : 16 bytes per lane
// Run:  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib   (and once more with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void read16(const uint4* __restrict__ p, size_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void read4(const uint32_t* __restrict__ p, size_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void write4(uint32_t* __restrict__ p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
__global__ void write16(uint4* __restrict__ p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  void *a, *b;
  uint32_t* sink;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
  hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 2; ++rep) {  // alternate the buffers so that nothing is served from the Infinity Cache
    hipLaunchKernelGGL(read16, dim3(4096), dim3(256), 0, 0, (const uint4*)a, bytes / 16, sink);
    hipLaunchKernelGGL(read4, dim3(4096), dim3(256), 0, 0, (const uint32_t*)b, bytes / 4, sink);
    hipLaunchKernelGGL(write4, dim3(4096), dim3(256), 0, 0, (uint32_t*)a, bytes / 4);
    hipLaunchKernelGGL(write16, dim3(4096), dim3(256), 0, 0, (uint4*)b, bytes / 16);
  }
  hipDeviceSynchronize();
  printf("each kernel moves %zu bytes\n", bytes);
  return 0;
}
