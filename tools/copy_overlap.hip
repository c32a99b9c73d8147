// tools/copy_overlap.hip -- does a pinned host->device copy run beside a kernel that fills every CU?  (PCIe-inclusive
// streaming)  Times a 1 GiB hipMemcpyAsync alone, then while a long VALU-bound kernel occupies the chip.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
__global__ void spin(uint32_t* out, int iters) {
  uint32_t x = threadIdx.x + blockIdx.x * 977u;
  for (int i = 0; i < iters; ++i) x = x * 1664525u + 1013904223u;
  if (x == 0x12345u) out[0] = x;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t bytes = (size_t)1 << 30;
  void *h, *d; uint32_t* o;
  if (hipHostMalloc(&h, bytes, hipHostMallocDefault) != hipSuccess || hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&o, 4) != hipSuccess) return 1;
  hipStream_t s1, s2;
  (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  (void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s2); (void)hipStreamSynchronize(s2);  // warm
  double t0 = now();
  (void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s2); (void)hipStreamSynchronize(s2);
  const double alone = now() - t0;
  hipLaunchKernelGGL(spin, dim3(256 * 64), dim3(256), 0, s1, o, 1 << 16); (void)hipStreamSynchronize(s1);
  t0 = now();
  hipLaunchKernelGGL(spin, dim3(256 * 64), dim3(256), 0, s1, o, 1 << 16); (void)hipStreamSynchronize(s1);
  const double kernel = now() - t0;
  t0 = now();
  hipLaunchKernelGGL(spin, dim3(256 * 64), dim3(256), 0, s1, o, 1 << 16);
  (void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s2);
  (void)hipStreamSynchronize(s2);
  const double copy_beside = now() - t0;
  (void)hipStreamSynchronize(s1);
  const double both = now() - t0;
  printf("copy alone %.2f ms (%.1f GB/s); kernel alone %.2f ms; copy beside the kernel done after %.2f ms; both done after %.2f ms\n",
         alone * 1e3, bytes / alone / 1e9, kernel * 1e3, copy_beside * 1e3, both * 1e3);
  return 0;
}
