// tools/valu_order.hip -- K3l's cell costs more than the sum of its instructions' issue slots (valu_rate3/4: four full-rate
// ones 4.1 ns, the half-rate v_max3_u32 alone 1.85 ns, together 8 ns per wavefront).  Does the ORDER of the instructions
// matter?  Eight cells per block; "chain" = the maximum of cell i takes the bitop3 result of cell i-1 (the strip's up value).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_order valu_order.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
// operands: %0..%7 a[i] (the left cells, updated), %8..%15 d[i] (diag terms), %16..%23 h[i] (horizontal terms), %24 b, %25 c, %26 up, %27 t
#define OPS : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), \
              "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]), \
              "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(h[4]), "+v"(h[5]), "+v"(h[6]), "+v"(h[7]), "+v"(b), "+v"(c), "+v"(up), "+v"(t)
#define S2(x) #x
#define S(x) S2(x)
#define ADDS(i, di, hi) "v_add_u32_e32 %27, %24, %" #i "\n v_add_u32_e32 %" #di ", %25, %27\n v_add_u32_e32 %" #hi ", %" #i ", %25\n"

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int iters) {
  uint32_t a[8], d[8], h[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = seed * (threadIdx.x + 1 + i * 977); d[i] = a[i] * 3u; h[i] = a[i] * 5u; }
  uint32_t b = seed ^ 0x9e3779b9u, c = seed * 31u + threadIdx.x, up = seed, t = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (OP == 0) {  // per cell: A A A M B, cells independent
        asm volatile(ADDS(0, 8, 16) "v_max3_u32 %8, %8, %24, %16\n v_bitop3_b32 %0, %8, %24, %25 bitop3:0xea\n"
                     ADDS(1, 9, 17) "v_max3_u32 %9, %9, %24, %17\n v_bitop3_b32 %1, %9, %24, %25 bitop3:0xea\n"
                     ADDS(2, 10, 18) "v_max3_u32 %10, %10, %24, %18\n v_bitop3_b32 %2, %10, %24, %25 bitop3:0xea\n"
                     ADDS(3, 11, 19) "v_max3_u32 %11, %11, %24, %19\n v_bitop3_b32 %3, %11, %24, %25 bitop3:0xea\n"
                     ADDS(4, 12, 20) "v_max3_u32 %12, %12, %24, %20\n v_bitop3_b32 %4, %12, %24, %25 bitop3:0xea\n"
                     ADDS(5, 13, 21) "v_max3_u32 %13, %13, %24, %21\n v_bitop3_b32 %5, %13, %24, %25 bitop3:0xea\n"
                     ADDS(6, 14, 22) "v_max3_u32 %14, %14, %24, %22\n v_bitop3_b32 %6, %14, %24, %25 bitop3:0xea\n"
                     ADDS(7, 15, 23) "v_max3_u32 %15, %15, %24, %23\n v_bitop3_b32 %7, %15, %24, %25 bitop3:0xea\n" OPS);
      } else if (OP == 1) {  // per cell A A A M B, chained through up (as in the kernel)
        asm volatile(ADDS(0, 8, 16) "v_max3_u32 %8, %8, %26, %16\n v_bitop3_b32 %26, %8, %24, %25 bitop3:0xea\n v_mov_b32 %0, %26\n"
                     ADDS(1, 9, 17) "v_max3_u32 %9, %9, %26, %17\n v_bitop3_b32 %26, %9, %24, %25 bitop3:0xea\n v_mov_b32 %1, %26\n"
                     ADDS(2, 10, 18) "v_max3_u32 %10, %10, %26, %18\n v_bitop3_b32 %26, %10, %24, %25 bitop3:0xea\n v_mov_b32 %2, %26\n"
                     ADDS(3, 11, 19) "v_max3_u32 %11, %11, %26, %19\n v_bitop3_b32 %26, %11, %24, %25 bitop3:0xea\n v_mov_b32 %3, %26\n"
                     ADDS(4, 12, 20) "v_max3_u32 %12, %12, %26, %20\n v_bitop3_b32 %26, %12, %24, %25 bitop3:0xea\n v_mov_b32 %4, %26\n"
                     ADDS(5, 13, 21) "v_max3_u32 %13, %13, %26, %21\n v_bitop3_b32 %26, %13, %24, %25 bitop3:0xea\n v_mov_b32 %5, %26\n"
                     ADDS(6, 14, 22) "v_max3_u32 %14, %14, %26, %22\n v_bitop3_b32 %26, %14, %24, %25 bitop3:0xea\n v_mov_b32 %6, %26\n"
                     ADDS(7, 15, 23) "v_max3_u32 %15, %15, %26, %23\n v_bitop3_b32 %26, %15, %24, %25 bitop3:0xea\n v_mov_b32 %7, %26\n" OPS);
      } else if (OP == 2) {  // all 24 adds, then M B per cell (independent cells)
        asm volatile(ADDS(0, 8, 16) ADDS(1, 9, 17) ADDS(2, 10, 18) ADDS(3, 11, 19) ADDS(4, 12, 20) ADDS(5, 13, 21) ADDS(6, 14, 22) ADDS(7, 15, 23)
                     "v_max3_u32 %8, %8, %24, %16\n v_bitop3_b32 %0, %8, %24, %25 bitop3:0xea\n"
                     "v_max3_u32 %9, %9, %24, %17\n v_bitop3_b32 %1, %9, %24, %25 bitop3:0xea\n"
                     "v_max3_u32 %10, %10, %24, %18\n v_bitop3_b32 %2, %10, %24, %25 bitop3:0xea\n"
                     "v_max3_u32 %11, %11, %24, %19\n v_bitop3_b32 %3, %11, %24, %25 bitop3:0xea\n"
                     "v_max3_u32 %12, %12, %24, %20\n v_bitop3_b32 %4, %12, %24, %25 bitop3:0xea\n"
                     "v_max3_u32 %13, %13, %24, %21\n v_bitop3_b32 %5, %13, %24, %25 bitop3:0xea\n"
                     "v_max3_u32 %14, %14, %24, %22\n v_bitop3_b32 %6, %14, %24, %25 bitop3:0xea\n"
                     "v_max3_u32 %15, %15, %24, %23\n v_bitop3_b32 %7, %15, %24, %25 bitop3:0xea\n" OPS);
      } else if (OP == 3) {  // all 24 adds, 8 M, 8 B (independent cells)
        asm volatile(ADDS(0, 8, 16) ADDS(1, 9, 17) ADDS(2, 10, 18) ADDS(3, 11, 19) ADDS(4, 12, 20) ADDS(5, 13, 21) ADDS(6, 14, 22) ADDS(7, 15, 23)
                     "v_max3_u32 %8, %8, %24, %16\n v_max3_u32 %9, %9, %24, %17\n v_max3_u32 %10, %10, %24, %18\n v_max3_u32 %11, %11, %24, %19\n"
                     "v_max3_u32 %12, %12, %24, %20\n v_max3_u32 %13, %13, %24, %21\n v_max3_u32 %14, %14, %24, %22\n v_max3_u32 %15, %15, %24, %23\n"
                     "v_bitop3_b32 %0, %8, %24, %25 bitop3:0xea\n v_bitop3_b32 %1, %9, %24, %25 bitop3:0xea\n v_bitop3_b32 %2, %10, %24, %25 bitop3:0xea\n"
                     "v_bitop3_b32 %3, %11, %24, %25 bitop3:0xea\n v_bitop3_b32 %4, %12, %24, %25 bitop3:0xea\n v_bitop3_b32 %5, %13, %24, %25 bitop3:0xea\n"
                     "v_bitop3_b32 %6, %14, %24, %25 bitop3:0xea\n v_bitop3_b32 %7, %15, %24, %25 bitop3:0xea\n" OPS);
      } else if (OP == 4) {  // all 24 adds, then the chain M B M B ... through up, the stored cell IS the chain register's copy (bitop3 writes a[i], next max3 reads a[i])
        asm volatile(ADDS(0, 8, 16) ADDS(1, 9, 17) ADDS(2, 10, 18) ADDS(3, 11, 19) ADDS(4, 12, 20) ADDS(5, 13, 21) ADDS(6, 14, 22) ADDS(7, 15, 23)
                     "v_max3_u32 %8, %8, %26, %16\n v_bitop3_b32 %0, %8, %24, %25 bitop3:0xea\n"
                     "v_max3_u32 %9, %9, %0, %17\n v_bitop3_b32 %1, %9, %24, %25 bitop3:0xea\n"
                     "v_max3_u32 %10, %10, %1, %18\n v_bitop3_b32 %2, %10, %24, %25 bitop3:0xea\n"
                     "v_max3_u32 %11, %11, %2, %19\n v_bitop3_b32 %3, %11, %24, %25 bitop3:0xea\n"
                     "v_max3_u32 %12, %12, %3, %20\n v_bitop3_b32 %4, %12, %24, %25 bitop3:0xea\n"
                     "v_max3_u32 %13, %13, %4, %21\n v_bitop3_b32 %5, %13, %24, %25 bitop3:0xea\n"
                     "v_max3_u32 %14, %14, %5, %22\n v_bitop3_b32 %6, %14, %24, %25 bitop3:0xea\n"
                     "v_max3_u32 %15, %15, %6, %23\n v_bitop3_b32 %7, %15, %24, %25 bitop3:0xea\n v_mov_b32 %26, %7\n" OPS);
      } else if (OP == 5) {  // the chain with the adds of the NEXT cell between M and B (the dependent B does not follow its M directly)
        asm volatile(ADDS(0, 8, 16)
                     "v_max3_u32 %8, %8, %26, %16\n" ADDS(1, 9, 17) "v_bitop3_b32 %0, %8, %24, %25 bitop3:0xea\n"
                     "v_max3_u32 %9, %9, %0, %17\n" ADDS(2, 10, 18) "v_bitop3_b32 %1, %9, %24, %25 bitop3:0xea\n"
                     "v_max3_u32 %10, %10, %1, %18\n" ADDS(3, 11, 19) "v_bitop3_b32 %2, %10, %24, %25 bitop3:0xea\n"
                     "v_max3_u32 %11, %11, %2, %19\n" ADDS(4, 12, 20) "v_bitop3_b32 %3, %11, %24, %25 bitop3:0xea\n"
                     "v_max3_u32 %12, %12, %3, %20\n" ADDS(5, 13, 21) "v_bitop3_b32 %4, %12, %24, %25 bitop3:0xea\n"
                     "v_max3_u32 %13, %13, %4, %21\n" ADDS(6, 14, 22) "v_bitop3_b32 %5, %13, %24, %25 bitop3:0xea\n"
                     "v_max3_u32 %14, %14, %5, %22\n" ADDS(7, 15, 23) "v_bitop3_b32 %6, %14, %24, %25 bitop3:0xea\n"
                     "v_max3_u32 %15, %15, %6, %23\n v_bitop3_b32 %7, %15, %24, %25 bitop3:0xea\n v_mov_b32 %26, %7\n" OPS);
      } else if (OP == 6) {  // the chain, one add between M and B and two between B and the next M
        asm volatile("v_add_u32_e32 %27, %24, %0\n v_add_u32_e32 %8, %25, %27\n v_add_u32_e32 %16, %0, %25\n"
                     "v_max3_u32 %8, %8, %26, %16\n v_add_u32_e32 %27, %24, %1\n v_bitop3_b32 %0, %8, %24, %25 bitop3:0xea\n v_add_u32_e32 %9, %25, %27\n v_add_u32_e32 %17, %1, %25\n"
                     "v_max3_u32 %9, %9, %0, %17\n v_add_u32_e32 %27, %24, %2\n v_bitop3_b32 %1, %9, %24, %25 bitop3:0xea\n v_add_u32_e32 %10, %25, %27\n v_add_u32_e32 %18, %2, %25\n"
                     "v_max3_u32 %10, %10, %1, %18\n v_add_u32_e32 %27, %24, %3\n v_bitop3_b32 %2, %10, %24, %25 bitop3:0xea\n v_add_u32_e32 %11, %25, %27\n v_add_u32_e32 %19, %3, %25\n"
                     "v_max3_u32 %11, %11, %2, %19\n v_add_u32_e32 %27, %24, %4\n v_bitop3_b32 %3, %11, %24, %25 bitop3:0xea\n v_add_u32_e32 %12, %25, %27\n v_add_u32_e32 %20, %4, %25\n"
                     "v_max3_u32 %12, %12, %3, %20\n v_add_u32_e32 %27, %24, %5\n v_bitop3_b32 %4, %12, %24, %25 bitop3:0xea\n v_add_u32_e32 %13, %25, %27\n v_add_u32_e32 %21, %5, %25\n"
                     "v_max3_u32 %13, %13, %4, %21\n v_add_u32_e32 %27, %24, %6\n v_bitop3_b32 %5, %13, %24, %25 bitop3:0xea\n v_add_u32_e32 %14, %25, %27\n v_add_u32_e32 %22, %6, %25\n"
                     "v_max3_u32 %14, %14, %5, %22\n v_add_u32_e32 %27, %24, %7\n v_bitop3_b32 %6, %14, %24, %25 bitop3:0xea\n v_add_u32_e32 %15, %25, %27\n v_add_u32_e32 %23, %7, %25\n"
                     "v_max3_u32 %15, %15, %6, %23\n v_bitop3_b32 %7, %15, %24, %25 bitop3:0xea\n v_mov_b32 %26, %7\n" OPS);
      }
    }
    b += 0x01010101u; c ^= b;
  }
  uint32_t s = t ^ up;
#pragma unroll
  for (int i = 0; i < 8; ++i) s ^= a[i] ^ d[i] ^ h[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, uint32_t* d_out, int waves) {
  const int blocks = 256 * waves, iters = 10000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, 16);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double cell_waves_per_simd = (double)waves * iters * 4 * 8;
  printf("%-64s %d waves/SIMD %8.3f ms  %5.2f ns per cell per SIMD\n", name, waves, ms, ms * 1e6 / cell_waves_per_simd);
}

int main() {
  uint32_t* d_out;
  (void)hipMalloc(&d_out, 256 * 8 * 256 * 4);
  for (int w : {4, 2, 1}) {
    run<0>("0 per cell A A A M B, independent cells", d_out, w);
    run<1>("1 per cell A A A M B mov, chained (the kernel)", d_out, w);
    run<2>("2 24 adds, then M B per cell, independent", d_out, w);
    run<3>("3 24 adds, 8 M, 8 B, independent", d_out, w);
    run<4>("4 24 adds, then the chain M B M B", d_out, w);
    run<5>("5 chain, next cell's three adds between M and B", d_out, w);
    run<6>("6 chain, one add between M and B, two between B and M", d_out, w);
  }
  return 0;
}
