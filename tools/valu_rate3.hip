// tools/valu_rate3.hip -- issue rates of the instructions of K3l's packed protein cell on gfx950 (v_max3_u32, v_max_u32,
// v_add3_u32, v_and_or_b32 against v_add_u32), independent blocks of eight, and of the cell's own mix as a dependent
// chain down a strip (add, add, add, max3, bitop3 per cell; the same with two v_max_u32).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate3 valu_rate3.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define OPERANDS : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c)

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int iters) {
  uint32_t a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1 + i * 977);
  uint32_t b = seed ^ 0x9e3779b9u, c = seed * 31u + threadIdx.x, t = seed + threadIdx.x, u = seed * 7u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (OP == 0) {
#define X(i) "v_add_u32_e32 %" #i ", %8, %" #i "\n"
        asm volatile(R8(X) OPERANDS);
#undef X
      } else if (OP == 1) {
#define X(i) "v_max_u32_e32 %" #i ", %8, %" #i "\n"
        asm volatile(R8(X) OPERANDS);
#undef X
      } else if (OP == 2) {
#define X(i) "v_max3_u32 %" #i ", %" #i ", %8, %9\n"
        asm volatile(R8(X) OPERANDS);
#undef X
      } else if (OP == 3) {
#define X(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
        asm volatile(R8(X) OPERANDS);
#undef X
      } else if (OP == 4) {
#define X(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
        asm volatile(R8(X) OPERANDS);
#undef X
      } else if (OP == 5) {  // max3 with two of its sources the same register (one register-file read less)
#define X(i) "v_max3_u32 %" #i ", %" #i ", %8, %8\n"
        asm volatile(R8(X) OPERANDS);
#undef X
      } else if (OP == 6) {  // the cell, 8 rows down a strip: t = address, x = diag term, y = horizontal term, max3, bitop3 (5 per cell)
#define X(i) "v_add_u32_e32 %10, %8, %" #i "\n v_add_u32_e32 %11, %9, %10\n v_add_u32_e32 %10, %" #i ", %9\n v_max3_u32 %11, %11, %8, %10\n v_bitop3_b32 %" #i ", %11, %8, %9 bitop3:0xea\n"
        asm volatile(R8(X) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(b), "+v"(c), "+v"(t), "+v"(u));
#undef X
      } else if (OP == 7) {  // the same with two v_max_u32 (6 per cell)
#define X(i) "v_add_u32_e32 %10, %8, %" #i "\n v_add_u32_e32 %11, %9, %10\n v_add_u32_e32 %10, %" #i ", %9\n v_max_u32_e32 %11, %11, %8\n v_max_u32_e32 %11, %11, %10\n v_bitop3_b32 %" #i ", %11, %8, %9 bitop3:0xea\n"
        asm volatile(R8(X) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(b), "+v"(c), "+v"(t), "+v"(u));
#undef X
      } else if (OP == 8) {  // the cell without its max3 (4 per cell): what the other four cost
#define X(i) "v_add_u32_e32 %10, %8, %" #i "\n v_add_u32_e32 %11, %9, %10\n v_add_u32_e32 %10, %" #i ", %9\n v_bitop3_b32 %" #i ", %11, %8, %10 bitop3:0xea\n"
        asm volatile(R8(X) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(b), "+v"(c), "+v"(t), "+v"(u));
#undef X
      }
    }
    b += 0x01010101u; c ^= b;
  }
  uint32_t s = t ^ u;
#pragma unroll
  for (int i = 0; i < 8; ++i) s ^= a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, uint32_t* d_out, int blocks_per_cu, int iters, double per_block) {
  const int blocks = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, 16);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double wave_instr = (double)blocks * 4 * iters * per_block;
  printf("%-40s %d waves/SIMD %8.3f ms  %.3f wave-instr per SIMD per ns\n", name, blocks_per_cu, ms, wave_instr / ms / 1e6 / 1024);
}

int main() {
  uint32_t* d_out;
  (void)hipMalloc(&d_out, 256 * 8 * 256 * 4);
  for (int w : {4, 8}) {
    run<0>("v_add_u32_e32", d_out, w, 20000, 64);
    run<1>("v_max_u32_e32", d_out, w, 20000, 64);
    run<2>("v_max3_u32 (3 vgpr)", d_out, w, 20000, 64);
    run<5>("v_max3_u32 (2 distinct vgpr)", d_out, w, 20000, 64);
    run<3>("v_add3_u32", d_out, w, 20000, 64);
    run<4>("v_and_or_b32", d_out, w, 20000, 64);
    run<6>("cell: 3 add, max3, bitop3", d_out, w, 20000, 64 * 5);
    run<7>("cell: 3 add, 2 max, bitop3", d_out, w, 20000, 64 * 6);
    run<8>("cell without the max3", d_out, w, 20000, 64 * 4);
  }
  return 0;
}
