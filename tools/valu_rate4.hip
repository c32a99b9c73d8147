// tools/valu_rate4.hip -- which gfx950 vector instructions issue at the full rate?  Independent blocks of eight of one
// opcode (two-operand VOP2 forms "op d, b, d" and three-operand VOP3 forms "op d, d, b, c"), 4 resident waves per SIMD,
// and K3l's cell with its maximum taken by float instructions (the packed cell is a positive normal float).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate4 valu_rate4.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define OPERANDS : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c)
#define CELLOPS : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(b), "+v"(c), "+v"(t), "+v"(u)

#define OPS2(F) \
  F(0, "v_add_u32_e32") F(1, "v_max_u32_e32") F(2, "v_max_f32_e32") F(3, "v_min_f32_e32") F(4, "v_add_f32_e32") F(5, "v_mul_f32_e32") \
  F(6, "v_max_i32_e32") F(7, "v_sub_u32_e32") F(8, "v_mul_u32_u24_e32") F(9, "v_max_u16_e32") F(10, "v_max_f16_e32") F(11, "v_and_b32_e32") \
  F(12, "v_lshlrev_b32_e32") F(13, "v_pk_fmac_f16_e32") F(14, "v_fmac_f32_e32") F(15, "v_xnor_b32_e32")
#define OPS3(F) \
  F(32, "v_max3_u32") F(33, "v_max3_f32") F(34, "v_fma_f32") F(35, "v_lshl_add_u32") F(36, "v_lshl_or_b32") F(37, "v_add_lshl_u32") \
  F(38, "v_xad_u32") F(39, "v_perm_b32") F(40, "v_mad_u32_u24") F(41, "v_med3_f32") \
  F(46, "v_max3_f16") F(47, "v_bfe_u32") F(48, "v_or3_b32")

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int iters) {
  uint32_t a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (seed * (threadIdx.x + 1 + i * 977) & 0x3FFFFFFFu) | 0x01000000u;
  uint32_t b = (seed ^ 0x1e3779b9u) & 0x3FFFFFFFu, c = ((seed * 31u + threadIdx.x) & 0x3FFFFFFFu) | 0x01000000u, t = seed + threadIdx.x, u = seed * 7u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#define F(n, op) if (OP == n) { asm volatile(op " %0, %8, %0\n" op " %1, %8, %1\n" op " %2, %8, %2\n" op " %3, %8, %3\n" op " %4, %8, %4\n" op " %5, %8, %5\n" op " %6, %8, %6\n" op " %7, %8, %7\n" OPERANDS); }
      OPS2(F)
#undef F
#define F(n, op) if (OP == n) { asm volatile(op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9\n" OPERANDS); }
      OPS3(F)
#undef F
      if (OP == 50) { asm volatile("v_pk_max_u16 %0, %0, %8\n v_pk_max_u16 %1, %1, %8\n v_pk_max_u16 %2, %2, %8\n v_pk_max_u16 %3, %3, %8\n v_pk_max_u16 %4, %4, %8\n v_pk_max_u16 %5, %5, %8\n v_pk_max_u16 %6, %6, %8\n v_pk_max_u16 %7, %7, %8\n" OPERANDS); }
      if (OP == 51) { asm volatile("v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n v_pk_add_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n v_pk_add_u16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8\n" OPERANDS); }
      if (OP == 52) { asm volatile("v_pk_max_f16 %0, %0, %8\n v_pk_max_f16 %1, %1, %8\n v_pk_max_f16 %2, %2, %8\n v_pk_max_f16 %3, %3, %8\n v_pk_max_f16 %4, %4, %8\n v_pk_max_f16 %5, %5, %8\n v_pk_max_f16 %6, %6, %8\n v_pk_max_f16 %7, %7, %8\n" OPERANDS); }
      // K3l's cell, 8 rows: address add, diag add, horizontal add, maximum, bitop3
      if (OP == 60) {
#define X(i) "v_add_u32_e32 %10, %8, %" #i "\n v_add_u32_e32 %11, %9, %10\n v_add_u32_e32 %10, %" #i ", %9\n v_max3_u32 %11, %11, %8, %10\n v_bitop3_b32 %" #i ", %11, %8, %9 bitop3:0xea\n"
        asm volatile(R8(X) CELLOPS);
#undef X
      }
      if (OP == 61) {
#define X(i) "v_add_u32_e32 %10, %8, %" #i "\n v_add_u32_e32 %11, %9, %10\n v_add_u32_e32 %10, %" #i ", %9\n v_max_f32_e32 %11, %11, %8\n v_max_f32_e32 %11, %11, %10\n v_bitop3_b32 %" #i ", %11, %8, %9 bitop3:0xea\n"
        asm volatile(R8(X) CELLOPS);
#undef X
      }
      if (OP == 62) {
#define X(i) "v_add_u32_e32 %10, %8, %" #i "\n v_add_u32_e32 %11, %9, %10\n v_add_u32_e32 %10, %" #i ", %9\n v_max3_f32 %11, %11, %8, %10\n v_bitop3_b32 %" #i ", %11, %8, %9 bitop3:0xea\n"
        asm volatile(R8(X) CELLOPS);
#undef X
      }
      if (OP == 63) {  // four of the five without any maximum
#define X(i) "v_add_u32_e32 %10, %8, %" #i "\n v_add_u32_e32 %11, %9, %10\n v_add_u32_e32 %10, %" #i ", %9\n v_bitop3_b32 %" #i ", %11, %8, %10 bitop3:0xea\n"
        asm volatile(R8(X) CELLOPS);
#undef X
      }
    }
    b = (b + 0x00010101u) & 0x3FFFFFFFu; c = ((c ^ b) & 0x3FFFFFFFu) | 0x01000000u;
  }
  uint32_t s = t ^ u;
#pragma unroll
  for (int i = 0; i < 8; ++i) s ^= a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, uint32_t* d_out, double per_block) {
  const int blocks = 256 * 4, iters = 10000;
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
  printf("%-34s %8.3f ms  %.3f wave-instr per SIMD per ns\n", name, ms, wave_instr / ms / 1e6 / 1024);
}

int main() {
  uint32_t* d_out;
  (void)hipMalloc(&d_out, 256 * 8 * 256 * 4);
#define F(n, op) run<n>(op, d_out, 64);
  OPS2(F)
  OPS3(F)
#undef F
  run<50>("v_pk_max_u16 (2 src)", d_out, 64);
  run<51>("v_pk_add_u16 (2 src)", d_out, 64);
  run<52>("v_pk_max_f16 (2 src)", d_out, 64);
  run<60>("cell: 3 add, max3_u32, bitop3", d_out, 64 * 5);
  run<61>("cell: 3 add, 2 max_f32, bitop3", d_out, 64 * 6);
  run<62>("cell: 3 add, max3_f32, bitop3", d_out, 64 * 5);
  run<63>("cell without a maximum", d_out, 64 * 4);
  return 0;
}
