// tools/valu_rate2.hip -- does the half rate of v_bitop3 / v_alignbit / v_addc_co on gfx950 come from the 64-bit (VOP3)
// encoding, from the operand count, or from the opcode?  Inline-asm blocks of 8 independent instructions.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate2 valu_rate2.hip ; run on the GPU box.
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
  uint32_t b = seed ^ 0x9e3779b9u, c = seed * 31u + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (OP == 0) {
#define X(i) "v_and_b32_e32 %" #i ", %8, %" #i "\n"
        asm volatile(R8(X) OPERANDS);
#undef X
      } else if (OP == 1) {
#define X(i) "v_and_b32_e64 %" #i ", %8, %" #i "\n"
        asm volatile(R8(X) OPERANDS);
#undef X
      } else if (OP == 2) {
#define X(i) "v_addc_co_u32_e32 %" #i ", vcc, %8, %" #i ", vcc\n"
        asm volatile(R8(X) OPERANDS : "vcc");
#undef X
      } else if (OP == 3) {
#define X(i) "v_addc_co_u32_e64 %" #i ", s[10:11], %8, %" #i ", s[10:11]\n"
        asm volatile(R8(X) OPERANDS : "s10", "s11");
#undef X
      } else if (OP == 4) {
#define X(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0xbe\n"
        asm volatile(R8(X) OPERANDS);
#undef X
      } else if (OP == 5) {
#define X(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 31\n"
        asm volatile(R8(X) OPERANDS);
#undef X
      } else if (OP == 6) {
#define X(i) "v_lshlrev_b32_e32 %" #i ", 1, %" #i "\n"
        asm volatile(R8(X) OPERANDS);
#undef X
      } else if (OP == 7) {  // alternate a half-rate and a full-rate op: additive (3 slots per pair) or overlapped?
#define X(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0xbe\n v_and_b32_e32 %" #i ", %8, %" #i "\n"
        asm volatile(R8(X) OPERANDS);
#undef X
      } else if (OP == 8) {
#define X(i) "v_add_u32_e32 %" #i ", %8, %" #i "\n"
        asm volatile(R8(X) OPERANDS);
#undef X
      } else if (OP == 9) {
#define X(i) "v_add_co_u32_e32 %" #i ", vcc, %8, %" #i "\n"
        asm volatile(R8(X) OPERANDS : "vcc");
#undef X
      } else if (OP == 10) {
#define X(i) "v_xor_b32_e32 %" #i ", %8, %" #i "\n"
        asm volatile(R8(X) OPERANDS);
#undef X
      } else if (OP == 11) {  // bitop3 with two VGPR sources + inline constant
#define X(i) "v_bitop3_b32 %" #i ", %" #i ", %8, -1 bitop3:0xbe\n"
        asm volatile(R8(X) OPERANDS);
#undef X
      } else if (OP == 12) {
#define X(i) "v_not_b32_e32 %" #i ", %" #i "\n"
        asm volatile(R8(X) OPERANDS);
#undef X
      } else if (OP == 13) {  // SDWA-free VOP2 with an SGPR source
#define X(i) "v_or_b32_e32 %" #i ", s12, %" #i "\n"
        asm volatile(R8(X) OPERANDS : "s12");
#undef X
      } else if (OP == 14) {
#define X(i) "v_bfi_b32 %" #i ", %" #i ", %8, %9\n"
        asm volatile(R8(X) OPERANDS);
#undef X
      } else if (OP == 15) {
#define X(i) "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
        asm volatile(R8(X) OPERANDS);
#undef X
      }
    }
    b += 0x01010101u; c ^= b;
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s ^= a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, uint32_t* d_out, int blocks, int iters, double per_block = 64.0) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, 16);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double wave_instr = (double)blocks * 4 * iters * per_block;
  printf("%-28s %8.3f ms  %.3f wave-instr per SIMD per ns\n", name, ms, wave_instr / ms / 1e6 / 1024);
}

int main() {
  uint32_t* d_out;
  int blocks = 256 * 8;
  hipMalloc(&d_out, blocks * 256 * 4);
  run<0>("v_and_b32_e32", d_out, blocks, 20000);
  run<1>("v_and_b32_e64", d_out, blocks, 20000);
  run<10>("v_xor_b32_e32", d_out, blocks, 20000);
  run<8>("v_add_u32_e32", d_out, blocks, 20000);
  run<9>("v_add_co_u32_e32 (vcc)", d_out, blocks, 20000);
  run<2>("v_addc_co_u32_e32 (vcc)", d_out, blocks, 20000);
  run<3>("v_addc_co_u32_e64 (sgpr)", d_out, blocks, 20000);
  run<4>("v_bitop3_b32 3 vgpr", d_out, blocks, 20000);
  run<11>("v_bitop3_b32 2 vgpr + const", d_out, blocks, 20000);
  run<14>("v_bfi_b32", d_out, blocks, 20000);
  run<5>("v_alignbit_b32 (.., 31)", d_out, blocks, 20000);
  run<6>("v_lshlrev_b32_e32", d_out, blocks, 20000);
  run<12>("v_not_b32_e32", d_out, blocks, 20000);
  run<13>("v_or_b32_e32 sgpr src", d_out, blocks, 20000);
  run<15>("v_mov_b32_dpp row_shr:1", d_out, blocks, 20000);
  run<7>("bitop3 + and (pairs)", d_out, blocks, 20000, 128.0);
  return 0;
}
