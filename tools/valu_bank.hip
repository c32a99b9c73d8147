// tools/valu_bank.hip -- VGPR bank conflicts on gfx950: issue rate of v_bitop3_b32 / v_and_b32 / v_alignbit_b32 when the
// source registers sit in distinct banks (register index mod 4) or share one.  Physical registers are named explicitly.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_bank valu_bank.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55"
// I(d, a, b): one instruction with dst = src0 = v<d>, src1 = v<a>, src2 = v<b>
#define B3(d,a,b) "v_bitop3_b32 v" #d ", v" #d ", v" #a ", v" #b " bitop3:0xbe\n"
#define AND(d,a,b) "v_and_b32_e32 v" #d ", v" #a ", v" #d "\n"
#define ALB(d,a,b) "v_alignbit_b32 v" #d ", v" #d ", v" #a ", 31\n"
#define BFI(d,a,b) "v_bfi_b32 v" #d ", v" #d ", v" #a ", v" #b "\n"
#define ADD3(d,a,b) "v_add3_u32 v" #d ", v" #d ", v" #a ", v" #b "\n"
// 8 instructions, dst v40..v47
#define DISTINCT(I) I(40,49,50) I(41,50,51) I(42,51,48) I(43,48,49) I(44,53,54) I(45,54,55) I(46,55,52) I(47,52,53)
#define S12SAME(I)  I(40,49,53) I(41,50,54) I(42,51,55) I(43,48,52) I(44,53,49) I(45,54,50) I(46,55,51) I(47,52,48)
#define S01SAME(I)  I(40,48,49) I(41,49,50) I(42,50,51) I(43,51,48) I(44,52,53) I(45,53,54) I(46,54,55) I(47,55,52)
#define ALLSAME(I)  I(40,48,52) I(41,49,53) I(42,50,54) I(43,51,55) I(44,52,48) I(45,53,49) I(46,54,50) I(47,55,51)

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int iters) {
  uint32_t acc = seed + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (OP == 0) asm volatile(DISTINCT(B3) ::: CLOB);
      if (OP == 1) asm volatile(S12SAME(B3) ::: CLOB);
      if (OP == 2) asm volatile(S01SAME(B3) ::: CLOB);
      if (OP == 3) asm volatile(ALLSAME(B3) ::: CLOB);
      if (OP == 4) asm volatile(DISTINCT(AND) ::: CLOB);
      if (OP == 5) asm volatile(S01SAME(AND) ::: CLOB);
      if (OP == 6) asm volatile(DISTINCT(ALB) ::: CLOB);
      if (OP == 7) asm volatile(S01SAME(ALB) ::: CLOB);
      if (OP == 8) asm volatile(DISTINCT(BFI) ::: CLOB);
      if (OP == 9) asm volatile(DISTINCT(ADD3) ::: CLOB);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int OP>
void run(const char* name, uint32_t* d_out, int blocks, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, 16);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double wave_instr = (double)blocks * 4 * iters * 64.0;
  printf("%-44s %8.3f ms  %.3f wave-instr per SIMD per ns\n", name, ms, wave_instr / ms / 1e6 / 1024);
}

int main() {
  uint32_t* d_out;
  int blocks = 256 * 8;
  (void)hipMalloc(&d_out, blocks * 256 * 4);
  run<0>("v_bitop3: three sources, three banks", d_out, blocks, 20000);
  run<1>("v_bitop3: src1, src2 share a bank", d_out, blocks, 20000);
  run<2>("v_bitop3: src0, src1 share a bank", d_out, blocks, 20000);
  run<3>("v_bitop3: all three in one bank", d_out, blocks, 20000);
  run<4>("v_and_b32: two banks", d_out, blocks, 20000);
  run<5>("v_and_b32: one bank", d_out, blocks, 20000);
  run<6>("v_alignbit (v,v,31): two banks", d_out, blocks, 20000);
  run<7>("v_alignbit (v,v,31): one bank", d_out, blocks, 20000);
  run<8>("v_bfi_b32: three banks", d_out, blocks, 20000);
  run<9>("v_add3_u32: three banks", d_out, blocks, 20000);
  return 0;
}
