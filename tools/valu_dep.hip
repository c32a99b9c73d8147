// tools/valu_dep.hip -- issue rate of DEPENDENT VALU chains on gfx950 as a function of resident waves per SIMD:
// every instruction reads the result of the previous one (the shape of a Myers column: one carry/shift chain per lane).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_dep valu_dep.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int iters) {
  uint32_t a = seed * (threadIdx.x + 1), b = seed ^ 0x9e3779b9u, c = seed * 31u + threadIdx.x, a2 = a ^ 0x5bd1e995u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (OP == 0) asm volatile("v_and_b32_e32 %0, %1, %0\n v_xor_b32_e32 %0, %2, %0\n v_and_b32_e32 %0, %1, %0\n v_xor_b32_e32 %0, %2, %0\n"
                                "v_and_b32_e32 %0, %1, %0\n v_xor_b32_e32 %0, %2, %0\n v_and_b32_e32 %0, %1, %0\n v_xor_b32_e32 %0, %2, %0\n" : "+v"(a) : "v"(b), "v"(c));
      if (OP == 1) asm volatile("v_alignbit_b32 %0, %0, %1, 31\n v_alignbit_b32 %0, %0, %2, 31\n v_alignbit_b32 %0, %0, %1, 31\n v_alignbit_b32 %0, %0, %2, 31\n"
                                "v_alignbit_b32 %0, %0, %1, 31\n v_alignbit_b32 %0, %0, %2, 31\n v_alignbit_b32 %0, %0, %1, 31\n v_alignbit_b32 %0, %0, %2, 31\n" : "+v"(a) : "v"(b), "v"(c));
      if (OP == 2) asm volatile("v_addc_co_u32_e32 %0, vcc, %1, %0, vcc\n v_addc_co_u32_e32 %0, vcc, %2, %0, vcc\n v_addc_co_u32_e32 %0, vcc, %1, %0, vcc\n v_addc_co_u32_e32 %0, vcc, %2, %0, vcc\n"
                                "v_addc_co_u32_e32 %0, vcc, %1, %0, vcc\n v_addc_co_u32_e32 %0, vcc, %2, %0, vcc\n v_addc_co_u32_e32 %0, vcc, %1, %0, vcc\n v_addc_co_u32_e32 %0, vcc, %2, %0, vcc\n" : "+v"(a) : "v"(b), "v"(c) : "vcc");
      // the Myers word mix, fully dependent: or, bitop3, addc, bitop3, and, bitop3, alignbit, alignbit(indep of prev), and, bitop3 (8 of 10 shown twice = 16 instr; count 10)
      if (OP == 3) asm volatile("v_or_b32_e32 %0, %1, %0\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0xc8\n v_addc_co_u32_e32 %0, vcc, %1, %0, vcc\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0xbe\n"
                                "v_and_b32_e32 %0, %1, %0\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf1\n v_alignbit_b32 %0, %0, %1, 31\n v_alignbit_b32 %0, %0, %2, 31\n"
                                "v_and_b32_e32 %0, %1, %0\n v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf1\n" : "+v"(a) : "v"(b), "v"(c) : "vcc");
      // the same mix as TWO independent chains interleaved instruction by instruction (count 20)
      if (OP == 4) asm volatile("v_or_b32_e32 %0, %2, %0\n v_or_b32_e32 %1, %3, %1\n v_bitop3_b32 %0, %0, %2, %3 bitop3:0xc8\n v_bitop3_b32 %1, %1, %3, %2 bitop3:0xc8\n"
                                "v_addc_co_u32_e32 %0, vcc, %2, %0, vcc\n v_addc_co_u32_e32 %1, vcc, %3, %1, vcc\n v_bitop3_b32 %0, %0, %2, %3 bitop3:0xbe\n v_bitop3_b32 %1, %1, %3, %2 bitop3:0xbe\n"
                                "v_and_b32_e32 %0, %2, %0\n v_and_b32_e32 %1, %3, %1\n v_bitop3_b32 %0, %0, %2, %3 bitop3:0xf1\n v_bitop3_b32 %1, %1, %3, %2 bitop3:0xf1\n"
                                "v_alignbit_b32 %0, %0, %2, 31\n v_alignbit_b32 %1, %1, %3, 31\n v_alignbit_b32 %0, %0, %3, 31\n v_alignbit_b32 %1, %1, %2, 31\n"
                                "v_and_b32_e32 %0, %2, %0\n v_and_b32_e32 %1, %3, %1\n v_bitop3_b32 %0, %0, %2, %3 bitop3:0xf1\n v_bitop3_b32 %1, %1, %3, %2 bitop3:0xf1\n" : "+v"(a), "+v"(a2) : "v"(b), "v"(c) : "vcc");
    }
    b += 0x01010101u; c ^= b;
  }
  out[blockIdx.x * 256 + threadIdx.x] = a ^ a2;
}

template <int OP>
void run(const char* name, uint32_t* d_out, int blocks_per_cu, int iters, double per_r) {
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
  double wave_instr = (double)blocks * 4 * iters * 8 * per_r;
  printf("%-34s %d waves/SIMD: %8.3f ms  %.3f wave-instr per SIMD per ns\n", name, blocks_per_cu, ms, wave_instr / ms / 1e6 / 1024);
}

int main() {
  uint32_t* d_out;
  (void)hipMalloc(&d_out, 256 * 8 * 256 * 4);
  for (int w : {1, 2, 3, 4, 5, 6, 8}) run<0>("dependent v_and/v_xor", d_out, w, 20000, 8);
  for (int w : {1, 2, 4, 8}) run<1>("dependent v_alignbit", d_out, w, 20000, 8);
  for (int w : {1, 2, 4, 8}) run<2>("dependent v_addc_co (vcc)", d_out, w, 20000, 8);
  for (int w : {1, 2, 3, 4, 5, 6, 8}) run<3>("Myers word mix, one chain", d_out, w, 20000, 10);
  for (int w : {1, 2, 3, 4, 6, 8}) run<4>("Myers word mix, two chains", d_out, w, 20000, 20);
  return 0;
}
