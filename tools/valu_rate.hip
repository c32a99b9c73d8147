// tools/valu_rate.hip -- microbenchmark: issue rate of the integer VALU ops the Myers kernel is made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int iters) {
  uint32_t a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1 + i * 977);
  uint32_t b = seed ^ 0x9e3779b9u, c = seed * 31u + threadIdx.x;
  uint32_t carry = 0;
  uint32_t cy[8] = {0, 1, 0, 1, 0, 1, 0, 1};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) a[i] = (a[i] & b) + 0;                                   // placeholder, replaced below
        if (OP == 1) a[i] = a[i] & (b ^ (uint32_t)r);                         // v_and (VOP2)
        if (OP == 2) a[i] = __builtin_amdgcn_bitop3_b32(a[i], b, c, 0xBE);    // v_bitop3
        if (OP == 3) a[i] = __builtin_amdgcn_alignbit(a[i], c, 31);           // v_alignbit
        if (OP == 4) { uint32_t co; a[i] = __builtin_addc(a[i], b, carry, &co); carry = co; }  // v_addc chain
        if (OP == 5) a[i] = a[i] | (c + (uint32_t)r);                         // v_or after v_add (2 VOP2)
        if (OP == 6) { uint32_t co; a[i] = __builtin_addc(a[i], b, cy[i], &co); cy[i] = co; }  // 8 independent carry chains
        if (OP == 7) a[i] = (a[i] << 1) | b;                                  // v_lshl_or_b32 (VOP3)
        if (OP == 8) a[i] = (a[i] >> (b & 31)) & 1u;                          // v_bfe_u32
        if (OP == 9) a[i] = ~(a[i] ^ (b + (uint32_t)r));                       // v_xnor (VOP2)
        if (OP == 10) a[i] = (a[i] & b) | c;                                  // v_and_or_b32 (VOP3)
        if (OP == 11) a[i] = a[i] + b + c;                                    // v_add3_u32 (VOP3)
      }
    }
    b += 0x01010101u; c ^= b;
  }
  uint32_t s = carry;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += cy[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s ^= a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
double run(const char* name, uint32_t* d_out, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, 16);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double wave_instr = (double)blocks * 4 * iters * 64.0;
  printf("%-10s %8.3f ms  %7.2f G wave-instr/s  (%.2f per SIMD per ns; 1024 SIMDs)\n", name, ms, wave_instr / ms / 1e6, wave_instr / ms / 1e6 / 1024);
  return ms;
}

int main() {
  uint32_t* d_out;
  int blocks = 256 * 8;  // 8 blocks per CU = 32 waves per CU
  hipMalloc(&d_out, blocks * 256 * 4);
  for (int rep = 0; rep < 1; ++rep) {
    run<1>("v_and", d_out, blocks, 20000);
    run<5>("v_or", d_out, blocks, 20000);
    run<2>("v_bitop3", d_out, blocks, 20000);
    run<3>("v_alignbit", d_out, blocks, 20000);
    run<4>("v_addc", d_out, blocks, 20000);
    run<5>("add+or", d_out, blocks, 20000);
    run<6>("addc_ind", d_out, blocks, 20000);
    run<7>("lshl_or", d_out, blocks, 20000);
    run<8>("bfe", d_out, blocks, 20000);
    run<9>("xnor", d_out, blocks, 20000);
    run<10>("and_or", d_out, blocks, 20000);
    run<11>("add3", d_out, blocks, 20000);
  }
  return 0;
}
