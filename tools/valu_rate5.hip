// tools/valu_rate5.hip -- 64-bit and carry/compare instructions on gfx950: could the Myers word pair (carry chain, shifts by
// one) run on 64-bit operations?  Independent blocks of four 64-bit (or eight 32-bit) destinations, 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate5 valu_rate5.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55"

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int iters) {
  uint32_t acc = seed + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      // 64-bit: four per block (count 4)
      if (OP == 0) asm volatile("v_lshl_add_u64 v[40:41], v[40:41], 1, v[48:49]\n v_lshl_add_u64 v[42:43], v[42:43], 1, v[50:51]\n v_lshl_add_u64 v[44:45], v[44:45], 1, v[52:53]\n v_lshl_add_u64 v[46:47], v[46:47], 1, v[54:55]\n" ::: CLOB);
      if (OP == 1) asm volatile("v_lshl_add_u64 v[40:41], v[40:41], 0, v[48:49]\n v_lshl_add_u64 v[42:43], v[42:43], 0, v[50:51]\n v_lshl_add_u64 v[44:45], v[44:45], 0, v[52:53]\n v_lshl_add_u64 v[46:47], v[46:47], 0, v[54:55]\n" ::: CLOB);
      if (OP == 2) asm volatile("v_lshlrev_b64 v[40:41], 1, v[40:41]\n v_lshlrev_b64 v[42:43], 1, v[42:43]\n v_lshlrev_b64 v[44:45], 1, v[44:45]\n v_lshlrev_b64 v[46:47], 1, v[46:47]\n" ::: CLOB);
      if (OP == 3) asm volatile("v_cmp_lt_u64 vcc, v[40:41], v[48:49]\n v_cmp_lt_u64 vcc, v[42:43], v[50:51]\n v_cmp_lt_u64 vcc, v[44:45], v[52:53]\n v_cmp_lt_u64 vcc, v[46:47], v[54:55]\n" ::: CLOB, "vcc");
      if (OP == 4) asm volatile("v_pk_mov_b32 v[40:41], v[48:49], v[50:51]\n v_pk_mov_b32 v[42:43], v[50:51], v[52:53]\n v_pk_mov_b32 v[44:45], v[52:53], v[54:55]\n v_pk_mov_b32 v[46:47], v[54:55], v[48:49]\n" ::: CLOB);
      if (OP == 5) asm volatile("v_mov_b64 v[40:41], v[48:49]\n v_mov_b64 v[42:43], v[50:51]\n v_mov_b64 v[44:45], v[52:53]\n v_mov_b64 v[46:47], v[54:55]\n" ::: CLOB);
      // 32-bit: eight per block (count 8)
      if (OP == 10) asm volatile("v_cmp_lt_u32 vcc, v40, v48\n v_cmp_lt_u32 vcc, v41, v49\n v_cmp_lt_u32 vcc, v42, v50\n v_cmp_lt_u32 vcc, v43, v51\n v_cmp_lt_u32 vcc, v44, v52\n v_cmp_lt_u32 vcc, v45, v53\n v_cmp_lt_u32 vcc, v46, v54\n v_cmp_lt_u32 vcc, v47, v55\n" ::: CLOB, "vcc");
      if (OP == 11) asm volatile("v_cndmask_b32 v40, v40, v48, vcc\n v_cndmask_b32 v41, v41, v49, vcc\n v_cndmask_b32 v42, v42, v50, vcc\n v_cndmask_b32 v43, v43, v51, vcc\n v_cndmask_b32 v44, v44, v52, vcc\n v_cndmask_b32 v45, v45, v53, vcc\n v_cndmask_b32 v46, v46, v54, vcc\n v_cndmask_b32 v47, v47, v55, vcc\n" ::: CLOB);
      if (OP == 12) asm volatile("v_mov_b32 v40, v48\n v_mov_b32 v41, v49\n v_mov_b32 v42, v50\n v_mov_b32 v43, v51\n v_mov_b32 v44, v52\n v_mov_b32 v45, v53\n v_mov_b32 v46, v54\n v_mov_b32 v47, v55\n" ::: CLOB);
      if (OP == 13) asm volatile("v_mov_b32_dpp v40, v48 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v41, v49 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v42, v50 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v43, v51 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                                 "v_mov_b32_dpp v44, v52 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v45, v53 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v46, v54 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v47, v55 wave_shr:1 row_mask:0xf bank_mask:0xf\n" ::: CLOB);
      if (OP == 14) asm volatile("v_lshlrev_b32_sdwa v40, v48, v49 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa v41, v48, v50 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                                 "v_lshlrev_b32_sdwa v42, v48, v51 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa v43, v48, v52 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                                 "v_lshlrev_b32_sdwa v44, v48, v53 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa v45, v48, v54 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                                 "v_lshlrev_b32_sdwa v46, v48, v55 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa v47, v48, v49 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n" ::: CLOB);
      if (OP == 15) asm volatile("v_add_u32_sdwa v40, v48, v49 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa v41, v48, v50 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                                 "v_add_u32_sdwa v42, v48, v51 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa v43, v48, v52 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                                 "v_add_u32_sdwa v44, v48, v53 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa v45, v48, v54 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
                                 "v_add_u32_sdwa v46, v48, v55 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa v47, v48, v49 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n" ::: CLOB);
      if (OP == 16) asm volatile("v_max_u16 v40, v40, v48\n v_max_u16 v41, v41, v49\n v_max_u16 v42, v42, v50\n v_max_u16 v43, v43, v51\n v_max_u16 v44, v44, v52\n v_max_u16 v45, v45, v53\n v_max_u16 v46, v46, v54\n v_max_u16 v47, v47, v55\n" ::: CLOB);
      if (OP == 17) asm volatile("v_add_u16 v40, v40, v48\n v_add_u16 v41, v41, v49\n v_add_u16 v42, v42, v50\n v_add_u16 v43, v43, v51\n v_add_u16 v44, v44, v52\n v_add_u16 v45, v45, v53\n v_add_u16 v46, v46, v54\n v_add_u16 v47, v47, v55\n" ::: CLOB);
      if (OP == 18) asm volatile("v_cndmask_b32_e64 v40, v40, v48, s[10:11]\n v_cndmask_b32_e64 v41, v41, v49, s[10:11]\n v_cndmask_b32_e64 v42, v42, v50, s[10:11]\n v_cndmask_b32_e64 v43, v43, v51, s[10:11]\n v_cndmask_b32_e64 v44, v44, v52, s[10:11]\n v_cndmask_b32_e64 v45, v45, v53, s[10:11]\n v_cndmask_b32_e64 v46, v46, v54, s[10:11]\n v_cndmask_b32_e64 v47, v47, v55, s[10:11]\n" ::: CLOB);
      if (OP == 19) asm volatile("v_cmp_lt_u32 vcc, v40, v48\n v_cndmask_b32 v40, v49, v50, vcc\n v_cmp_lt_u32 vcc, v41, v49\n v_cndmask_b32 v41, v50, v51, vcc\n v_cmp_lt_u32 vcc, v42, v50\n v_cndmask_b32 v42, v51, v52, vcc\n v_cmp_lt_u32 vcc, v43, v51\n v_cndmask_b32 v43, v52, v53, vcc\n" ::: CLOB, "vcc");
      if (OP == 20) asm volatile("v_cndmask_b32 v40, v49, v48, vcc\n v_cndmask_b32 v41, v50, v49, vcc\n v_cndmask_b32 v42, v51, v50, vcc\n v_cndmask_b32 v43, v52, v51, vcc\n v_cndmask_b32 v44, v53, v52, vcc\n v_cndmask_b32 v45, v54, v53, vcc\n v_cndmask_b32 v46, v55, v54, vcc\n v_cndmask_b32 v47, v48, v55, vcc\n" ::: CLOB);
      if (OP == 21) asm volatile("v_bitop3_b32 v40, v49, v48, v50 bitop3:0xca\n v_bitop3_b32 v41, v50, v49, v51 bitop3:0xca\n v_bitop3_b32 v42, v51, v50, v52 bitop3:0xca\n v_bitop3_b32 v43, v52, v51, v53 bitop3:0xca\n v_bitop3_b32 v44, v53, v52, v54 bitop3:0xca\n v_bitop3_b32 v45, v54, v53, v55 bitop3:0xca\n v_bitop3_b32 v46, v55, v54, v48 bitop3:0xca\n v_bitop3_b32 v47, v48, v55, v49 bitop3:0xca\n" ::: CLOB);
      if (OP == 22) asm volatile("v_cmp_lt_u32 vcc, v40, v48\n s_and_b64 vcc, s[10:11], vcc\n v_cndmask_b32 v40, v49, v50, vcc\n v_cmp_lt_u32 vcc, v41, v49\n s_and_b64 vcc, s[10:11], vcc\n v_cndmask_b32 v41, v50, v51, vcc\n v_cmp_lt_u32 vcc, v42, v50\n s_and_b64 vcc, s[10:11], vcc\n v_cndmask_b32 v42, v51, v52, vcc\n v_cmp_lt_u32 vcc, v43, v51\n s_and_b64 vcc, s[10:11], vcc\n v_cndmask_b32 v43, v52, v53, vcc\n" ::: CLOB, "vcc", "scc");
      if (OP == 23) asm volatile("v_add_u32 v40, 0xfff00000, v40\n v_add_u32 v41, 0xfff00000, v41\n v_add_u32 v42, 0xfff00000, v42\n v_add_u32 v43, 0xfff00000, v43\n v_add_u32 v44, 0xfff00000, v44\n v_add_u32 v45, 0xfff00000, v45\n v_add_u32 v46, 0xfff00000, v46\n v_add_u32 v47, 0xfff00000, v47\n" ::: CLOB);
      if (OP == 24) asm volatile("v_add_u32 v40, 1, v40\n v_add_u32 v41, 1, v41\n v_add_u32 v42, 1, v42\n v_add_u32 v43, 1, v43\n v_add_u32 v44, 1, v44\n v_add_u32 v45, 1, v45\n v_add_u32 v46, 1, v46\n v_add_u32 v47, 1, v47\n" ::: CLOB);
      if (OP == 25) asm volatile("v_cmp_lt_u32 s[12:13], v40, v48\n s_and_b64 s[12:13], s[10:11], s[12:13]\n v_cndmask_b32_e64 v40, v49, v50, s[12:13]\n v_cmp_lt_u32 s[12:13], v41, v49\n s_and_b64 s[12:13], s[10:11], s[12:13]\n v_cndmask_b32_e64 v41, v50, v51, s[12:13]\n v_cmp_lt_u32 s[12:13], v42, v50\n s_and_b64 s[12:13], s[10:11], s[12:13]\n v_cndmask_b32_e64 v42, v51, v52, s[12:13]\n v_cmp_lt_u32 s[12:13], v43, v51\n s_and_b64 s[12:13], s[10:11], s[12:13]\n v_cndmask_b32_e64 v43, v52, v53, s[12:13]\n" ::: CLOB, "s12", "s13", "scc");
    }
    acc += it;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
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
  setvbuf(stdout, nullptr, _IONBF, 0);
  uint32_t* d_out;
  (void)hipMalloc(&d_out, 256 * 8 * 256 * 4);
  run<0>("v_lshl_add_u64 (<<1)", d_out, 32);
  run<1>("v_lshl_add_u64 (<<0)", d_out, 32);
  run<2>("v_lshlrev_b64", d_out, 32);
  run<3>("v_cmp_lt_u64", d_out, 32);
  run<4>("v_pk_mov_b32", d_out, 32);
  run<5>("v_mov_b64", d_out, 32);
  run<10>("v_cmp_lt_u32 (vcc)", d_out, 64);
  run<11>("v_cndmask_b32 (vcc)", d_out, 64);
  run<12>("v_mov_b32", d_out, 64);
  run<13>("v_mov_b32_dpp wave_shr:1", d_out, 64);
  run<14>("v_lshlrev_b32_sdwa (byte)", d_out, 64);
  run<15>("v_add_u32_sdwa (byte)", d_out, 64);
  run<16>("v_max_u16", d_out, 64);
  run<17>("v_add_u16", d_out, 64);
  run<18>("v_cndmask_b32_e64 (sgpr pair)", d_out, 64);
  run<20>("v_cndmask_b32 (vcc), dst != src", d_out, 64);
  run<19>("v_cmp_lt_u32 + v_cndmask pairs", d_out, 64);
  run<21>("v_bitop3_b32 0xca (bit select)", d_out, 64);
  run<22>("v_cmp, s_and vcc, v_cndmask (x4: 8 VALU)", d_out, 64);
  run<25>("the same through an SGPR pair", d_out, 64);
  run<23>("v_add_u32 literal", d_out, 64);
  run<24>("v_add_u32 inline constant", d_out, 64);
  return 0;
}
