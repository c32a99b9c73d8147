// tools/myers_pair.hip -- the Myers column of a four-word window (k_myers_short_band4's inner step) with the two shifts by
// one (HP << 1 | carry, HN << 1 | carry) done per WORD with v_alignbit_b32 (the kernel) or per PAIR of words with one
// 64-bit shift-and-add (v_lshl_add_u64, tools/valu_rate5.hip: 2.3 ns for two words against 2 x 1.8).  Eq masks from
// registers (no LDS): the instruction mix only.
// Build: hipcc --offload-arch=gfx950 -O3 -o myers_pair myers_pair.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int W = 4;

__device__ __forceinline__ void column_words(uint32_t (&VP)[W], uint32_t (&VN)[W], const uint32_t (&E)[W]) {
  uint32_t carry = 0, hp_prev = 0x80000000u, hn_prev = 0;
#pragma unroll
  for (int wi = 0; wi < W; ++wi) {
    const uint32_t vp = VP[wi], vn = VN[wi];
    const uint32_t X = E[wi] | vn;
    uint32_t cout;
    const uint32_t sum = __builtin_addc(vp, X & vp, carry, &cout);
    carry = cout;
    const uint32_t D0 = __builtin_amdgcn_bitop3_b32(sum, vp, X, 0xBE);
    const uint32_t HN = vp & D0;
    const uint32_t HP = __builtin_amdgcn_bitop3_b32(vn, vp, D0, 0xF1);
    const uint32_t Xs = __builtin_amdgcn_alignbit(HP, hp_prev, 31);
    const uint32_t Hs = __builtin_amdgcn_alignbit(HN, hn_prev, 31);
    hp_prev = HP; hn_prev = HN;
    VN[wi] = Xs & D0;
    VP[wi] = __builtin_amdgcn_bitop3_b32(Hs, Xs, D0, 0xF1);
  }
}

__device__ __forceinline__ void column_pairs(uint32_t (&VP)[W], uint32_t (&VN)[W], const uint32_t (&E)[W]) {
  uint32_t carry = 0;
  uint32_t D0[W], HP[W], HN[W];
#pragma unroll
  for (int wi = 0; wi < W; ++wi) {
    const uint32_t vp = VP[wi], vn = VN[wi];
    const uint32_t X = E[wi] | vn;
    uint32_t cout;
    const uint32_t sum = __builtin_addc(vp, X & vp, carry, &cout);
    carry = cout;
    D0[wi] = __builtin_amdgcn_bitop3_b32(sum, vp, X, 0xBE);
    HN[wi] = vp & D0[wi];
    HP[wi] = __builtin_amdgcn_bitop3_b32(vn, vp, D0[wi], 0xF1);
  }
  uint64_t cp = 1ull, cn = 0ull;  // carry into the window's lowest word: HP 1, HN 0 (the row above the window)
#pragma unroll
  for (int k = 0; k < W; k += 2) {
    const uint64_t hp = ((uint64_t)HP[k + 1] << 32) | HP[k], hn = ((uint64_t)HN[k + 1] << 32) | HN[k];
    uint64_t xs, hs;
    asm("v_lshl_add_u64 %0, %1, 1, %2" : "=v"(xs) : "v"(hp), "v"(cp));
    asm("v_lshl_add_u64 %0, %1, 1, %2" : "=v"(hs) : "v"(hn), "v"(cn));
    cp = HP[k + 1] >> 31; cn = HN[k + 1] >> 31;
    const uint32_t xs0 = (uint32_t)xs, xs1 = (uint32_t)(xs >> 32), hs0 = (uint32_t)hs, hs1 = (uint32_t)(hs >> 32);
    VN[k] = xs0 & D0[k]; VN[k + 1] = xs1 & D0[k + 1];
    VP[k] = __builtin_amdgcn_bitop3_b32(hs0, xs0, D0[k], 0xF1);
    VP[k + 1] = __builtin_amdgcn_bitop3_b32(hs1, xs1, D0[k + 1], 0xF1);
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed, int iters) {
  uint32_t VP[W], VN[W], E[W];
#pragma unroll
  for (int i = 0; i < W; ++i) { VP[i] = ~0u; VN[i] = 0u; E[i] = seed * (threadIdx.x * 4 + i + 1) * 2654435761u; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (MODE == 0) column_words(VP, VN, E); else column_pairs(VP, VN, E);
#pragma unroll
      for (int i = 0; i < W; ++i) E[i] = (E[i] ^ VP[(i + 1) % W]) + 0x9e3779b9u;  // (two full-rate instructions per word and column stand in for the Eq read)
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < W; ++i) s ^= VP[i] ^ VN[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
float run(const char* name, uint32_t* d_out, int waves) {
  const int blocks = 256 * waves, iters = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, 16);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double word_waves_per_simd = (double)waves * iters * 8 * W;
  printf("%-34s %d waves/SIMD %8.3f ms  %5.2f ns per word per SIMD\n", name, waves, ms, ms * 1e6 / word_waves_per_simd);
  return ms;
}

int main() {
  uint32_t* d_out;
  (void)hipMalloc(&d_out, 256 * 8 * 256 * 4);
  for (int w : {4, 6}) {
    run<0>("shift per word (v_alignbit)", d_out, w);
    run<1>("shift per pair (v_lshl_add_u64)", d_out, w);
  }
  // check: same results
  uint32_t* h0 = new uint32_t[256 * 256]; uint32_t* h1 = new uint32_t[256 * 256];
  hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d_out, 777u, 50); (void)hipMemcpy(h0, d_out, 256 * 256 * 4, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d_out, 777u, 50); (void)hipMemcpy(h1, d_out, 256 * 256 * 4, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 256 * 256; ++i) bad += h0[i] != h1[i];
  printf("results differ in %d of %d lanes\n", bad, 256 * 256);
  return 0;
}
