// tools/k3l_cell.hip -- the column loop of K3l (k_nw_protein_lanes<24, G>, align_protein_kernels.hip) in isolation: what
// does a strip column of 24 packed cells cost per wavefront as a function of how its 24 table reads are scheduled and of
// the resident waves per SIMD?  Same cell (address add, ds_read_b32 from the bank-replicated table, two adds, v_max3_u32,
// v_bitop3_b32), same DPP hand-over per column, synthetic residues.
//   MODE 0  the kernel's source form (the compiler clusters the reads in groups of eight)
//   MODE 1  no table: the "read" is an add (the VALU-only rate of the cell)
//   MODE 2  all 24 reads of a column issued before its first cell (sched_barrier between the two loops)
//   MODE 3  the reads of column c+1 issued before the cells of column c (software pipeline, 24 more registers)
//   MODE 4  reads in groups of 12
//   MODE 5  MODE 3 with the reads of the second half of column c+1 issued in the middle of column c (12 more registers)
//   MODE 6  MODE 0 with a table of 16-bit entries, [b][a][lane] (128 bytes per entry as before: two lanes per bank word), loaded into
//           the HIGH half of a register whose low half holds the constant part of the addend (ds_read_u16_d16_hi): half the LDS bytes,
//           no instruction more -- possible for the score-and-length cell, whose addend has nothing data-dependent below bit 16
// Build: hipcc --offload-arch=gfx950 -O3 -o k3l_cell k3l_cell.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int R = 24;
constexpr int kAX = 28;
constexpr uint32_t kPkScore = 21, kPkMove = 20, kPkMatch = 9;

__device__ __forceinline__ uint32_t from_lane_below(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false);
}

template <int MODE, int BLOCK, int SEL = 0>
__global__ __launch_bounds__(BLOCK) void k(uint32_t* out, uint32_t seed, uint32_t n_steps) {
  __shared__ uint32_t tab[kAX * kAX * 32];
  for (int idx = threadIdx.x; idx < kAX * kAX * 32; idx += BLOCK) {
    const int b = idx / (kAX * 32), a = (idx / 32) % kAX;
    tab[idx] = ((uint32_t)(((a * 7 + b * 3) % 16) + 1) << kPkScore) + (1u << kPkMove) + (a == b ? 1u << kPkMatch : 0u) + 1u;
    if (MODE == 6) { const uint32_t hi = tab[idx] >> 16; tab[idx] = hi | (hi << 16); }  // two identical 16-bit entries per word
  }
  __syncthreads();
  const uint32_t lane2 = (threadIdx.x & 63u) * 2u;
  const uint8_t* __restrict__ tabb = reinterpret_cast<const uint8_t*>(tab);
  uint32_t c_horiz = 0u - (1u << kPkScore) - (1u << kPkMove), c_keep = ~(1u << kPkMove), c_one = 1u << kPkMove;
  asm volatile("" : "+v"(c_horiz), "+v"(c_keep), "+v"(c_one));
  const uint32_t bank4 = (threadIdx.x & 31u) * 4u;
  uint32_t h = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x * 977u;
  uint32_t rowoff[R], L[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    h = h * 1664525u + 1013904223u;
    rowoff[r] = ((h >> 24) % 27u) * (kAX * 128u) + (MODE == 6 ? lane2 : bank4);
    L[r] = (300u << kPkScore) | c_one;
  }
  uint32_t dtop = L[0], bottom = L[0], res = 0;
  const bool first = (threadIdx.x % 3u) == 0u;
  uint32_t fmask = first ? ~0u : 0u, row0 = (1000u << kPkScore) | c_one;
  asm volatile("" : "+v"(fmask));
  auto cells = [&](const uint32_t* tv, int r0, int r1, uint32_t& up, uint32_t& dg) {
#pragma unroll
    for (int r = r0; r < r1; ++r) {
      const uint32_t left = L[r];
      const uint32_t raw = max(max(dg + tv[r], up), left + c_horiz);
      const uint32_t cell = __builtin_amdgcn_bitop3_b32(raw, c_keep, c_one, 0xEA);
      dg = left; L[r] = cell; up = cell;
    }
  };
  auto reads = [&](uint32_t* tv, int r0, int r1, uint32_t colterm) {
#pragma unroll
    for (int r = r0; r < r1; ++r) {
      if (MODE == 1) tv[r] = rowoff[r] + colterm;
      else tv[r] = *reinterpret_cast<const uint32_t*>(tabb + rowoff[r] + colterm);
    }
  };
  uint32_t ad = h;
  uint32_t tvn[R];
  if (MODE == 3 || MODE == 5) reads(tvn, 0, R, ((ad >> 3) % 27u) * 128u);
  for (uint32_t u = 0; u < n_steps; ++u) {
    ad = ad * 1664525u + 1013904223u;
    const uint32_t a = (ad >> 24) % 27u;  // (the kernel: a byte of a prefetched dword + a select)
    const uint32_t colterm = a * 128u;
    const uint32_t above = from_lane_below(bottom);
    uint32_t top;
    if (SEL == 0) top = first ? ((1000u - u) << kPkScore) | c_one : above;
    else if (SEL == 1) top = __builtin_amdgcn_bitop3_b32(above, row0, fmask, 0xCA);  // fmask ? row0 : above, bit by bit
    else top = above + row0;
    uint32_t up = top, dg = dtop;
    dtop = top;
    if (MODE == 6) {  // (inline asm: the compiler selects ds_read_u16 + a v_perm for this pattern, not the d16_hi load; the wait is then ours to place)
      uint32_t tv[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        tv[r] = 1u;
        asm volatile("ds_read_u16_d16_hi %0, %1" : "+v"(tv[r]) : "v"(rowoff[r] + colterm));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int r = 0; r < R; ++r) asm volatile("" : "+v"(tv[r]));
      cells(tv, 0, R, up, dg);
    } else if (MODE == 0 || MODE == 1) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t tv = MODE == 1 ? rowoff[r] + colterm : *reinterpret_cast<const uint32_t*>(tabb + rowoff[r] + colterm);
        const uint32_t left = L[r];
        const uint32_t raw = max(max(dg + tv, up), left + c_horiz);
        const uint32_t cell = __builtin_amdgcn_bitop3_b32(raw, c_keep, c_one, 0xEA);
        dg = left; L[r] = cell; up = cell;
      }
    } else if (MODE == 2) {
      uint32_t tv[R];
      reads(tv, 0, R, colterm);
      __builtin_amdgcn_sched_barrier(0);
      cells(tv, 0, R, up, dg);
      __builtin_amdgcn_sched_barrier(0);
    } else if (MODE == 4) {
      uint32_t tv[R];
      reads(tv, 0, 12, colterm);
      __builtin_amdgcn_sched_barrier(0);
      cells(tv, 0, 12, up, dg);
      __builtin_amdgcn_sched_barrier(0);
      reads(tv, 12, R, colterm);
      __builtin_amdgcn_sched_barrier(0);
      cells(tv, 12, R, up, dg);
      __builtin_amdgcn_sched_barrier(0);
    } else if (MODE == 3) {
      uint32_t tv[R];
#pragma unroll
      for (int r = 0; r < R; ++r) tv[r] = tvn[r];
      reads(tvn, 0, R, colterm);  // (this step's residue stands in for the next column's: same work)
      __builtin_amdgcn_sched_barrier(0);
      cells(tv, 0, R, up, dg);
      __builtin_amdgcn_sched_barrier(0);
    } else if (MODE == 5) {
      // tvn holds this column's 24 values; the first half is free after the first 12 cells
      __builtin_amdgcn_sched_barrier(0);
      cells(tvn, 0, 12, up, dg);
      __builtin_amdgcn_sched_barrier(0);
      reads(tvn, 0, 12, colterm);
      __builtin_amdgcn_sched_barrier(0);
      cells(tvn, 12, R, up, dg);
      __builtin_amdgcn_sched_barrier(0);
      reads(tvn, 12, R, colterm);
      __builtin_amdgcn_sched_barrier(0);
    }
    bottom = up;
    if (SEL == 0) res = (u & 63u) == 17u ? bottom : res;
    else if (SEL == 1) { const uint32_t vm = (uint32_t)((int32_t)((u & 63u) - 18u) >> 31) & ~(uint32_t)((int32_t)((u & 63u) - 17u) >> 31); res = __builtin_amdgcn_bitop3_b32(res, bottom, vm, 0xCA); }
    else res ^= bottom;
    row0 -= 1u << kPkScore;
  }
  uint32_t acc = res ^ bottom;
#pragma unroll
  for (int r = 0; r < R; ++r) acc ^= L[r];
  out[blockIdx.x * BLOCK + threadIdx.x] = acc;
}

template <int MODE, int BLOCK, int SEL = 0>
void run(const char* name, uint32_t* d_out, uint32_t n_steps) {
  const int blocks = 256;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, BLOCK, SEL>), dim3(blocks), dim3(BLOCK), 0, 0, d_out, 12345u, 64u);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, BLOCK, SEL>), dim3(blocks), dim3(BLOCK), 0, 0, d_out, 12345u, n_steps);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k<MODE, BLOCK, SEL>));
  const double cell_waves_per_simd = (double)(BLOCK / 64) / 4.0 * n_steps * R;  // per SIMD (256 CUs x 4 SIMDs, one block per CU)
  printf("%-44s %2d waves/SIMD %3d VGPRs %4zu B scratch: %8.3f ms  %6.2f ns per cell-wave per SIMD  (%.2f T cells/s)\n", name, BLOCK / 256,
         fa.numRegs, (size_t)fa.localSizeBytes, ms, ms * 1e6 / cell_waves_per_simd, (double)blocks * BLOCK * n_steps * R / ms / 1e9);
}

int main() {
  uint32_t* d_out;
  (void)hipMalloc(&d_out, 256 * 1024 * 4);
  const uint32_t n = 20000;
  run<0, 1024>("0 as in the kernel", d_out, n);
  run<0, 512>("0 as in the kernel", d_out, n);
  run<0, 256>("0 as in the kernel", d_out, n);
  run<1, 1024>("1 no table (VALU only)", d_out, n);
  run<1, 512>("1 no table (VALU only)", d_out, n);
  run<1, 256>("1 no table (VALU only)", d_out, n);
  run<2, 1024>("2 all 24 reads first", d_out, n);
  run<2, 512>("2 all 24 reads first", d_out, n);
  run<4, 1024>("4 reads in groups of 12", d_out, n);
  run<3, 1024>("3 reads one column ahead", d_out, n);
  run<3, 768>("3 reads one column ahead", d_out, n);
  run<3, 512>("3 reads one column ahead", d_out, n);
  run<5, 1024>("5 reads half a column ahead", d_out, n);
  run<5, 768>("5 reads half a column ahead", d_out, n);
  run<6, 1024>("6 16-bit table, d16_hi loads", d_out, n);
  run<6, 512>("6 16-bit table, d16_hi loads", d_out, n);
  run<0, 1024, 1>("0 + the two selects as bitop3 masks", d_out, n);
  run<0, 1024, 2>("0 + no selects (add / xor instead)", d_out, n);
  run<1, 1024, 2>("1 no table, no selects", d_out, n);
  return 0;
}
