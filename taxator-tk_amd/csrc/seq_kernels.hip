// taxator-tk_amd/csrc/seq_kernels.hip -- residue packing of the resident sequence stores (gfx950).
//
// Replaces, on the device, the byte->ordinal conversion of the reference's sequence readers
//   core/src/faidx.h:315-350 (readRegion: chars -> alphabet ordinals through the SeqAn tables,
//   seqan/basic/alphabet_residue_tabs.h:107-140, 397-430).
// The sub-sequence fetch itself (core/src/sequencestorage.hh:341-369, taxonpredictionmodelsequence.hh:856-880) needs no
// kernel: a segment is a descriptor and the alignment kernels read it in place (SegView in rpa_device.h).
// HBM-streaming work: 8 bytes in, 4 bytes out per thread; the 256-entry table sits in LDS.
#include "rpa_device.h"
#include "rpa_tables.inc"

namespace rpa {

__device__ __constant__ uint8_t c_char_to_dna5[256];
__device__ __constant__ uint8_t c_char_to_aa27[256];
static bool g_tables_uploaded[64] = {false};

static hipError_t upload_tables() {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 64 && g_tables_uploaded[dev]) return hipSuccess;
  e = hipMemcpyToSymbol(HIP_SYMBOL(c_char_to_dna5), RPA_CHAR_TO_DNA5, 256);
  if (e != hipSuccess) return e;
  e = hipMemcpyToSymbol(HIP_SYMBOL(c_char_to_aa27), RPA_CHAR_TO_AA27, 256);
  if (e != hipSuccess) return e;
  if (dev < 64) g_tables_uploaded[dev] = true;
  return hipSuccess;
}

// One thread per output word: 8 letters -> 8 nibbles (DNA) or 4 letters -> 4 bytes (protein).
// The 256-entry table is staged in LDS (a constant-memory lookup with a per-lane index serialises).
template <int BITS>
__global__ __launch_bounds__(kBlock) void k_pack_letters(const uint8_t* __restrict__ letters, uint64_t n_residues,
                                                          uint32_t* __restrict__ words, uint64_t n_words) {
  __shared__ uint8_t tab[256];
  tab[threadIdx.x] = BITS == 4 ? c_char_to_dna5[threadIdx.x] : c_char_to_aa27[threadIdx.x];
  __syncthreads();
  constexpr int PER = 32 / BITS;
  for (uint64_t wi = (uint64_t)blockIdx.x * kBlock + threadIdx.x; wi < n_words; wi += (uint64_t)gridDim.x * kBlock) {
    const uint64_t r0 = wi * PER;
    uint32_t out = 0;
    if (r0 + PER <= n_residues) {
      if (PER == 8) {
        const uint2 v = *reinterpret_cast<const uint2*>(letters + r0);  // r0 % 8 == 0 and the base is 8-byte aligned
#pragma unroll
        for (int k = 0; k < 4; ++k) out |= (uint32_t)tab[(v.x >> (8 * k)) & 0xFF] << (BITS * k);
#pragma unroll
        for (int k = 0; k < 4; ++k) out |= (uint32_t)tab[(v.y >> (8 * k)) & 0xFF] << (BITS * (k + 4));
      } else {
        const uint32_t v = *reinterpret_cast<const uint32_t*>(letters + r0);
#pragma unroll
        for (int k = 0; k < 4; ++k) out |= (uint32_t)tab[(v >> (8 * k)) & 0xFF] << (BITS * k);
      }
    } else {
      for (int k = 0; k < PER; ++k)
        if (r0 + k < n_residues) out |= (uint32_t)tab[letters[r0 + k]] << (BITS * k);
    }
    words[wi] = out;
  }
}

hipError_t launch_pack_letters(hipStream_t st, int alphabet, const uint8_t* letters, uint64_t n_residues, uint32_t* words) {
  hipError_t e = upload_tables();
  if (e != hipSuccess) return e;
  if (n_residues == 0) return hipSuccess;
  const int bits = alphabet == 0 ? 4 : 8;
  const uint64_t n_words = (n_residues * bits + 31) / 32;
  uint64_t blocks = (n_words + kBlock - 1) / kBlock;
  if (blocks > 256 * 64) blocks = 256 * 64;  // grid-stride beyond ~64 blocks per CU
  if (bits == 4)
    hipLaunchKernelGGL(k_pack_letters<4>, dim3((unsigned)blocks), dim3(kBlock), 0, st, letters, n_residues, words, n_words);
  else
    hipLaunchKernelGGL(k_pack_letters<8>, dim3((unsigned)blocks), dim3(kBlock), 0, st, letters, n_residues, words, n_words);
  return hipGetLastError();
}

__global__ __launch_bounds__(kBlock) void k_store_extents(const uint64_t* __restrict__ offsets, const uint32_t* __restrict__ index, uint32_t n,
                                                          uint64_t base, ulonglong2* __restrict__ extent) {
  const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= n) return;
  const uint64_t lo = offsets[k], hi = offsets[k + 1];
  extent[index ? index[k] : k] = make_ulonglong2(base + lo, hi - lo);
}

hipError_t launch_store_extents(hipStream_t st, const uint64_t* offsets, const uint32_t* index, uint32_t n, uint64_t base, ulonglong2* extent) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_store_extents, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, offsets, index, n, base, extent);
  return hipGetLastError();
}

__global__ __launch_bounds__(kBlock) void k_store_lengths(const uint64_t* __restrict__ lengths, uint32_t n, ulonglong2* __restrict__ extent,
                                                          uint32_t* __restrict__ win_count) {
  const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= n) return;
  extent[k] = make_ulonglong2(0ull, lengths[k]);
  win_count[k] = 0u;
}
hipError_t launch_store_lengths(hipStream_t st, const uint64_t* lengths, uint32_t n, ulonglong2* extent, uint32_t* win_count) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_store_lengths, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, lengths, n, extent, win_count);
  return hipGetLastError();
}
__global__ __launch_bounds__(kBlock) void k_store_window_lists(const uint32_t* __restrict__ index, const uint32_t* __restrict__ first,
                                                               const uint32_t* __restrict__ count, uint32_t n, ulonglong2* __restrict__ extent,
                                                               uint32_t* __restrict__ win_count) {
  const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= n) return;
  const uint32_t i = index[k];
  // first entry | entries << 40 in ONE 64-bit store: a job that is running while a later block's windows are appended reads the
  // sequence's old list or its new one, never the first entry of one with the length of the other (kWinFirstBits, rpa_device.h)
  extent[i].x = (unsigned long long)first[k] | ((unsigned long long)count[k] << kWinFirstBits);
  win_count[i] = count[k];
}
hipError_t launch_store_window_lists(hipStream_t st, const uint32_t* index, const uint32_t* first, const uint32_t* count, uint32_t n,
                                     ulonglong2* extent, uint32_t* win_count) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_store_window_lists, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, index, first, count, n, extent, win_count);
  return hipGetLastError();
}

// A few words from device memory into host-mapped (pinned) memory by a kernel instead of a copy command: the per-round
// counter read-back of the placement loop then never meets another job's 80 MB host->device transfers on a copy queue.
// (Measured: no change by itself -- what stalled jobs in flight were pooled hardware queues, see make_stream in
// rpa_engine.hip -- and polling a sequence number written behind the words instead of hipStreamSynchronize gains
// nothing either, 11.9 ms per step both ways: the runtime's wait already spins.)
__global__ void k_publish_words(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst_host, uint32_t n) {
  if (threadIdx.x < n) dst_host[threadIdx.x] = src[threadIdx.x];
}
hipError_t launch_publish_words(hipStream_t st, const uint32_t* src, uint32_t* dst_host_mapped, uint32_t n) {
  if (n == 0 || n > 64) return n ? hipErrorInvalidValue : hipSuccess;
  hipLaunchKernelGGL(k_publish_words, dim3(1), dim3(64), 0, st, src, dst_host_mapped, n);
  return hipGetLastError();
}

}  // namespace rpa
