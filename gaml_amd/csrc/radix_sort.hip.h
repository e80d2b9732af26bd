// radix_sort.hip.h -- stable LSD radix sort of 64-bit keys (with an optional payload) and an inclusive running maximum,
// written for the two cold-path orderings of this library:
//   * the aligner's hits of a large batch by (window, position, read, strand, order) -- the order in which the reference
//     files alignments per window (graph.cc:841, 891, 895-897; graph.h:229-232) -- aligner_launch.hip.h
//   * the PacBio coverage sweep's intervals by (contig, begin) and its positions (graph.cc:3198-3250) -- pacbio_launch.hip.h
// Eight bits per pass, four small dispatches per pass, no library:
//   rs_histogram_kernel   a block counts the digits of its tile (kRsTile keys) -> hist[digit][block]
//   rs_scan_digit_kernel  + rs_scan_totals_kernel: exclusive prefix over hist in (digit, block) order = where each block's
//                         keys of each digit start in the output
//   rs_scatter_kernel     the block walks its tile again in chunks of 256 keys IN INPUT ORDER; a key's rank among the
//                         chunk's equal digits comes from wave ballots (eight, one per digit bit) and the waves' counts in
//                         LDS; a running base per digit carries over the chunks. Equal digits keep their input order:
//                         the sort is stable, pass after pass.
// HBM-bound integer work: a pass reads the keys twice and writes them once (+ payload once each way).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace gaml {

constexpr int kRsBlock = 256;                 // threads per block (four waves)
constexpr int kRsChunks = 16;                 // chunks of kRsBlock keys per tile
constexpr int kRsTile = kRsBlock * kRsChunks; // keys per block and pass
constexpr int kRsScanBlock = 1024;

typedef unsigned long long rs_u64;

__global__ __launch_bounds__(kRsBlock) void rs_histogram_kernel(const rs_u64* keys, unsigned n, int shift, unsigned mask, unsigned n_blocks, unsigned* hist) {
  __shared__ unsigned cnt[256];
  cnt[threadIdx.x] = 0;
  __syncthreads();
  const unsigned lo = blockIdx.x * (unsigned)kRsTile;
#pragma unroll 4
  for (int k = 0; k < kRsChunks; k++) {
    const unsigned i = lo + (unsigned)k * kRsBlock + threadIdx.x;
    if (i < n) atomicAdd(&cnt[(unsigned)(keys[i] >> shift) & mask], 1u);
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * n_blocks + blockIdx.x] = cnt[threadIdx.x];
}

// The offsets in two small steps (one block walking all 256 * n_blocks counters took 150 us a pass -- every thread a run
// of its own, a cache line per lane and load):
//   rs_scan_digit_kernel   block d: exclusive prefix of digit d's counts over the blocks, in place (coalesced chunks of
//                          256 with a carry), and the digit's total -> totals[d]
//   rs_scan_totals_kernel  one block: exclusive prefix of the 256 totals, in place
// A block's keys of digit d then start at totals[d] + hist[d][block] (rs_scatter_kernel adds the two).
__global__ __launch_bounds__(kRsBlock) void rs_scan_digit_kernel(unsigned* hist, unsigned n_blocks, unsigned* totals) {
  __shared__ unsigned wave_sum[kRsBlock / 64];
  __shared__ unsigned carry_sh;
  unsigned* row = hist + (size_t)blockIdx.x * n_blocks;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_sh = 0;
  __syncthreads();
  for (unsigned c0 = 0; c0 < n_blocks; c0 += kRsBlock) {  // (block-uniform)
    const unsigned i = c0 + threadIdx.x;
    const unsigned v = i < n_blocks ? row[i] : 0u;
    unsigned incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned u = __shfl_up(incl, d); if (lane >= d) incl += u; }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    unsigned before = carry_sh;
    for (int w = 0; w < wave; w++) before += wave_sum[w];
    if (i < n_blocks) row[i] = before + incl - v;
    __syncthreads();  // everyone has read carry_sh and wave_sum
    if (threadIdx.x == kRsBlock - 1) carry_sh = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry_sh;
}
__global__ __launch_bounds__(kRsBlock) void rs_scan_totals_kernel(unsigned* totals) {
  __shared__ unsigned wave_sum[kRsBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned v = totals[threadIdx.x];
  unsigned incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const unsigned u = __shfl_up(incl, d); if (lane >= d) incl += u; }
  if (lane == 63) wave_sum[wave] = incl;
  __syncthreads();
  unsigned before = 0;
  for (int w = 0; w < wave; w++) before += wave_sum[w];
  totals[threadIdx.x] = before + incl - v;
}

template <class V, bool HAS_V>
__global__ __launch_bounds__(kRsBlock) void rs_scatter_kernel(const rs_u64* keys, const V* vals, unsigned n, int shift, unsigned mask, unsigned n_blocks, const unsigned* offs, const unsigned* totals,
                                                             rs_u64* keys_out, V* vals_out) {
  __shared__ unsigned base[256];                   // where this block's next key of each digit goes
  __shared__ unsigned wcnt[kRsBlock / 64][256];    // the chunk's digit counts per wave
  base[threadIdx.x] = totals[threadIdx.x] + offs[(size_t)threadIdx.x * n_blocks + blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned lo = blockIdx.x * (unsigned)kRsTile;
  for (int k = 0; k < kRsChunks; k++) {
    const unsigned c0 = lo + (unsigned)k * kRsBlock;
    if (c0 >= n) break;  // (block-uniform)
#pragma unroll
    for (int w = 0; w < kRsBlock / 64; w++) wcnt[w][threadIdx.x] = 0;
    __syncthreads();  // (also: base[] of the previous chunk / of the prologue is in place)
    const unsigned i = c0 + threadIdx.x;
    const bool live = i < n;
    rs_u64 key = 0;
    V val{};
    if (live) { key = keys[i]; if (HAS_V) val = vals[i]; }
    const unsigned d = (unsigned)(key >> shift) & mask;
    // lanes of this wave holding the same digit (dead lanes match nobody)
    rs_u64 same = __ballot(live);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const rs_u64 bal = __ballot(live && ((d >> b) & 1u));
      same &= ((d >> b) & 1u) ? bal : ~bal;
    }
    const unsigned rank = (unsigned)__popcll(same & ((1ull << lane) - 1ull));
    if (live && rank == 0) wcnt[wave][d] = (unsigned)__popcll(same);
    __syncthreads();
    unsigned at = 0;
    if (live) {
      at = base[d] + rank;
      for (int w = 0; w < wave; w++) at += wcnt[w][d];
    }
    __syncthreads();  // every lane has read base[] before it moves on
    {
      unsigned t = 0;
#pragma unroll
      for (int w = 0; w < kRsBlock / 64; w++) t += wcnt[w][threadIdx.x];
      base[threadIdx.x] += t;
    }
    if (live) { keys_out[at] = key; if (HAS_V) vals_out[at] = val; }
    // (the next chunk's first barrier separates this base[] update and the wcnt[] reads above from its writes)
    __syncthreads();
  }
}

// bytes of `hist` scratch a sort of n keys needs
inline size_t rs_hist_bytes(size_t n) { return (size_t)256 * ((n + kRsTile - 1) / kRsTile + 1) * sizeof(unsigned) + 16; }  // counts + the 256 digit totals

// keys_in / vals_in are left untouched; the result lands in keys_out / vals_out; keys_tmp / vals_tmp: n entries each
// (vals_*: null for a key-only sort). Bits [begin_bit, end_bit) take part. Everything is enqueued on `st`.
inline int rs_passes(int begin_bit, int end_bit) { return (end_bit - begin_bit + 7) / 8; }
// A caller that spreads a chain of launches over several of its own steps numbers the launches as they come and lets through
// those of the step at hand: gate() is asked once per launch, in a fixed order.
struct RsGate {
  int next = 0, from = 0, to = 1 << 30;
  bool operator()() { const int u = next++; return u >= from && u <= to; }
};
template <class V>
inline hipError_t rs_sort(const rs_u64* keys_in, rs_u64* keys_out, rs_u64* keys_tmp, const V* vals_in, V* vals_out, V* vals_tmp, size_t n, int begin_bit, int end_bit,
                          unsigned* hist, hipStream_t st, RsGate* gate = nullptr) {
  if (n == 0) return hipSuccess;
  if (n >= ((size_t)1 << 31)) return hipErrorInvalidValue;
  const bool has_v = vals_in != nullptr;
  const int passes = (end_bit - begin_bit + 7) / 8;
  if (passes <= 0) {
    hipError_t e = hipMemcpyAsync(keys_out, keys_in, n * sizeof(rs_u64), hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess && has_v) e = hipMemcpyAsync(vals_out, vals_in, n * sizeof(V), hipMemcpyDeviceToDevice, st);
    return e;
  }
  const unsigned n_blocks = (unsigned)((n + kRsTile - 1) / kRsTile);
  const rs_u64* src_k = keys_in;
  const V* src_v = vals_in;
  for (int p = 0; p < passes; p++) {
    const bool to_out = ((passes - 1 - p) & 1) == 0;  // the last pass writes the caller's output
    rs_u64* dst_k = to_out ? keys_out : keys_tmp;
    V* dst_v = to_out ? vals_out : vals_tmp;
    const int shift = begin_bit + 8 * p;
    const unsigned mask = (1u << (end_bit - shift < 8 ? end_bit - shift : 8)) - 1u;  // the last pass may hold fewer than eight bits
    auto go = [&]() { return gate ? (*gate)() : true; };
    if (go()) hipLaunchKernelGGL(rs_histogram_kernel, dim3(n_blocks), dim3(kRsBlock), 0, st, src_k, (unsigned)n, shift, mask, n_blocks, hist);
    unsigned* const totals = hist + (size_t)256 * n_blocks;
    if (go()) hipLaunchKernelGGL(rs_scan_digit_kernel, dim3(256), dim3(kRsBlock), 0, st, hist, n_blocks, totals);
    if (go()) hipLaunchKernelGGL(rs_scan_totals_kernel, dim3(1), dim3(kRsBlock), 0, st, totals);
    if (go()) {
      if (has_v) hipLaunchKernelGGL((rs_scatter_kernel<V, true>), dim3(n_blocks), dim3(kRsBlock), 0, st, src_k, src_v, (unsigned)n, shift, mask, n_blocks, hist, totals, dst_k, dst_v);
      else hipLaunchKernelGGL((rs_scatter_kernel<V, false>), dim3(n_blocks), dim3(kRsBlock), 0, st, src_k, (const V*)nullptr, (unsigned)n, shift, mask, n_blocks, hist, totals, dst_k, (V*)nullptr);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    src_k = dst_k; src_v = dst_v;
  }
  return hipSuccess;
}

// ---- inclusive running maximum of n 64-bit values (three dispatches: tile maxima, their prefix, the tiles again) --------
constexpr int kRmTile = 4096;
__global__ __launch_bounds__(kRsBlock) void rm_tile_max_kernel(const rs_u64* in, unsigned n, rs_u64* tile_max) {
  __shared__ rs_u64 sh[kRsBlock / 64];
  const unsigned lo = blockIdx.x * (unsigned)kRmTile;
  rs_u64 m = 0;
  for (unsigned i = lo + threadIdx.x; i < min(n, lo + (unsigned)kRmTile); i += kRsBlock) m = in[i] > m ? in[i] : m;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const rs_u64 v = __shfl_xor(m, d); m = v > m ? v : m; }
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) { for (int w = 1; w < kRsBlock / 64; w++) m = sh[w] > m ? sh[w] : m; tile_max[blockIdx.x] = m; }
}
// tile_max[t] becomes the maximum of the tiles BEFORE t (0 for the first); one block
__global__ __launch_bounds__(kRsScanBlock) void rm_scan_tiles_kernel(rs_u64* tile_max, unsigned n_tiles) {
  __shared__ rs_u64 wave_max[kRsScanBlock / 64];
  const unsigned per = (n_tiles + kRsScanBlock - 1) / kRsScanBlock;
  const unsigned lo = min(n_tiles, threadIdx.x * per), hi = min(n_tiles, lo + per);
  rs_u64 own = 0;
  for (unsigned i = lo; i < hi; i++) own = tile_max[i] > own ? tile_max[i] : own;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  rs_u64 incl = own;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const rs_u64 v = __shfl_up(incl, d); if (lane >= d) incl = v > incl ? v : incl; }
  rs_u64 excl = __shfl_up(incl, 1);
  if (lane == 0) excl = 0;
  if (lane == 63) wave_max[wave] = incl;
  __syncthreads();
  rs_u64 run = excl;
  for (int w = 0; w < wave; w++) run = wave_max[w] > run ? wave_max[w] : run;
  for (unsigned i = lo; i < hi; i++) { const rs_u64 v = tile_max[i]; tile_max[i] = run; run = v > run ? v : run; }
}
__global__ __launch_bounds__(kRsBlock) void rm_apply_kernel(const rs_u64* in, unsigned n, const rs_u64* tile_before, rs_u64* out) {
  __shared__ rs_u64 wave_max[kRsBlock / 64];
  __shared__ rs_u64 carry_sh;
  const unsigned lo = blockIdx.x * (unsigned)kRmTile;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_sh = tile_before[blockIdx.x];
  __syncthreads();
  for (unsigned c0 = lo; c0 < min(n, lo + (unsigned)kRmTile); c0 += kRsBlock) {  // (block-uniform)
    const unsigned i = c0 + threadIdx.x;
    const rs_u64 v = i < n ? in[i] : 0;
    rs_u64 incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const rs_u64 u = __shfl_up(incl, d); if (lane >= d) incl = u > incl ? u : incl; }
    if (lane == 63) wave_max[wave] = incl;
    __syncthreads();
    rs_u64 r = carry_sh > incl ? carry_sh : incl;
    for (int w = 0; w < wave; w++) r = wave_max[w] > r ? wave_max[w] : r;
    if (i < n) out[i] = r;
    __syncthreads();  // everyone has read carry_sh and wave_max
    if (threadIdx.x == kRsBlock - 1) carry_sh = r;  // the chunk's last lane holds the running maximum so far
    __syncthreads();
  }
}
inline size_t rm_scratch_bytes(size_t n) { return ((n + kRmTile - 1) / kRmTile) * sizeof(rs_u64) + 16; }
inline hipError_t rm_inclusive_max(const rs_u64* in, rs_u64* out, size_t n, rs_u64* scratch, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const unsigned n_tiles = (unsigned)((n + kRmTile - 1) / kRmTile);
  hipLaunchKernelGGL(rm_tile_max_kernel, dim3(n_tiles), dim3(kRsBlock), 0, st, in, (unsigned)n, scratch);
  hipLaunchKernelGGL(rm_scan_tiles_kernel, dim3(1), dim3(kRsScanBlock), 0, st, scratch, n_tiles);
  hipLaunchKernelGGL(rm_apply_kernel, dim3(n_tiles), dim3(kRsBlock), 0, st, in, (unsigned)n, scratch, out);
  return hipGetLastError();
}

}  // namespace gaml
