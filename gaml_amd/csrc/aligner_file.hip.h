// aligner_file.hip.h -- a small alignment batch's hits filed ON THE DEVICE: ordered per window by (position, read), the first
// alignment found for a key survives (the reference collects into a set ordered by (position, read): graph.cc:841, 891,
// 895-897; per read its candidates are visited forward-strand spans first), appended to the mates' window-major record
// pools. What goes back to the host is a window's HEADER -- count, largest position (the position filter graph.cc:577 needs
// it for the occurrence tables) -- not its records.
//   aln_file_small_kernel   one block: keys (window, position | read, strand, order) sorted in LDS (bitonic; the keys are
//                           distinct), duplicates of a (window, position, read) dropped, a prefix sum places the
//                           survivors, per-window counts / maxima by LDS atomics; publishes behind a sequence word.
#pragma once
#include "aligner_small.hip.h"

namespace gaml {

constexpr int kFileThreads = 1024, kFileMaxHits = 2048, kFileMaxWins = 64;
struct AlnFileArgs {
  int n_win, split;                 // windows of the batch; [split, n_win) belong to mate 2
  int pool_base[2];                 // where each mate's new records start in its pool
  int pool_cap[2];                  // records the pools hold
  int4* pool[2];
  int wid[kFileMaxWins];            // the batch window's id in its mate's window table
};
// host view of a published batch: [0] candidates, [1] status (0 filed, 1 not filed: too many hits / pool full), [2..3] records
// appended per mate, then per window {count, largest position}
struct AlnFileOut { unsigned n_cands, status, added[2]; int hdr[kFileMaxWins][2]; };

__global__ __launch_bounds__(kFileThreads) void aln_file_small_kernel(AlnFileArgs a, const AlnHit* d_hits, unsigned* counters, unsigned cap_cands, AlnFileOut* h_out,
                                                                    volatile unsigned long long* h_seq, unsigned long long seq) {
  __shared__ unsigned long long khi[kFileMaxHits], klo[kFileMaxHits];
  __shared__ int w_cnt[kFileMaxWins], w_max[kFileMaxWins];
  __shared__ int sc[kFileThreads / 64];
  __shared__ int n_mate0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthr = (int)blockDim.x;  // (as many threads as the batch is likely to need: the network's barriers cost by the wave)
  const unsigned n_cands = counters[1];
  const bool fits = n_cands <= cap_cands && n_cands <= (unsigned)kFileMaxHits;
  const int n = fits ? (int)n_cands : 0;
  if (tid < kFileMaxWins) { w_cnt[tid] = 0; w_max[tid] = INT_MIN; }
  if (tid == 0) n_mate0 = 0;
  int N = 64;
  while (N < n) N <<= 1;
  for (int i = tid; i < N; i += nthr) {
    unsigned long long hi = ~0ull, lo = ~0ull;
    if (i < n) {
      const AlnHit h = d_hits[i];
      if (h.edit >= 0) {
        hi = ((unsigned long long)(unsigned)h.win << 32) | (unsigned)(h.pos + 0x40000000);
        lo = ((unsigned long long)(unsigned)h.read << 32) | ((unsigned long long)(h.strand & 1) << 31) | ((unsigned long long)(h.order & 0xffffff) << 7) | (unsigned)(h.edit & 0x7f);
      }
    }
    khi[i] = hi; klo[i] = lo;
  }
  __syncthreads();
  for (int k = 2; k <= N; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (N >> 1); t += nthr) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
        const unsigned long long xh = khi[i], xl = klo[i], yh = khi[l], yl = klo[l];
        const bool gt = xh != yh ? xh > yh : xl > yl;
        if (gt == ((i & k) == 0)) { khi[i] = yh; klo[i] = yl; khi[l] = xh; klo[l] = xl; }
      }
      // (strides below 64: a wave's exchanges stay inside the 128 keys it owns -- no block barrier, see dl_bitonic_sort)
      const int jn = j > 1 ? j >> 1 : k;
      if (j >= 64 || jn >= 64) __syncthreads();
      else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  }
  __syncthreads();
  // survivors: the first of every (window, position, read); thread t owns the sorted positions [t * per, (t + 1) * per)
  const int per = (N + nthr - 1) / nthr;
  const int p_lo = tid * per, p_hi = min(N, p_lo + per);
  auto survives = [&](int p) { return khi[p] != ~0ull && (p == 0 || khi[p - 1] != khi[p] || (klo[p - 1] >> 32) != (klo[p] >> 32)); };
  int mine = 0, mine0 = 0;
  for (int p = p_lo; p < p_hi; p++)
    if (survives(p)) { mine++; mine0 += (int)(khi[p] >> 32) < a.split; }
  int incl = mine;
  for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(incl, d, 64); if (lane >= d) incl += u; }
  if (lane == 63) sc[wave] = incl;
  if (mine0) atomicAdd(&n_mate0, mine0);
  __syncthreads();
  int before = incl - mine, total = 0;
  for (int w = 0; w < (nthr >> 6); w++) { if (w < wave) before += sc[w]; total += sc[w]; }
  const int add0 = n_mate0, add1 = total - n_mate0;
  const bool room = a.pool_base[0] + add0 <= a.pool_cap[0] && a.pool_base[1] + add1 <= a.pool_cap[1];
  const bool filed = fits && room && a.n_win <= kFileMaxWins;
  if (filed) {
    int rank = before;  // among all survivors, in (window, position, read) order: mate 1's windows come first in it
    for (int p = p_lo; p < p_hi; p++) {
      if (!survives(p)) continue;
      const int win = (int)(khi[p] >> 32), pos = (int)(unsigned)khi[p] - 0x40000000;
      const int read = (int)(klo[p] >> 32), strand = (int)((klo[p] >> 31) & 1), edit = (int)(klo[p] & 0x7f);
      const int mt = win >= a.split ? 1 : 0;
      const int at = mt == 0 ? a.pool_base[0] + rank : a.pool_base[1] + (rank - add0);
      a.pool[mt][at] = make_int4(a.wid[win], pos, edit | (strand << 8), read);
      atomicAdd(&w_cnt[win], 1);
      atomicMax(&w_max[win], pos);
      rank++;
    }
  }
  __syncthreads();
  if (tid < kFileMaxWins) {
    h_out->hdr[tid][0] = filed && tid < a.n_win ? w_cnt[tid] : 0;
    h_out->hdr[tid][1] = filed && tid < a.n_win ? w_max[tid] : INT_MIN;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {  // publish, and leave the aligner's counters at zero for the next batch
    h_out->n_cands = n_cands; h_out->status = filed ? 0u : 1u; h_out->added[0] = filed ? (unsigned)add0 : 0u; h_out->added[1] = filed ? (unsigned)add1 : 0u;
    counters[0] = 0; counters[1] = 0; counters[kAlnTicketWord] = 0;
    __threadfence_system();
    __hip_atomic_store((unsigned long long*)h_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace gaml

namespace gaml {

// ---- large batches (the cold first evaluation: every window of the start assembly, millions of hits) -------------------------
// The hits are already ordered on the device by (window, position, read, strand, order) with the failed extensions last
// (two stable radix sorts, aligner_launch.hip.h). Filing them there too: a flag per hit "first of its (window, position,
// read)" (graph.cc:841, 891, 895-897), an exclusive prefix sum of the flags (table_build.hip.h's scan), then the survivors
// written into the mate's pool and counted per window -- instead of 67 MB of hits to the host, 20 ms of host filing and
// the records back up into the pool.
__global__ __launch_bounds__(256) void aln_file_flags_kernel(const AlnHit* hits, const unsigned* n_ok, unsigned n, int* flags) {
  const unsigned ok = *n_ok;
  for (unsigned t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) {
    int f = 0;
    if (t < ok) {
      const AlnHit h = hits[t];
      f = 1;
      if (t > 0) { const AlnHit p = hits[t - 1]; f = !(p.win == h.win && p.pos == h.pos && p.read == h.read); }
    }
    flags[t] = f;
  }
}
__global__ __launch_bounds__(256) void aln_file_write_kernel(const AlnHit* hits, const int* flags, const int* place, unsigned n, int4* pool, int base, const int* wid_of,
                                                            int* win_cnt, int* win_max) {
  for (unsigned t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) {
    if (!flags[t]) continue;
    const AlnHit h = hits[t];
    pool[base + place[t]] = make_int4(wid_of[h.win], h.pos, (h.edit & 0xff) | ((h.strand & 1) << 8), h.read);
    atomicAdd(&win_cnt[h.win], 1);
    atomicMax(&win_max[h.win], h.pos);
  }
}

}  // namespace gaml
