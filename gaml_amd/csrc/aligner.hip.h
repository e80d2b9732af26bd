// aligner.hip.h -- window alignment on the GPU (the cold path of CalcProb: reference
// AlignSubpathInternal graph.cc:839-899, ReadIndexMinHash::GetMinHashWithPoses / GetReadCandsWithPoses
// graph.cc:1289-1348, ProcessHit graph.cc:730-837). Same accept set, error counts and positions as
// the host aligner in host_model.cc (tests/test_gpu_aligner.py compares them record by record).
//
// Three kernels per batch of windows:
//   span_maxima_kernel   one block per (window, strand): scrambled 15-mer codes of the window
//                        string, sliding maximum over read-length spans, emission where it changes
//   candidates_kernel    one lane per emitted span: bucket of the max-hash index -> candidates
//   extend_kernel        one lane per candidate: locate the seed in the read, then the reference's
//                        0-1 BFS as a FIFO of chain heads with a per-diagonal visited bitset
// Records leave as (window, position, edit distance, read, strand, order) and are sorted / deduplicated
// per window on the host exactly like the host aligner does.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <climits>

namespace gaml {

#ifdef GAML_ALN_STAMPS
__device__ unsigned long long g_aln_stamp[32];  // timing builds only (aligner_small.hip.h)
#endif
constexpr int kAlnBlock = 256;
constexpr int kAlnSeed = 15;
constexpr int kAlnMaxRead = 254;  // visited bitset: 4 x 64 bits per diagonal

struct AlnWindow { int32_t str_off, len, offset; };   // window string in the batch buffer; offset = trimmed prefix (graph.cc:850)

struct AlnSpan { uint32_t hash; int32_t pos, win, strand, order; };  // emitted (hash, index of the seed's last base)
struct AlnCand { int32_t win, strand, order, seed_end, read; };
struct AlnHit { int32_t win, pos, edit, read, strand, order; };       // edit < 0: no alignment

__device__ __forceinline__ uint32_t aln_code(char c) {  // graph.h:326-331 (G0 A1 T2 C3; anything else 0)
  // branch-free: (c >> 1) & 3 tells A C T G apart (0 1 2 3), 0x2D holds their codes two bits each; a compare chain became
  // a chain of branches, fifteen of them per seed code
  const uint32_t known = (uint32_t)(c == 'A') | (uint32_t)(c == 'C') | (uint32_t)(c == 'T');
  return ((0x2Du >> ((((uint32_t)(unsigned char)c >> 1) & 3u) * 2u)) & 3u) * known;
}
__device__ __forceinline__ char aln_comp(char c) {  // graph.h:58-64
  return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c;
}
// base j of the window string on the given strand (strand 1 = reverse complement, graph.h:66-72)
__device__ __forceinline__ char aln_wbase(const char* s, int W, int strand, int j) {
  return strand == 0 ? s[j] : aln_comp(s[W - 1 - j]);
}

// ---------------------------------------------------------------------------------------------
// 1. sliding maximum (graph.cc:1289-1323). Emission rule: at i == R-1 and wherever the span maximum differs from
//    the previous span's; the position is the EARLIEST seed attaining the maximum. `order` of a span = the index of
//    its last base: increasing in emission order, which is all the later stages use it for.
//    One block per (window, strand, chunk of 256 span ends): the chunk's slice of the window string and the scrambled
//    15-mer codes of its spans live in LDS; a span's maximum is one scan of its R - 14 codes, its predecessor's
//    maximum is the neighbouring lane's. Chunks are independent, so a 30 kbp window is 2 x 117 blocks, not 2.
//    blk[w] = first block of window w (blk[n_win] = grid size): 2 x chunks(window length) blocks per window.
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline int aln_span_first(int R) { return R - 1 > kAlnSeed ? R - 1 : kAlnSeed; }  // the reference's loop starts at i = kIndexKmer
constexpr int kAlnSpans = kAlnBlock - 1;  // span ends per block: lane 0 only computes the predecessor of lane 1's span
__host__ __device__ inline int aln_span_chunks(int W, int R) { const int f = aln_span_first(R); return W > f ? (W - f + kAlnSpans - 1) / kAlnSpans : 0; }

// the bucket whose key is `hash`, or -1: top[h] = first bucket with key >= h << 16 (65537 entries), so the lower
// bound runs over the few dozen keys that share the hash's upper half -- 5 dependent loads instead of 21
__device__ __forceinline__ int aln_find_bucket(const uint64_t* bucket_hash, const int32_t* top, int n_buckets, uint32_t hash) {
  int lo = top[hash >> 16], hi = top[(hash >> 16) + 1];
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (bucket_hash[mid] < (uint64_t)hash) lo = mid + 1; else hi = mid; }
  return lo < n_buckets && bucket_hash[lo] == (uint64_t)hash ? lo : -1;
}

// Both mates of a paired read set in ONE small batch: windows [0, split) are mate 1's, [split, n) mate 2's (the same
// junction strings, looked up in the other mate's index and extended against the other mate's reads).
struct AlnHashSlot { uint32_t key; int32_t b0, cnt, used; };  // open addressing, linear probing (aln_hash_home); used = 0 ends a probe sequence
__host__ __device__ inline uint32_t aln_hash_home(uint32_t key, int bits) { return (key * 0x9E3779B1u) >> (32 - bits); }
struct AlnMates {
  const AlnHashSlot* htab[2];  // max-hash key -> its bucket (first entry in bucket_reads, size): 2^hbits slots
  int hbits[2];
  const uint64_t* bucket_hash[2];
  const int32_t* bucket_top[2];
  const int32_t* bucket_off[2];
  const int32_t* bucket_reads[2];
  int n_buckets[2];
  const char* reads[2];
  const int64_t* read_off[2];
  int split;
};

// One block's share of the sliding maximum: the chunk's slice of the window string and its scrambled codes staged in
// LDS, then per lane (span end i = base + threadIdx.x) the span maximum m, the earliest seed attaining it p, and whether
// the reference emits here. Returns false (block-uniform) when the block has no work.
constexpr int kAlnGrpSlots = (kAlnBlock + kAlnMaxRead + 2 + 7) / 8;
constexpr int kAlnGroups = (kAlnMaxRead - kAlnSeed + 1 + 7) / 8 + 1;  // whole groups of eight keys a span can hold
struct AlnSpanLds {
  char str[kAlnBlock + kAlnMaxRead + 2];
  unsigned long long key[8 * kAlnGrpSlots];             // code << 32 | ~position per seed end (>= kAlnBlock + kAlnMaxRead + 2)
  unsigned long long grp[kAlnGrpSlots];                 // maxima of aligned groups of eight keys
  uint32_t mx[kAlnBlock];
};
struct AlnSpanLane { bool emit; uint32_t m; int p, i, w, strand; };
// which window a block of the span grid belongs to: the last w with blk[w] <= blockIdx.x (blk, wins: global memory, or
// the kernel's own argument segment for a handful of windows -- AlnWinArgs)
template <class BlkT, class WinT>
__device__ __forceinline__ int aln_span_locate(const BlkT& blk, const WinT& wins, int n_win, AlnWindow& win, int& rel) {
  int lo = 0, hi = n_win;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (blk[mid] <= (int)blockIdx.x) lo = mid; else hi = mid; }
  win = wins[lo];
  rel = (int)blockIdx.x - blk[lo];
  return lo;
}
__device__ __forceinline__ bool aln_span_front(AlnSpanLds& L, const char* wstr, const AlnWindow win, const int w, const int rel, int R, AlnSpanLane& o, char* copy = nullptr) {
  const char* s = wstr + win.str_off;
  const int W = win.len;
  const int chunks = aln_span_chunks(W, R);
  if (chunks == 0 || rel >= 2 * chunks) return false;
  const int strand = rel / chunks, chunk = rel % chunks;
  const int base = aln_span_first(R) + chunk * kAlnSpans;     // span ends [base, base + 255): lane t holds span base - 1 + t
  // codes of the seeds ending at j in [c_lo, c_hi]: those of span base - 1 (lane 0: the predecessor of the block's first span) up to the last span's
  const int c_lo = max(base - 1 - R + kAlnSeed, kAlnSeed - 1), c_hi = min(base + kAlnSpans - 1, W - 1);
  const int s_lo = c_lo - (kAlnSeed - 1);                        // string bases [s_lo, c_hi]
  {  // (both rounds' loads requested before the first LDS store: c_hi - s_lo < 2 * kAlnBlock)
    const int k0 = threadIdx.x, k1 = threadIdx.x + kAlnBlock, n = c_hi - s_lo;
    const char a0 = k0 <= n ? aln_wbase(s, W, strand, s_lo + k0) : '\0', a1 = k1 <= n ? aln_wbase(s, W, strand, s_lo + k1) : '\0';
    if (k0 <= n) L.str[k0] = a0;
    if (k1 <= n) L.str[k1] = a1;
    if (copy && strand == 0) {  // the forward chunks of a window cover its whole string (chunk 0 starts at base 0)
      if (k0 <= n) copy[win.str_off + s_lo + k0] = a0;
      if (k1 <= n) copy[win.str_off + s_lo + k1] = a1;
    }
  }
  __syncthreads();
#ifdef GAML_ALN_STAMPS
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 2) atomicMax(&g_aln_stamp[22], (unsigned long long)wall_clock64());
#endif
  // scrambled code of the 15-mer ENDING at c_lo + k, as a key (code << 32 | ~position): the maximum key over a span is its
  // largest code at the EARLIEST position (graph.cc:1303-1321 keeps the first of equal codes). All fifteen bases of a
  // code are requested before the first is used (a load per base with its own wait: 3.7 us of this kernel).
  const int n_codes = c_hi - c_lo + 1;
  for (int k = threadIdx.x; k < n_codes; k += kAlnBlock) {
    unsigned char bs[kAlnSeed];
#pragma unroll
    for (int q = 0; q < kAlnSeed; q++) bs[q] = (unsigned char)L.str[k + q];
    uint32_t code = 0;
#pragma unroll
    for (int q = 0; q < kAlnSeed; q++) code = (code << 2) | aln_code((char)bs[q]);
    L.key[k] = ((unsigned long long)(code ^ 0x2204abcdu) << 32) | (uint32_t)~(uint32_t)(c_lo + k);
  }
  __syncthreads();
#ifdef GAML_ALN_STAMPS
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 2) atomicMax(&g_aln_stamp[23], (unsigned long long)wall_clock64());
#endif
  // maxima of aligned groups of eight keys; a span's maximum is then <= 7 keys at either end + <= 17 group maxima
  // instead of R - 14 keys (2.8 us)
  if ((int)threadIdx.x * 8 < n_codes) {
    unsigned long long v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { const unsigned long long t = L.key[threadIdx.x * 8 + u]; v[u] = (int)threadIdx.x * 8 + u < n_codes ? t : 0ull; }  // (L.key holds 8 x 64 slots)
    unsigned long long g8 = 0;
#pragma unroll
    for (int u = 0; u < 8; u++) g8 = v[u] > g8 ? v[u] : g8;
    L.grp[threadIdx.x] = g8;
  }
  __syncthreads();
  const int i = base - 1 + (int)threadIdx.x;
  unsigned long long best = 0;
  if (i < W) {
    const int xa = max(i - R + kAlnSeed, kAlnSeed - 1) - c_lo, xb = i - c_lo;  // the span's keys: [xa, xb]
    if (xa <= xb) {
      const int ga = (xa + 7) >> 3, gb = (xb + 1) >> 3;  // whole groups [ga, gb)
      unsigned long long e[14], gm[kAlnGroups];
      // (every load unconditional, its index clamped, the value masked afterwards: a predicated load compiles to a branch
      // around the load with a wait of its own)
#pragma unroll
      for (int u = 0; u < 7; u++) { const int x = xa + u; const unsigned long long v = L.key[min(x, xb)]; e[u] = (x <= xb && x < 8 * ga) ? v : 0ull; }       // before the first whole group
#pragma unroll
      for (int u = 0; u < 7; u++) { const int x = xb - u; const unsigned long long v = L.key[max(x, xa)]; e[7 + u] = (x >= xa && x >= 8 * gb) ? v : 0ull; }  // after the last (a key counted twice changes no maximum)
#pragma unroll
      for (int u = 0; u < kAlnGroups; u++) { const unsigned long long v = L.grp[min(ga + u, kAlnGrpSlots - 1)]; gm[u] = ga + u < gb ? v : 0ull; }
#pragma unroll
      for (int u = 0; u < 14; u++) best = e[u] > best ? e[u] : best;
#pragma unroll
      for (int u = 0; u < kAlnGroups; u++) best = gm[u] > best ? gm[u] : best;
    }
  }
  const uint32_t m = (uint32_t)(best >> 32);
  const int p = (int)~(uint32_t)best;  // (-1 when the span holds no seed)
  L.mx[threadIdx.x] = m;
  __syncthreads();
  // emission: the first full span, and wherever the maximum differs from the previous span's (== what the reference
  // last emitted; i - 1 < R - 1 can only happen when R - 1 < kAlnSeed: the host path handles such short reads)
  const bool emit = threadIdx.x > 0 && i < W && (i == R - 1 || m != L.mx[threadIdx.x - 1]);
  o = AlnSpanLane{emit, m, p, i, w, strand};
  return true;
}

__global__ __launch_bounds__(kAlnBlock) void span_maxima_kernel(const char* wstr, const AlnWindow* wins, int n_win, int R, const int* blk,
                                                               AlnSpan* spans, unsigned* n_spans, unsigned cap_spans) {
  __shared__ AlnSpanLds L;
  __shared__ int sh_wave[kAlnBlock / 64];
  __shared__ unsigned sh_base;
  AlnSpanLane o;
  AlnWindow win;
  int rel;
  const int w = aln_span_locate(blk, wins, n_win, win, rel);
  if (!aln_span_front(L, wstr, win, w, rel, R, o)) return;
  // ordered compaction inside the block: ballots per wave, wave totals through LDS
  const unsigned long long bal = __ballot(o.emit);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int before = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) sh_wave[wave] = __popcll(bal);
  __syncthreads();
  int wave_before = 0, total = 0;
  for (int q = 0; q < kAlnBlock / 64; q++) { if (q < wave) wave_before += sh_wave[q]; total += sh_wave[q]; }
  if (threadIdx.x == 0) sh_base = total ? atomicAdd(n_spans, (unsigned)total) : 0;
  __syncthreads();
  if (o.emit) {
    const unsigned at = sh_base + (unsigned)(wave_before + before);
    if (at < cap_spans) spans[at] = AlnSpan{o.m, o.p, o.w, o.strand, o.i};
  }
}

// ---------------------------------------------------------------------------------------------
// 2. candidates: every read of the bucket whose key is the span's hash (graph.cc:1329-1347)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kAlnBlock) void candidates_kernel(const AlnSpan* spans, const unsigned* n_spans, unsigned cap_spans,
                                                              const uint64_t* bucket_hash, const int32_t* bucket_top, const int32_t* bucket_off,
                                                              const int32_t* bucket_reads, int n_buckets, AlnCand* cands,
                                                              unsigned* n_cands, unsigned cap_cands) {
  const unsigned n = *n_spans < cap_spans ? *n_spans : cap_spans;
  for (unsigned t = blockIdx.x * kAlnBlock + threadIdx.x; t < n; t += gridDim.x * kAlnBlock) {
    const AlnSpan sp = spans[t];
    const int lo = aln_find_bucket(bucket_hash, bucket_top, n_buckets, sp.hash);
    if (lo < 0) continue;
    const int b0 = bucket_off[lo], b1 = bucket_off[lo + 1];
    const unsigned at = atomicAdd(n_cands, (unsigned)(b1 - b0));
    for (int k = b0; k < b1; k++) {
      const unsigned o = at + (unsigned)(k - b0);
      if (o < cap_cands) cands[o] = AlnCand{sp.win, sp.strand, sp.order, sp.pos, bucket_reads[k]};
    }
  }
}

// ---------------------------------------------------------------------------------------------
// 3. seed extension (ProcessHit graph.cc:753-837). A popped state slides down its diagonal while
//    bases match and the next cell is unvisited (the reference pushes such a step to the FRONT of its
//    deque, so it is popped next); at the first mismatch it appends its <= 3 successors at cost + 1.
//    Heads leave the FIFO in non-decreasing cost; cost > 3 ends the search. Visited cells: one
//    256-bit set per diagonal shift in [-4, 4].
// ---------------------------------------------------------------------------------------------
// One WAVE per candidate. A single lane running the search serially spends ~0.5 us per visited
// cell (dependent LDS / memory operations of a lone wave), i.e. ~0.3 ms per launch whatever the batch
// size; but nearly all cells of a search are runs of matching bases along one diagonal. The wave
// compares 64 cells of the current diagonal at once (one ballot finds the first cell where the run
// stops and why), marks the run in the visited set with a handful of word operations, and only the
// bookkeeping at a mismatch (<= 3 successors) is scalar. Same states, same order, same visited
// semantics as the FIFO-of-chain-heads search above, so the same records.
constexpr int kAlnQueue = 128;                 // heads: r+1 (8 bits) | diag+4 (4 bits) | cost (3 bits)
constexpr int kAlnWinSeg = kAlnMaxRead + 18;   // window bases a search can touch (read length + 2 x 4 diagonals + slack)
constexpr int kAlnWaves = 4;                   // candidates per block
template <int WS>
struct AlnWaveLdsT {
  unsigned char rd[kAlnMaxRead + 2];           // the read as aligned (strand applied)
  unsigned char ws[WS];                        // window segment, 0 beyond the window's end
  unsigned char seed[16];
  uint32_t vis[9 * 8];                         // 256-bit visited set per diagonal shift in [-4, 4]; bit = read index + 1
  unsigned short q[kAlnQueue];
};
using AlnWaveLds = AlnWaveLdsT<kAlnWinSeg + 2>;

__device__ __forceinline__ void aln_lds_sync() {  // LDS traffic of ONE wave: in order in hardware, keep the compiler from reordering
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <class Lds>
struct AlnWaveSearch {
  Lds& L;
  const int lane;
  int qn;
  __device__ __forceinline__ AlnWaveSearch(Lds& l, int ln) : L(l), lane(ln), qn(0) {}
  __device__ __forceinline__ void reset() {
    qn = 0;
    L.vis[lane] = 0;
    if (lane < 8) L.vis[64 + lane] = 0;
    aln_lds_sync();
  }
  // wave-uniform arguments: true if (diag, r) was unvisited (then it is marked)
  __device__ __forceinline__ bool mark(int diag, int r) {
    const int b = r + 1;
    const uint32_t m = 1u << (b & 31);
    const int w = (diag + 4) * 8 + (b >> 5);
    const uint32_t old = L.vis[w];
    if (old & m) return false;
    if (lane == 0) L.vis[w] = old | m;
    aln_lds_sync();
    return true;
  }
  __device__ __forceinline__ void push(int d, int diag, int r) {
    if (qn < kAlnQueue) {
      if (lane == 0) L.q[qn] = (unsigned short)((r + 1) | ((diag + 4) << 8) | (d << 12));
      qn++;
    }
  }
  __device__ __forceinline__ bool visited(int diag, int r) const {  // per-lane r
    const int b = r + 1;
    return (L.vis[(diag + 4) * 8 + (b >> 5)] >> (b & 31)) & 1u;
  }
  // mark read indices [r_lo, r_hi] (inclusive) of one diagonal: lanes 0..7 own one word each
  __device__ __forceinline__ void mark_range(int diag, int r_lo, int r_hi) {
    if (lane < 8 && r_hi >= r_lo) {
      const int lo = r_lo + 1, hi = r_hi + 1;  // bit range, inclusive
      const int w0 = 32 * lane;
      const int a = max(lo, w0), b = min(hi, w0 + 31);
      if (a <= b) {
        const uint32_t m = (b - a == 31 ? 0xffffffffu : ((1u << (b - a + 1)) - 1u)) << (a - w0);
        L.vis[(diag + 4) * 8 + lane] |= m;
      }
    }
    aln_lds_sync();
  }
  // The end of a head's run in ONE round of LDS reads: the run's cells [r_lo, r_hi] of `diag` are marked, and -- when the
  // run ended in a mismatch (`succ`) -- its up to three successors are tested and marked in the order the reference
  // pushes them: S = (diag, rs), W = (dw, rw) (both only if `sw`), R = (dr, rr). Every word involved is requested
  // before any is used; the separate mark_range + 3 x mark took seven dependent LDS round trips per head. Same final
  // visited set: the successors' bits never lie inside the run's range. Returns bit 0 / 1 / 2 = S / W / R was unvisited.
  __device__ __forceinline__ unsigned finish_run(int diag, int r_lo, int r_hi, bool succ, bool sw, int rs, int dw, int rw, int dr, int rr) {
    const int bs = rs + 1, bw = rw + 1, br = rr + 1;
    const int iw = (dw + 4) * 8 + (bw >> 5), ir = (dr + 4) * 8 + (br >> 5);
    uint32_t own = 0, ww = ~0u, wr = ~0u;
    if (lane < 8) own = L.vis[(diag + 4) * 8 + lane];
    if (succ) { if (sw) ww = L.vis[iw]; wr = L.vis[ir]; }
    // S lives in one of the run's own words (lane bs >> 5)
    const uint32_t s_word = (uint32_t)__builtin_amdgcn_readlane((int)own, __builtin_amdgcn_readfirstlane(bs >> 5) & 7);
    const bool s_new = succ && sw && !((s_word >> (bs & 31)) & 1u);
    const bool w_new = succ && sw && !((ww >> (bw & 31)) & 1u);
    const bool r_new = succ && !((wr >> (br & 31)) & 1u);
    if (lane < 8) {
      uint32_t m = 0;
      if (r_hi >= r_lo) {
        const int lo = r_lo + 1, hi = r_hi + 1, w0 = 32 * lane;  // bit range, inclusive
        const int a = max(lo, w0), b = min(hi, w0 + 31);
        if (a <= b) m = (b - a == 31 ? 0xffffffffu : ((1u << (b - a + 1)) - 1u)) << (a - w0);
      }
      if (s_new && lane == (bs >> 5)) m |= 1u << (bs & 31);
      if (m) L.vis[(diag + 4) * 8 + lane] = own | m;
    }
    if (lane == 0) {
      if (w_new) L.vis[iw] = ww | (1u << (bw & 31));
      if (r_new) L.vis[ir] = wr | (1u << (br & 31));
    }
    return (s_new ? 1u : 0u) | (w_new ? 2u : 0u) | (r_new ? 4u : 0u);
  }
  // graph.cc:761-793: cost of the cheapest chain from the seed's end to the read's end, or -1. wbase(g) = window base g.
  template <class WB>
  __device__ __forceinline__ int forward(const WB& wbase, const int R, const int W, const int win_pos, const int read_pos) {
    reset();
    push(0, 0, read_pos + kAlnSeed);
    aln_lds_sync();
    for (int qi = 0; qi < qn; qi++) {
      const uint32_t e = L.q[qi];
      const int d = (int)(e >> 12), diag = (int)((e >> 8) & 15) - 4;
      int r = (int)(e & 255) - 1;
      int g = win_pos + (r - read_pos) + diag;
      if (d > 3) return -1;
      while (true) {
        // lane k looks at cell (g + k, r + k) of the diagonal
        const int rr = r + lane, gg = g + lane;
        const bool at_end = rr == R;
        // (the three LDS reads unconditional and side by side, indices clamped; `||` chains compile to a branch per load)
        const bool in_read = rr < R;
        const unsigned char wb = wbase(gg), rb = L.rd[min(rr, kAlnMaxRead)];
        const bool seen = visited(diag, min(rr, kAlnMaxRead) + 1);
        const bool may_advance = (gg + 1 < W) | (rr + 1 == R);
        const bool mism = in_read & (wb != rb);
        const bool stop = at_end | (in_read & ((wb != rb) | !may_advance | seen));
        const unsigned long long stops = __ballot(stop & (rr <= R));
        if (!stops) {  // 64 matching, unvisited cells: the run goes on
          mark_range(diag, r + 1, r + 64);
          g += 64; r += 64;
          continue;
        }
        const int k = __ffsll((long long)stops) - 1;
        const bool end_here = (__ballot(at_end) >> k) & 1ull;
        const bool mism_here = (__ballot(mism) >> k) & 1ull;
        if (end_here) return d;
        // the cells the run moved into, and at a mismatch the successors: substitution, window base skipped, read base skipped
        const int r0 = r + 1;
        g += k; r += k;
        const unsigned fresh = finish_run(diag, r0, r, mism_here, g + 1 < W, r + 1, diag + 1, r, diag - 1, r + 1);
        if (fresh & 1u) push(d + 1, diag, r + 1);
        if (fresh & 2u) push(d + 1, diag + 1, r);
        if (fresh & 4u) push(d + 1, diag - 1, r + 1);
        aln_lds_sync();
        break;  // mismatch handled, or the run ended at the window's edge / a visited cell
      }
    }
    return -1;
  }
  // graph.cc:794-835: cost from the seed's start back to the read's start and where the alignment begins, or -1
  template <class WB>
  __device__ __forceinline__ int backward(const WB& wbase, const int W, const int win_pos, const int read_pos, int& begin_pos) {
    begin_pos = -1;
    if (win_pos == 0) return read_pos < 6 ? read_pos : -1;
    reset();
    push(0, 0, read_pos - 1);
    aln_lds_sync();
    for (int qi = 0; qi < qn; qi++) {
      const uint32_t e = L.q[qi];
      const int d = (int)(e >> 12), diag = (int)((e >> 8) & 15) - 4;
      int r = (int)(e & 255) - 1;
      int g = win_pos + (r - read_pos) + diag;
      if (d > 3) return -1;
      while (true) {
        // lane k looks at cell (g - k, r - k)
        const int rr = r - lane, gg = g - lane;
        const bool at_end = rr == -1;
        const bool in_read = rr >= 0;
        const unsigned char wb = wbase(gg), rb = L.rd[max(rr, 0)];
        const bool seen = visited(diag, max(rr, 0) - 1);
        const bool may_advance = (gg - 1 >= 0) | (rr - 1 == -1);
        const bool mism = in_read & (wb != rb);
        const bool stop = at_end | (in_read & ((wb != rb) | !may_advance | seen));
        const unsigned long long stops = __ballot(stop & (rr >= -1));
        if (!stops) {
          mark_range(diag, r - 64, r - 1);
          g -= 64; r -= 64;
          continue;
        }
        const int k = __ffsll((long long)stops) - 1;
        const bool end_here = (__ballot(at_end) >> k) & 1ull;
        const bool mism_here = (__ballot(mism) >> k) & 1ull;
        const int r1 = r - 1;
        g -= k; r -= k;
        if (end_here) { begin_pos = g + 1; return d; }
        const unsigned fresh = finish_run(diag, r, r1, mism_here, g - 1 >= 0, r - 1, diag - 1, r, diag + 1, r - 1);
        if (fresh & 1u) push(d + 1, diag, r - 1);
        if (fresh & 2u) push(d + 1, diag - 1, r);
        if (fresh & 4u) push(d + 1, diag + 1, r - 1);
        aln_lds_sync();
        break;
      }
    }
    return -1;
  }
};

// one candidate, one wave (every `return` leaves the candidate, not the kernel)
__device__ __forceinline__ void extend_candidate(AlnWaveLds& L, const int lane, const unsigned t, const AlnCand* cands, const char* wstr,
                                                 const AlnWindow* wins, const char* reads, const int64_t* read_off, AlnHit* hits) {
  const AlnCand c = cands[t];
  AlnHit out{c.win, 0, -1, c.read, c.strand, c.order};
  const AlnWindow win = wins[c.win];
  const char* ws = wstr + win.str_off;  // ProcessHit always works on the FORWARD window string
  const int W = win.len;
  const char* rd = reads + read_off[c.read];
  const int R = (int)(read_off[c.read + 1] - read_off[c.read]);
  if (R > kAlnMaxRead || R < kAlnSeed) { if (lane == 0) hits[t] = out; return; }
  // seed start in the forward window string (graph.cc:866-872)
  const int win_pos = c.strand == 0 ? c.seed_end - kAlnSeed + 1 : W - (c.seed_end + 1);
  // the read as aligned: strand 1 = reverse complement of the stored read (graph.cc:873-876); coalesced
  for (int b = lane; b < R; b += 64) {
    const char ch = rd[b];
    if (c.strand == 0) L.rd[b] = (unsigned char)ch;
    else L.rd[R - 1 - b] = (unsigned char)aln_comp(ch);
  }
  if (lane < kAlnSeed) L.seed[lane] = (unsigned char)ws[win_pos + lane];
  aln_lds_sync();
  // first position of the (oriented) read carrying the window's seed (graph.cc:873-879): 64 positions at a time
  int read_pos = -1;
  for (int base = 0; base + kAlnSeed <= R && read_pos < 0; base += 64) {
    const int i = base + lane;
    bool same = i + kAlnSeed <= R;
#pragma unroll
    for (int k = 0; k < kAlnSeed; k++) same = same & (L.rd[min(i + k, kAlnMaxRead)] == L.seed[k]);
    const unsigned long long hit = __ballot(same);
    if (hit) read_pos = base + (__ffsll((long long)hit) - 1);
  }
  if (read_pos < 0) { if (lane == 0) hits[t] = out; return; }
  // window bases the search can touch: g = win_pos + (r - read_pos) + diag, r in [-1, R], |diag| <= 4
  const int g0 = max(0, win_pos - read_pos - 6);
  const int seg = min(kAlnWinSeg, win_pos + (R - read_pos) + 6 - g0);
  for (int b = lane; b < seg; b += 64) L.ws[b] = (unsigned char)(g0 + b < W ? ws[g0 + b] : '\0');  // the reference reads the terminator at g == W
  aln_lds_sync();
  auto wbase = [&](int g) -> unsigned char { const int i = g - g0; const unsigned char v = L.ws[min(max(i, 0), kAlnWinSeg)]; return (i >= 0 && i < seg) ? v : (unsigned char)'\0'; };
  AlnWaveSearch<AlnWaveLds> S(L, lane);
  const int fwd = S.forward(wbase, R, W, win_pos, read_pos);
  if (fwd < 0) { if (lane == 0) hits[t] = out; return; }
  int begin_pos = -1;
  const int bwd = S.backward(wbase, W, win_pos, read_pos, begin_pos);
  if (bwd < 0) { if (lane == 0) hits[t] = out; return; }
  out.pos = begin_pos + 1 + win.offset;  // graph.cc:890
  out.edit = fwd + bwd;
  if (lane == 0) hits[t] = out;
}

// grid-stride over the candidates: the grid does not depend on their number, so the launch needs no count on the host
__global__ __launch_bounds__(64 * kAlnWaves) void extend_kernel(const AlnCand* cands, const unsigned* n_cands, unsigned cap_cands,
                                                                const char* wstr, const AlnWindow* wins, const char* reads,
                                                                const int64_t* read_off, AlnHit* hits) {
  __shared__ AlnWaveLds lds_all[kAlnWaves];
  const unsigned n = *n_cands < cap_cands ? *n_cands : cap_cands;
  const int lane = (int)(threadIdx.x & 63);
  AlnWaveLds& L = lds_all[threadIdx.x >> 6];
  for (unsigned t = blockIdx.x * kAlnWaves + (threadIdx.x >> 6); t < n; t += gridDim.x * kAlnWaves) {  // whole waves move together
    extend_candidate(L, lane, t, cands, wstr, wins, reads, read_off, hits);
    aln_lds_sync();  // the wave's LDS slice is reused by its next candidate
  }
}

// Small batches (an annealing move's handful of new junction windows): the counters and the hits go to mapped pinned
// host memory and a sequence word tells the host they are there -- ONE wait per batch, no copy commands. The counters
// are left at zero for the next batch.
__global__ __launch_bounds__(256) void publish_hits_kernel(unsigned* counters, const AlnHit* hits, unsigned cap_cands, unsigned* h_counts,
                                                           AlnHit* h_hits, unsigned cap_host, volatile unsigned long long* h_seq, unsigned long long seq) {
  const unsigned n_spans = counters[0], n_cands = counters[1];
  const unsigned n = n_cands <= cap_cands ? (n_cands <= cap_host ? n_cands : 0u) : 0u;  // overflow: counts only, the host takes the slow route
  for (unsigned t = threadIdx.x; t < n; t += 256) h_hits[t] = hits[t];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    h_counts[0] = n_spans; h_counts[1] = n_cands;
    counters[0] = 0; counters[1] = 0;
    __threadfence_system();
    *h_seq = seq;
  }
}


// ---------------------------------------------------------------------------------------------
// 4. large batches: put the hits in the host's order on the device -- (window, position, read, strand,
//    order), failed extensions last -- with two stable radix sorts (the whole key does not fit 64 bits).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hit_keys_kernel(const AlnHit* hits, unsigned n, unsigned long long* key_minor,
                                                       unsigned long long* key_major, unsigned* idx, unsigned* n_ok) {
  for (unsigned t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) {
    const AlnHit h = hits[t];
    idx[t] = t;
    if (h.edit < 0) { key_minor[t] = ~0ull; key_major[t] = ~0ull; continue; }
    key_minor[t] = ((unsigned long long)(unsigned)h.read << 25) | ((unsigned long long)(h.strand & 1) << 24) | (unsigned long long)(h.order & 0xffffff);
    key_major[t] = ((unsigned long long)(unsigned)h.win << 32) | (unsigned long long)(unsigned)h.pos;
    atomicAdd(n_ok, 1u);
  }
}
__global__ __launch_bounds__(256) void gather_u64_kernel(const unsigned long long* src, const unsigned* idx, unsigned n, unsigned long long* dst) {
  for (unsigned t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) dst[t] = src[idx[t]];
}
__global__ __launch_bounds__(256) void gather_hits_kernel(const AlnHit* src, const unsigned* idx, unsigned n, AlnHit* dst) {
  for (unsigned t = blockIdx.x * 256 + threadIdx.x; t < n; t += gridDim.x * 256) dst[t] = src[idx[t]];
}

}  // namespace gaml
