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

constexpr int kAlnBlock = 256;
constexpr int kAlnSeed = 15;
constexpr int kAlnMaxRead = 254;  // visited bitset: 4 x 64 bits per diagonal

struct AlnWindow { int32_t str_off, len, offset; };   // window string in the batch buffer; offset = trimmed prefix (graph.cc:850)

struct AlnSpan { uint32_t hash; int32_t pos, win, strand, order; };  // emitted (hash, index of the seed's last base)
struct AlnCand { int32_t win, strand, order, seed_end, read; };
struct AlnHit { int32_t win, pos, edit, read, strand, order; };       // edit < 0: no alignment

__device__ __forceinline__ uint32_t aln_code(char c) {  // graph.h:326-331 (G0 A1 T2 C3; anything else 0)
  return c == 'A' ? 1u : c == 'T' ? 2u : c == 'C' ? 3u : 0u;
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
__host__ __device__ inline int aln_span_chunks(int W, int R) { const int f = aln_span_first(R); return W > f ? (W - f + kAlnBlock - 1) / kAlnBlock : 0; }

// the bucket whose key is `hash`, or -1: top[h] = first bucket with key >= h << 16 (65537 entries), so the lower
// bound runs over the few dozen keys that share the hash's upper half -- 5 dependent loads instead of 21
__device__ __forceinline__ int aln_find_bucket(const uint64_t* bucket_hash, const int32_t* top, int n_buckets, uint32_t hash) {
  int lo = top[hash >> 16], hi = top[(hash >> 16) + 1];
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (bucket_hash[mid] < (uint64_t)hash) lo = mid + 1; else hi = mid; }
  return lo < n_buckets && bucket_hash[lo] == (uint64_t)hash ? lo : -1;
}

// Both mates of a paired read set in ONE small batch: windows [0, split) are mate 1's, [split, n) mate 2's (the same
// junction strings, looked up in the other mate's index and extended against the other mate's reads).
struct AlnMates {
  const uint64_t* bucket_hash[2];
  const int32_t* bucket_top[2];
  const int32_t* bucket_off[2];
  const int32_t* bucket_reads[2];
  int n_buckets[2];
  const char* reads[2];
  const int64_t* read_off[2];
  int split;
};

// FUSED (small batches of both mates): an emitted span looks its bucket up right here and appends its candidates -- no
// span list, no second launch.
// (windows [split, n_win) belong to a second read set -- the other mate -- whose index was built for read length R2)
template <bool FUSED>
__global__ __launch_bounds__(kAlnBlock) void span_maxima_kernel(const char* wstr, const AlnWindow* wins, int n_win, int R, const int* blk,
                                                               AlnSpan* spans, unsigned* n_spans, unsigned cap_spans, int split, int R2,
                                                               AlnMates ix, AlnCand* cands, unsigned* n_cands, unsigned cap_cands) {
  __shared__ char sh_str[kAlnBlock + kAlnMaxRead + 2];
  __shared__ uint32_t sh_code[kAlnBlock + kAlnMaxRead + 2];
  __shared__ uint32_t sh_max[kAlnBlock];
  __shared__ int sh_wave[kAlnBlock / 64];
  __shared__ unsigned sh_base;
  int lo = 0, hi = n_win;  // the window this block belongs to: last w with blk[w] <= blockIdx.x
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (blk[mid] <= (int)blockIdx.x) lo = mid; else hi = mid; }
  const int w = lo;
  if (w >= split) R = R2;
  const AlnWindow win = wins[w];
  const char* s = wstr + win.str_off;
  const int W = win.len;
  const int chunks = aln_span_chunks(W, R);
  const int rel = (int)blockIdx.x - blk[w];
  if (chunks == 0 || rel >= 2 * chunks) return;
  const int strand = rel / chunks, chunk = rel % chunks;
  const int base = aln_span_first(R) + chunk * kAlnBlock;     // span ends [base, base + 256)
  // codes of the seeds ending at j in [c_lo, c_hi]: those of span base - 1 (the first lane's predecessor) up to the last span's
  const int c_lo = max(base - 1 - R + kAlnSeed, kAlnSeed - 1), c_hi = min(base + kAlnBlock - 1, W - 1);
  const int s_lo = c_lo - (kAlnSeed - 1);                        // string bases [s_lo, c_hi]
  for (int k = threadIdx.x; k <= c_hi - s_lo; k += kAlnBlock) sh_str[k] = aln_wbase(s, W, strand, s_lo + k);
  __syncthreads();
  for (int k = threadIdx.x; k <= c_hi - c_lo; k += kAlnBlock) {  // scrambled code of the 15-mer ENDING at c_lo + k
    uint32_t code = 0;
    for (int q = 0; q < kAlnSeed; q++) code = (code << 2) | aln_code(sh_str[k + q]);
    sh_code[k] = code ^ 0x2204abcdu;
  }
  __syncthreads();
  auto span_max = [&](int i, int& p) -> uint32_t {  // maximum over the seeds of the span ending at i; p = the earliest seed attaining it
    const int lo_j = max(i - R + kAlnSeed, kAlnSeed - 1);
    uint32_t m = 0;
    p = -1;
    for (int j = lo_j; j <= i; j++) { const uint32_t v = sh_code[j - c_lo]; if (p < 0 || v > m) { m = v; p = j; } }  // strictly greater: earliest maximum
    return m;
  };
  const int i = base + (int)threadIdx.x;
  uint32_t m = 0;
  int p = -1;
  if (i < W) m = span_max(i, p);
  sh_max[threadIdx.x] = m;
  __syncthreads();
  bool emit = false;
  if (i < W) {
    if (i == R - 1) emit = true;
    else {
      // previous span's maximum == what the reference last emitted, except before the first emission
      // (i-1 < R-1 can only happen when R-1 < kAlnSeed; the host path handles such short reads)
      uint32_t mprev;
      if (threadIdx.x > 0) mprev = sh_max[threadIdx.x - 1];
      else { int pp; mprev = span_max(i - 1, pp); }
      emit = m != mprev;
    }
  }
  // ordered compaction inside the block: ballots per wave, wave totals through LDS
  const unsigned long long bal = __ballot(emit);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int before = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) sh_wave[wave] = __popcll(bal);
  __syncthreads();
  int wave_before = 0, total = 0;
  for (int q = 0; q < kAlnBlock / 64; q++) { if (q < wave) wave_before += sh_wave[q]; total += sh_wave[q]; }
  if (threadIdx.x == 0) sh_base = total ? atomicAdd(n_spans, (unsigned)total) : 0;
  __syncthreads();
  if (emit) {
    if (FUSED) {
      const int mt = w >= ix.split ? 1 : 0;
      const int b = aln_find_bucket(ix.bucket_hash[mt], ix.bucket_top[mt], ix.n_buckets[mt], m);
      if (b >= 0) {
        const int b0 = ix.bucket_off[mt][b], b1 = ix.bucket_off[mt][b + 1];
        const unsigned at = atomicAdd(n_cands, (unsigned)(b1 - b0));
        for (int k = b0; k < b1; k++) {
          const unsigned o = at + (unsigned)(k - b0);
          if (o < cap_cands) cands[o] = AlnCand{w, strand, i, p, ix.bucket_reads[mt][k]};
        }
      }
    } else {
      const unsigned at = sh_base + (unsigned)(wave_before + before);
      if (at < cap_spans) spans[at] = AlnSpan{m, p, w, strand, i};
    }
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
struct AlnWaveLds {
  unsigned char rd[kAlnMaxRead + 2];           // the read as aligned (strand applied)
  unsigned char ws[kAlnWinSeg + 2];            // window segment, 0 beyond the window's end
  unsigned char seed[16];
  uint32_t vis[9 * 8];                         // 256-bit visited set per diagonal shift in [-4, 4]; bit = read index + 1
  unsigned short q[kAlnQueue];
};

__device__ __forceinline__ void aln_lds_sync() {  // LDS traffic of ONE wave: in order in hardware, keep the compiler from reordering
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct AlnWaveSearch {
  AlnWaveLds& L;
  const int lane;
  int qn;
  __device__ __forceinline__ AlnWaveSearch(AlnWaveLds& l, int ln) : L(l), lane(ln), qn(0) {}
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
    for (int k = 0; k < kAlnSeed; k++) same = same && L.rd[min(i + k, kAlnMaxRead)] == L.seed[k];
    const unsigned long long hit = __ballot(same);
    if (hit) read_pos = base + (__ffsll((long long)hit) - 1);
  }
  if (read_pos < 0) { if (lane == 0) hits[t] = out; return; }
  // window bases the search can touch: g = win_pos + (r - read_pos) + diag, r in [-1, R], |diag| <= 4
  const int g0 = max(0, win_pos - read_pos - 6);
  const int seg = min(kAlnWinSeg, win_pos + (R - read_pos) + 6 - g0);
  for (int b = lane; b < seg; b += 64) L.ws[b] = (unsigned char)(g0 + b < W ? ws[g0 + b] : '\0');  // the reference reads the terminator at g == W
  aln_lds_sync();
  auto wbase = [&](int g) -> unsigned char { const int i = g - g0; return (i >= 0 && i < seg) ? L.ws[i] : (unsigned char)'\0'; };
  AlnWaveSearch S(L, lane);
  // ---- forward (graph.cc:761-793)
  int fwd = -1;
  S.reset();
  S.push(0, 0, read_pos + kAlnSeed);
  aln_lds_sync();
  for (int qi = 0; qi < S.qn && fwd < 0; qi++) {
    const uint32_t e = L.q[qi];
    const int d = (int)(e >> 12), diag = (int)((e >> 8) & 15) - 4;
    int r = (int)(e & 255) - 1;
    int g = win_pos + (r - read_pos) + diag;
    if (d > 3) { if (lane == 0) hits[t] = out; return; }
    while (true) {
      // lane k looks at cell (g + k, r + k) of the diagonal
      const int rr = r + lane, gg = g + lane;
      const bool at_end = rr == R;
      bool stop = at_end, mism = false;
      if (rr < R) {
        const bool match = wbase(gg) == L.rd[rr];
        const bool may_advance = gg + 1 < W || rr + 1 == R;
        mism = !match;
        stop = !match || !may_advance || S.visited(diag, rr + 1);
      }
      const unsigned long long stops = __ballot(stop && rr <= R);
      if (!stops) {  // 64 matching, unvisited cells: the run goes on
        S.mark_range(diag, r + 1, r + 64);
        g += 64; r += 64;
        continue;
      }
      const int k = __ffsll((long long)stops) - 1;
      S.mark_range(diag, r + 1, r + k);  // the cells the run moved into
      const bool end_here = (__ballot(at_end) >> k) & 1ull;
      const bool mism_here = (__ballot(mism) >> k) & 1ull;
      g += k; r += k;
      if (end_here) { fwd = d; break; }
      if (mism_here) {
        if (g + 1 < W) {
          if (S.mark(diag, r + 1)) S.push(d + 1, diag, r + 1);          // substitution
          if (S.mark(diag + 1, r)) S.push(d + 1, diag + 1, r);          // window base skipped
        }
        if (S.mark(diag - 1, r + 1)) S.push(d + 1, diag - 1, r + 1);    // read base skipped
        aln_lds_sync();
      }
      break;  // mismatch handled, or the run ended at the window's edge / a visited cell
    }
  }
  if (fwd < 0) { if (lane == 0) hits[t] = out; return; }
  // ---- backward (graph.cc:794-835)
  int bwd = -1, begin_pos = -1;
  if (win_pos == 0) {
    if (read_pos < 6) bwd = read_pos;
  } else {
    S.reset();
    S.push(0, 0, read_pos - 1);
    aln_lds_sync();
    for (int qi = 0; qi < S.qn && bwd < 0; qi++) {
      const uint32_t e = L.q[qi];
      const int d = (int)(e >> 12), diag = (int)((e >> 8) & 15) - 4;
      int r = (int)(e & 255) - 1;
      int g = win_pos + (r - read_pos) + diag;
      if (d > 3) { if (lane == 0) hits[t] = out; return; }
      while (true) {
        // lane k looks at cell (g - k, r - k)
        const int rr = r - lane, gg = g - lane;
        const bool at_end = rr == -1;
        bool stop = at_end, mism = false;
        if (rr >= 0) {
          const bool match = wbase(gg) == L.rd[rr];
          const bool may_advance = gg - 1 >= 0 || rr - 1 == -1;
          mism = !match;
          stop = !match || !may_advance || S.visited(diag, rr - 1);
        }
        const unsigned long long stops = __ballot(stop && rr >= -1);
        if (!stops) {
          S.mark_range(diag, r - 64, r - 1);
          g -= 64; r -= 64;
          continue;
        }
        const int k = __ffsll((long long)stops) - 1;
        S.mark_range(diag, r - k, r - 1);
        const bool end_here = (__ballot(at_end) >> k) & 1ull;
        const bool mism_here = (__ballot(mism) >> k) & 1ull;
        g -= k; r -= k;
        if (end_here) { bwd = d; begin_pos = g + 1; break; }
        if (mism_here) {
          if (g - 1 >= 0) {
            if (S.mark(diag, r - 1)) S.push(d + 1, diag, r - 1);
            if (S.mark(diag - 1, r)) S.push(d + 1, diag - 1, r);
          }
          if (S.mark(diag + 1, r - 1)) S.push(d + 1, diag + 1, r - 1);
          aln_lds_sync();
        }
        break;
      }
    }
  }
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

__global__ __launch_bounds__(64 * kAlnWaves) void extend_pair_kernel(const AlnCand* cands, const unsigned* n_cands, unsigned cap_cands,
                                                                     const char* wstr, const AlnWindow* wins, AlnMates ix, AlnHit* hits) {
  __shared__ AlnWaveLds lds_all[kAlnWaves];
  const unsigned n = *n_cands < cap_cands ? *n_cands : cap_cands;
  const int lane = (int)(threadIdx.x & 63);
  AlnWaveLds& L = lds_all[threadIdx.x >> 6];
  for (unsigned t = blockIdx.x * kAlnWaves + (threadIdx.x >> 6); t < n; t += gridDim.x * kAlnWaves) {  // whole waves move together
    const int mt = cands[t].win >= ix.split ? 1 : 0;  // wave-uniform
    extend_candidate(L, lane, t, cands, wstr, wins, ix.reads[mt], ix.read_off[mt], hits);
    aln_lds_sync();  // the wave's LDS slice is reused by its next candidate
  }
}

// Small batches (an annealing move's handful of new junction windows): the counters and the hits go to mapped pinned
// host memory and a sequence word tells the host they are there -- ONE wait per batch, no copy commands. The counters
// are left at zero for the next batch.
__global__ __launch_bounds__(256) void publish_hits_kernel(unsigned* counters, const AlnHit* hits, unsigned cap_cands, unsigned* h_counts,
                                                           AlnHit* h_hits, unsigned cap_host, volatile unsigned long long* h_seq, unsigned long long seq) {
  const unsigned n_spans = counters[0], n_cands = counters[1];
  const unsigned n = n_cands < cap_cands ? (n_cands < cap_host ? n_cands : 0u) : 0u;  // overflow: counts only, the host takes the slow route
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
