// aligner_small.hip.h -- the small batches an annealing move brings (a handful of new junction windows, a few hundred
// seed candidates), BOTH mates of a paired set, in TWO dispatches and one wait (reference AlignSubpathInternal
// graph.cc:839-899, GetMinHashWithPoses / GetReadCandsWithPoses graph.cc:1289-1348, ProcessHit graph.cc:753-837).
// With so few candidates the batch lasts as long as ONE candidate's chain of dependent steps, so both kernels are
// built to keep that chain short rather than to move bytes:
//   span_cands_kernel   sliding maximum as in span_maxima_kernel; an emitted span finds its bucket through an open-
//                       addressing table (one or two loads instead of a two-level binary search), and the block expands
//                       all its spans' buckets TOGETHER (prefix sum over the bucket sizes, one lane per candidate: the
//                       loads of a bucket's reads go out side by side instead of one after the other in a lane). A
//                       candidate carries everything the extension needs (window slice, read offset and length).
//   extend_pair2_kernel two waves per candidate: the forward and the backward search (AlnWaveSearch, aligner.hip.h --
//                       the same code as the general route, so the same records) side by side, each wave with its
//                       own copy of the read and of the window slice, ALL their bytes requested at once. (The search
//                       state in registers -- visited sets and match masks spread over the lanes, readlane instead
//                       of LDS -- was tried: bit-equal and three to four times SLOWER per chain head.)
//                       Hits go straight to mapped pinned host memory; the block that finishes last publishes the
//                       counts and the sequence word the host polls, and leaves the counters at zero.
#pragma once
#include "aligner.hip.h"

namespace gaml {

// timing builds only (tools/build_variant.sh NAME -DGAML_ALN_STAMPS): when the LAST wave passed each stage of the two kernels
#ifdef GAML_ALN_STAMPS
#define g_aln_stamp_pub g_aln_stamp[16]
#define ALN_STAMP(k) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 2) atomicMax(&g_aln_stamp[k], (unsigned long long)wall_clock64()); } while (0)
#define ALN_STAMP_FIRST(k) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 2) atomicMin(&g_aln_stamp[k], (unsigned long long)wall_clock64()); } while (0)
#else
#define ALN_STAMP(k) do { } while (0)
#define ALN_STAMP_FIRST(k) do { } while (0)
#endif

struct AlnCandX {  // 48 bytes
  int32_t win, strand, order, seed_end, read, rlen;
  int64_t roff;                         // the read's bases in its mate's read buffer
  int32_t w_off, w_len, w_offset, mate; // the window's string in the batch buffer, its trimmed prefix (graph.cc:850)
};

__device__ __forceinline__ bool aln_hash_find(const AlnHashSlot* tab, int bits, uint32_t key, int& b0, int& cnt) {
  const uint32_t mask = (1u << bits) - 1u;
  uint32_t h = aln_hash_home(key, bits);
  for (int probe = 0; probe < 64; probe++) {  // (the host builds the table at load <= 1/2 and checks its longest probe sequence)
    const int4 e = *(const int4*)(tab + h);
    if (!e.w) return false;
    if ((uint32_t)e.x == key) { b0 = e.y; cnt = e.z; return true; }
    h = (h + 1) & mask;
  }
  return false;
}

// A handful of windows travel in the kernel's argument segment (scalar loads, no trip to the input block the host
// wrote through the BAR -- uncached memory: the block search and the window header were four dependent loads of ~1 us).
constexpr int kAlnArgWins = 16;
struct AlnWinArgs { int n; int blk[kAlnArgWins + 1]; AlnWindow w[kAlnArgWins]; };
// ... and so do their strings when they fit: FIRST kernel parameter, read through the argument segment's address (never
// by name: taking the address of a by-value parameter would copy it to scratch). Byte loads from the BAR-written input
// block took 7-10 us in either kernel (in-kernel stamps), the argument segment is ordinary cached device memory.
constexpr int kAlnArgStr = 3072;
struct AlnStrArgs { char s[kAlnArgStr]; };
__device__ __forceinline__ const char* aln_arg_strings(const char* fallback, int in_args) {
#if defined(__HIP_DEVICE_COMPILE__)
  return in_args ? (const char*)__builtin_amdgcn_kernarg_segment_ptr() : fallback;
#else
  return fallback;
#endif
}
// (windows [split, n_win) belong to a second read set -- the other mate -- whose index was built for read length R2)
__global__ __launch_bounds__(kAlnBlock) void span_cands_kernel(AlnStrArgs, int str_in_args, const char* wstr, const AlnWindow* wins, int n_win, int R, const int* blk, int split, int R2,
                                                              AlnMates ix, AlnWinArgs wa, AlnCandX* cands, unsigned* n_cands, unsigned cap_cands, char* wcopy,
                                                              unsigned long long* h_started = nullptr, unsigned long long started_seq = 0) {
  __shared__ AlnSpanLds L;
  __shared__ int sh_pref[kAlnBlock], sh_b0[kAlnBlock], sh_p[kAlnBlock];
  __shared__ int sh_wave[kAlnBlock / 64];
  __shared__ unsigned sh_base;
  AlnSpanLane o;
  AlnWindow win;
  int rel;
  ALN_STAMP_FIRST(0); ALN_STAMP(1);
  // (development A/B: block 0 tells the host that the grid has started)
  if (h_started && blockIdx.x == 0 && threadIdx.x == 0) {
    __hip_atomic_store(h_started + 1, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // the device's clock (100 MHz) when the grid started
    __hip_atomic_store(h_started, started_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  wstr = aln_arg_strings(wstr, str_in_args);
  const int w = wa.n > 0 ? aln_span_locate(wa.blk, wa.w, wa.n, win, rel) : aln_span_locate(blk, wins, n_win, win, rel);
  if (!aln_span_front(L, wstr, win, w, rel, w >= split ? R2 : R, o, wcopy)) return;
  ALN_STAMP(2);
  const int mt = o.w >= ix.split ? 1 : 0;  // block-uniform, like o.w and o.strand
  int b0 = 0, cnt = 0;
  if (o.emit) { if (!aln_hash_find(ix.htab[mt], ix.hbits[mt], o.m, b0, cnt)) cnt = 0; }
  ALN_STAMP(3);
  // exclusive prefix of the bucket sizes over the block
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
  if (lane == 63) sh_wave[wave] = incl;
  __syncthreads();
  int wave_before = 0, total = 0;
  for (int q = 0; q < kAlnBlock / 64; q++) { if (q < wave) wave_before += sh_wave[q]; total += sh_wave[q]; }
  sh_pref[threadIdx.x] = wave_before + incl - cnt;
  sh_b0[threadIdx.x] = b0;
  sh_p[threadIdx.x] = o.p;
  if (threadIdx.x == 0) sh_base = total ? atomicAdd(n_cands, (unsigned)total) : 0u;
  __syncthreads();
  ALN_STAMP(4);
  if (total == 0) return;
  const int i0 = o.i - (int)threadIdx.x;  // span end of lane 0
  const int32_t* breads = ix.bucket_reads[mt];
  const int64_t* roffs = ix.read_off[mt];
  const unsigned at0 = sh_base;
  for (int j = threadIdx.x; j < total; j += 2 * kAlnBlock) {  // two candidates per lane and round: their loads side by side
    int own[2], rd[2];
    bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int jj = j + u * kAlnBlock;
      ok[u] = jj < total;
      own[u] = 0; rd[u] = 0;
      if (ok[u]) {
        int lo = 0, hi = kAlnBlock;  // last lane whose prefix is <= jj (lanes without candidates share their successor's prefix)
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sh_pref[mid] <= jj) lo = mid; else hi = mid; }
        own[u] = lo;
        rd[u] = breads[sh_b0[lo] + (jj - sh_pref[lo])];
      }
    }
    int64_t r0[2], r1[2];
#pragma unroll
    for (int u = 0; u < 2; u++) { r0[u] = 0; r1[u] = 0; if (ok[u]) { r0[u] = roffs[rd[u]]; r1[u] = roffs[rd[u] + 1]; } }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const unsigned at = at0 + (unsigned)(j + u * kAlnBlock);
      if (ok[u] && at < cap_cands)
        cands[at] = AlnCandX{o.w, o.strand, i0 + own[u], sh_p[own[u]], rd[u], (int32_t)(r1[u] - r0[u]), r0[u], win.str_off, win.len, win.offset, mt};
    }
  }
  ALN_STAMP(5);
}

using AlnWave2Lds = AlnWaveLdsT<2 * kAlnMaxRead + 16>;  // ws: the window bases ANY placement of the seed in the read can reach
constexpr int kAlnPairs = 2;                             // candidates per block (two waves each)
constexpr int kAlnTicketWord = 32;                       // counters[32]: working blocks done -- a cache line of its own

// counters: [1] candidates (span_cands_kernel), [kAlnTicketWord] working blocks done. h_counts / h_hits / h_seq: mapped pinned host memory.
__global__ __launch_bounds__(128 * kAlnPairs) void extend_pair2_kernel(AlnStrArgs, int str_in_args, const AlnCandX* cands, unsigned* counters, unsigned cap_cands, const char* wstr, AlnMates ix,
                                                                      AlnHit* h_hits, unsigned* h_counts, volatile unsigned long long* h_seq, unsigned long long seq, AlnHit* d_hits, int file_on_device = 0) {
  __shared__ AlnWave2Lds lds_all[2 * kAlnPairs];
  __shared__ int sh_res[kAlnPairs][4];
  __shared__ int sh_last;
  ALN_STAMP_FIRST(8); ALN_STAMP(9);
#ifdef GAML_ALN_STAMPS
  const unsigned long long ck0 = clock64(), wc0 = wall_clock64();
#endif
  (void)str_in_args;  // wstr: the copy span_cands_kernel left in ordinary device memory
  const unsigned n_cands = counters[1];
  const unsigned n = n_cands <= cap_cands ? n_cands : 0u;  // overflow: counts only, the host takes the other route
  const int lane = (int)(threadIdx.x & 63), wv = (int)(threadIdx.x >> 6), pair = wv >> 1, dir = wv & 1;
  AlnWave2Lds& L = lds_all[wv];
  for (unsigned base = blockIdx.x * kAlnPairs; base < n; base += gridDim.x * kAlnPairs) {  // (block-uniform trip count: the barriers below)
    const unsigned t = base + (unsigned)pair;
    const bool live = t < n;
    AlnCandX c{};
    int res = -1, begin_pos = -1;
    if (live) {
      c = cands[t];
#ifdef GAML_ALN_STAMPS
      { int pv = c.win + c.mate + c.rlen; asm volatile("s_waitcnt vmcnt(0)" : "+v"(pv) :: "memory"); }
#endif
      ALN_STAMP(10);
      const int R = c.rlen, W = c.w_len;
      if (R <= kAlnMaxRead && R >= kAlnSeed) {
        const char* ws = wstr + c.w_off;  // ProcessHit always works on the FORWARD window string
        const char* rd = (__builtin_amdgcn_readfirstlane(c.mate) ? ix.reads[1] : ix.reads[0]) + c.roff;  // (a select between kernel arguments: indexing them by a loaded value is a VECTOR load from the argument segment, 10 us on first touch)
        // seed start in the forward window string (graph.cc:866-872)
        const int win_pos = c.strand == 0 ? c.seed_end - kAlnSeed + 1 : W - (c.seed_end + 1);
        // window bases a search can touch whatever the seed's place in the read: g = win_pos + (r - read_pos) + diag, r in [-1, R], |diag| <= 4
        const int seg0 = max(0, win_pos - (R - kAlnSeed) - 6), seg_end = min(W, win_pos + R + 6);
        // the read as aligned: strand 1 = reverse complement of the stored read (graph.cc:873-876). Every byte is requested
        // before the first one is used: a load per loop iteration, each waiting for the previous LDS store, was 10 us.
        unsigned char rb[4], wb[8];
#ifdef GAML_ALN_STAMPS
        { unsigned long long pv = (unsigned long long)rd; asm volatile("s_waitcnt vmcnt(0)" : "+v"(pv) :: "memory"); ALN_STAMP(18); }
#endif
#pragma unroll
        for (int u = 0; u < 4; u++) { const int b = lane + 64 * u; rb[u] = b < R ? (unsigned char)rd[b] : (unsigned char)0; }
#ifdef GAML_ALN_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ALN_STAMP(19);
#endif
#pragma unroll
        for (int u = 0; u < 8; u++) { const int b = lane + 64 * u; wb[u] = b < seg_end - seg0 ? (unsigned char)ws[seg0 + b] : (unsigned char)0; }
#ifdef GAML_ALN_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ALN_STAMP(17);
#endif
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int b = lane + 64 * u;
          if (b < R) { if (c.strand == 0) L.rd[b] = rb[u]; else L.rd[R - 1 - b] = (unsigned char)aln_comp((char)rb[u]); }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) { const int b = lane + 64 * u; if (b < seg_end - seg0) L.ws[b] = wb[u]; }
        aln_lds_sync();
        ALN_STAMP(11);
        // first position of the (oriented) read carrying the window's seed (graph.cc:873-879): 64 positions at a time
        int read_pos = -1;
        for (int b0 = 0; b0 + kAlnSeed <= R && read_pos < 0; b0 += 64) {
          const int i = b0 + lane;
          bool same = i + kAlnSeed <= R;
#pragma unroll
          for (int k = 0; k < kAlnSeed; k++) same = same & (L.rd[min(i + k, kAlnMaxRead)] == L.ws[win_pos - seg0 + k]);
          const unsigned long long hit = __ballot(same);
          if (hit) read_pos = b0 + (__ffsll((long long)hit) - 1);
        }
        ALN_STAMP(12);
        if (read_pos >= 0) {
          auto wbase = [&](int g) -> unsigned char {  // the reference reads the terminator at g == W
            const unsigned char v = L.ws[min(max(g - seg0, 0), 2 * kAlnMaxRead + 15)];
            return (g >= seg0 && g < seg_end) ? v : (unsigned char)'\0';
          };
          AlnWaveSearch<AlnWave2Lds> S(L, lane);
          res = dir == 0 ? S.forward(wbase, R, W, win_pos, read_pos) : S.backward(wbase, W, win_pos, read_pos, begin_pos);
        }
      }
    }
    if (live) ALN_STAMP(13 + dir);
    if (lane == 0) { sh_res[pair][dir * 2] = res; sh_res[pair][dir * 2 + 1] = begin_pos; }
    __syncthreads();
    if (live && dir == 0 && lane == 0) {
      const int fwd = sh_res[pair][0], bwd = sh_res[pair][2], bp = sh_res[pair][3];
      AlnHit out{c.win, 0, -1, c.read, c.strand, c.order};
      if (fwd >= 0 && bwd >= 0) { out.pos = bp + 1 + c.w_offset; out.edit = fwd + bwd; }  // graph.cc:890
      d_hits[t] = out;  // device memory; the publishing block carries all of them to the host in one go (below)
    }
    __syncthreads();  // sh_res and the waves' LDS slices are reused by the next round
  }
  // Only blocks that had candidates draw a ticket (a thousand idle blocks' atomics on one line took 30 us, and the
  // candidate count next to it waited behind them); with no candidate at all block 0 publishes.
  // A block's hits are in DEVICE memory, released at agent scope before its ticket is drawn (the publisher sits on another
  // XCD as a rule: it must find them in memory, not in this XCD's L2). The block that draws the last ticket copies all
  // hits to the host's buffer -- whole words, neighbouring lanes neighbouring words: a few dozen wide PCIe writes --, and
  // publishes. (Every block storing its own hits straight into host memory meant ~900 eight-byte writes in a burst at the
  // end of the kernel, the sequence word queued behind them: the host saw a batch 13-35 us after the device had finished
  // it, by the two clocks side by side -- GAML_ALN_WAIT=6. And with PLAIN stores from every block the host could see the
  // sequence word before some blocks' hits had left their XCD's L2: two processes sharing one GPU lost a few dozen hits of a
  // 3,000-candidate batch once in four runs, tools/dist_diag.py.)
  if (file_on_device) return;  // aln_file_small_kernel, next on the stream, files the hits and publishes (aligner_file.hip.h)
  const unsigned working = n == 0 ? 1u : min((n + kAlnPairs - 1) / kAlnPairs, gridDim.x);
  if (blockIdx.x >= working) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  ALN_STAMP(15);
  if (threadIdx.x == 0) {
    __threadfence();  // release (agent scope): this block's hits, whichever wave stored them (the barrier above)
    const unsigned ticket = __hip_atomic_fetch_add(&counters[kAlnTicketWord], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh_last = ticket == working - 1;
    if (sh_last) __threadfence();  // acquire: every other working block's hits
  }
  __syncthreads();
  if (!sh_last) return;
  {
    const unsigned long long* src = (const unsigned long long*)d_hits;
    unsigned long long* dst = (unsigned long long*)h_hits;
    const unsigned words = n * (unsigned)(sizeof(AlnHit) / 8);
    for (unsigned w = threadIdx.x; w < words; w += blockDim.x) {
      const unsigned long long v = __hip_atomic_load(src + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (past this CU's L1)
      __hip_atomic_store(dst + w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {  // publish, and leave the counters at zero for the next batch
    h_counts[0] = 0; h_counts[1] = n_cands;
    h_counts[2] = (unsigned)wall_clock64(); h_counts[3] = (unsigned)(wall_clock64() >> 32);  // the device's clock at publication (timing probes)
    counters[0] = 0; counters[1] = 0; counters[kAlnTicketWord] = 0;
    __threadfence_system();
#ifdef GAML_ALN_STAMP_AFTER_FENCE
    { const unsigned long long wc = wall_clock64(); __hip_atomic_store((unsigned long long*)(h_counts + 4), wc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
#endif
    __hip_atomic_store((unsigned long long*)h_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#ifdef GAML_ALN_STAMPS
    atomicMax(&g_aln_stamp_pub, (unsigned long long)wall_clock64());
#endif
  }
}

}  // namespace gaml
